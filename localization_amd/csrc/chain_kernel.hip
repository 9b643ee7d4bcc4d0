// gfx950 (MI355X / CDNA4): chain_lm_kernel — chain windows, one LANE per window (split out of window_kernel.hip; the general solver and
// the shared residual / g2o semantics are described there).
#include "se3_edge_device.h"

namespace locamd {

// =====================================================================================================================
// CHAIN windows, one LANE per window (large batches of the reference's own sliding window: cfg/uwb_only.yaml, uwb_imu.yaml).
//
// A window whose binary edges all join CONSECUTIVE poses (the zero-range smoothness edge of Robot::new_vertex) and whose other
// factors are unary (ranges to anchors with the antenna lever arm, IMU / lidar priors) has a block-TRIDIAGONAL H.  With one wave
// per window such a system occupies 2 .. 14 lanes and costs ~4 k wave instructions per LM trial whatever is done (see
// factor_and_solve_small).  Here every lane runs a whole window by itself — the sequential block-tridiagonal Cholesky a CPU
// would run — and the wave's instruction stream is shared by 64 windows: ~12 k instructions per LM trial per 64 windows.
// No cross-lane traffic, no LDS, no barriers.  The state of a window lives in an HBM workspace laid out [entry][lane]
// (every load / store of the wave is one 512-byte line), per pose: H_pp (21), H_p,p-1 (36), b_p (6), G_p (15 + 6 inverse
// pivots), y_p, x_p and two pose buffers: 120 doubles (W_p = L_p,p-1 is never stored: the back-substitution re-forms W^T x).  The LM loops are flattened into one loop of
// passes (a pass = one trial; a lane that starts an iteration linearises first), lanes leave when their window is done.
// The kernel is bound by that workspace traffic (~16 KB per window per trial), not by instruction issue.
// Conditions (checked on the host, capi_window.cpp: batch_topology): every pose-to-pose edge — range edge or EdgeSE3 (addTwistEdge;
// the <JAC, true> instantiation) — joins poses p - 1 and p; edges sorted by their later pose, priors sorted by pose (the order the
// reference adds them in).
// Elimination order = pose order (no fill), so the rounding differs from the general kernel's in the last bits.
namespace chainw {
constexpr int HD = 0, HO = 21, HB = 57, G = 63, Y = 84, X = 90, P = 96, PB = 108, N = 120;
}

// per window: nv_max poses x N doubles, then the edges as [entry][lane] too (range: v0, v1, measurement, information, lever
// arm = 7 doubles; prior: v + 18 values = 19; SE3: vi, vj, robust + 48 values = 51)
__host__ __device__ inline size_t chain_window_doubles(const WindowCaps& c) {
    return (size_t)c.nv_max * chainw::N + (size_t)c.nr_max * 7 + (size_t)c.np_max * 19 + (size_t)c.ns_max * 51;
}
size_t window_chain_workspace_doubles(const WindowCaps& c, long long B) {
    return (size_t)((B + 63) / 64) * 64 * chain_window_doubles(c);
}

namespace {


// one sweep over the window's edges in pose order: chi sums always; FULL: H and b as well (returns the largest diagonal entry)
template <bool FULL, int JAC, bool SE3>
__device__ __forceinline__ void chain_sweep(const WindowArgs& a, double* slab, long long inst, int nv, int nr, int np, int ns, int buf,
                                            double& robust_chi, double& plain_chi, double& max_diag, unsigned long long& ho_kind, int& shared_edges) {
    using namespace chainw;
#define CH(p, f, k) slab[((size_t)(p) * N + (f) + (k)) * 64]
    const WindowCaps& c = a.caps;
    (void)inst;
    // the window's edges, copied into the workspace once per launch ([entry][lane]: one line per load of the wave)
    const size_t eoff = (size_t)c.nv_max * N, poff = eoff + (size_t)c.nr_max * 7, soff = poff + (size_t)c.np_max * 19;
#define CE(e, k) slab[(eoff + (size_t)(e) * 7 + (k)) * 64]
#define CP(e, k) slab[(poff + (size_t)(e) * 19 + (k)) * 64]
#define CS(e, k) slab[(soff + (size_t)(e) * 51 + (k)) * 64]
    (void)soff; (void)ns;
    int es = 0;   // (SE3 edges between consecutive poses — the reference's addTwistEdge —, sorted by their later pose like the ranges)
    double rsum = 0.0, csum = 0.0, md = 0.0;
    double Dp[27], Dc[27], O[36], Xp[12], Xc[12];
#pragma unroll
    for (int k = 0; k < 27; ++k) { Dp[k] = 0.0; Dc[k] = 0.0; }
#pragma unroll
    for (int k = 0; k < 12; ++k) { Xp[k] = 0.0; Xc[k] = 0.0; }
    // The coupling block H_p,p-1 of a pair of poses joined by ONE range edge (the reference's smoothness edge) is the rank-1
    // matrix (w J_p) J_{p-1}^T: it is stored as those two vectors (12 doubles instead of 36 — the block is read twice per LM
    // trial, a quarter of all the bytes a trial moved).  ho_kind: two bits per pose (0 no coupling, 1 rank-1, 2 full block;
    // poses from 32 on always store the full block).
    unsigned long long kinds = 0;
    int nbin = 0, nshared = 0;   // nshared: pose-to-pose edges that share their pair of poses with another edge (result[6]; an EdgeSE3 counts as one edge)
    int nedges_pair = 0;
    double fu[6], fv[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { fu[k] = 0.0; fv[k] = 0.0; }
    int e = 0, q = 0;
    // (the next edge and the next pose are requested one step ahead: nothing else hides a memory round trip here)
    double ne[7], nX[12], npv = 0.0;
#pragma unroll
    for (int k = 0; k < 7; ++k) ne[k] = nr > 0 ? CE(0, k) : 0.0;
    if (np > 0) npv = CP(0, 0);
#pragma unroll
    for (int k = 0; k < 12; ++k) nX[k] = nv > 0 ? CH(0, P + 12 * buf, k) : 0.0;
    for (int p = 0; p < nv; ++p) {
#pragma unroll
        for (int k = 0; k < 12; ++k) Xc[k] = nX[k];
        if (p + 1 < nv) {
#pragma unroll
            for (int k = 0; k < 12; ++k) nX[k] = CH(p + 1, P + 12 * buf, k);
        }
        if (FULL) {
#pragma unroll
            for (int k = 0; k < 27; ++k) Dc[k] = 0.0;
#pragma unroll
            for (int k = 0; k < 36; ++k) O[k] = 0.0;
            nbin = 0;
            nedges_pair = 0;
        }
        // range edges whose later pose is p
        while (e < nr) {
            const int v0 = (int)ne[0], v1 = (int)ne[1];
            if ((v1 > v0 ? v1 : v0) != p) break;
            const double meas = ne[2], info = ne[3];
            const double off[3] = {ne[4], ne[5], ne[6]};
            if (e + 1 < nr) {
#pragma unroll
                for (int k = 0; k < 7; ++k) ne[k] = CE(e + 1, k);
            }
            const bool first_is_cur = v0 == p;   // endpoint 0 (the lever arm) is pose p (else p - 1)
            double X0[12], X1[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) { X0[k] = first_is_cur ? Xc[k] : Xp[k]; X1[k] = first_is_cur ? Xp[k] : Xc[k]; }
            double p0[3], p1[3];
            mat_vec(X0, off, p0);
            p0[0] += X0[9]; p0[1] += X0[10]; p0[2] += X0[11];
            if (v1 >= 0) { p1[0] = X1[9]; p1[1] = X1[10]; p1[2] = X1[11]; }
            else { const double* an = a.anchors + (size_t)(-1 - v1) * 3; p1[0] = an[0]; p1[1] = an[1]; p1[2] = an[2]; }
            double u[3] = {p0[0] - p1[0], p0[1] - p1[1], p0[2] - p1[2]};
            const double n = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
            const double err = JAC == 0 ? meas - n : range_error_plain(X0, X0 + 9, off, p1, meas);
            const double chi = err * (info * err);
            const double aux = 1.0 + chi;
            rsum += fast_log_ge1(aux);
            csum += chi;
            if (FULL) {
                double J0[6], J1[3];
                if (JAC == 0) {
                    const double inv = n > 0.0 ? 1.0 / n : 0.0;
                    u[0] *= inv; u[1] *= inv; u[2] *= inv;
                    double uR[3];
                    mat_tvec(X0, u, uR);
                    J0[0] = -uR[0]; J0[1] = -uR[1]; J0[2] = -uR[2];
                    J0[3] = 2.0 * (uR[1] * off[2] - uR[2] * off[1]);
                    J0[4] = 2.0 * (uR[2] * off[0] - uR[0] * off[2]);
                    J0[5] = 2.0 * (uR[0] * off[1] - uR[1] * off[0]);
                    if (v1 >= 0) mat_tvec(X1, u, J1); else { J1[0] = 0; J1[1] = 0; J1[2] = 0; }
                } else {
                    J0[0] = range_jac_numeric<0>(X0, off, X1, p1, 0, meas);
                    J0[1] = range_jac_numeric<1>(X0, off, X1, p1, 0, meas);
                    J0[2] = range_jac_numeric<2>(X0, off, X1, p1, 0, meas);
                    J0[3] = range_jac_numeric<3>(X0, off, X1, p1, 0, meas);
                    J0[4] = range_jac_numeric<4>(X0, off, X1, p1, 0, meas);
                    J0[5] = range_jac_numeric<5>(X0, off, X1, p1, 0, meas);
                    if (v1 >= 0) {
                        J1[0] = range_jac_numeric<0>(X0, off, X1, p1, 1, meas);
                        J1[1] = range_jac_numeric<1>(X0, off, X1, p1, 1, meas);
                        J1[2] = range_jac_numeric<2>(X0, off, X1, p1, 1, meas);
                    } else { J1[0] = 0; J1[1] = 0; J1[2] = 0; }
                }
                const double wr = info / aux, wre = -wr * err;
                // endpoint 0's block and b: into pose p (Dc) or pose p - 1 (Dp)
#pragma unroll
                for (int r = 0; r < 6; ++r) {
#pragma unroll
                    for (int cc = 0; cc <= r; ++cc) {
                        const double h = wr * J0[r] * J0[cc];
                        Dc[r * (r + 1) / 2 + cc] += first_is_cur ? h : 0.0;
                        Dp[r * (r + 1) / 2 + cc] += first_is_cur ? 0.0 : h;
                    }
                    const double bb = J0[r] * wre;
                    Dc[21 + r] += first_is_cur ? bb : 0.0;
                    Dp[21 + r] += first_is_cur ? 0.0 : bb;
                }
                if (v1 >= 0) {
                    // endpoint 1 (translation part only): into the OTHER pose; off-diagonal block rows = pose p, columns = p - 1
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
#pragma unroll
                        for (int cc = 0; cc <= r; ++cc) {
                            const double h = wr * J1[r] * J1[cc];
                            Dc[r * (r + 1) / 2 + cc] += first_is_cur ? 0.0 : h;
                            Dp[r * (r + 1) / 2 + cc] += first_is_cur ? h : 0.0;
                        }
                        const double bb = J1[r] * wre;
                        Dc[21 + r] += first_is_cur ? 0.0 : bb;
                        Dp[21 + r] += first_is_cur ? bb : 0.0;
                    }
                    // J of pose p (rows) x J of pose p - 1 (columns); J1 has three entries
                    double jr[6], jc[6];
#pragma unroll
                    for (int r = 0; r < 6; ++r) {
                        jr[r] = first_is_cur ? J0[r] : (r < 3 ? J1[r] : 0.0);
                        jc[r] = first_is_cur ? (r < 3 ? J1[r] : 0.0) : J0[r];
                    }
                    if (nbin == 0 && p < 32) {
#pragma unroll
                        for (int r = 0; r < 6; ++r) { fu[r] = wr * jr[r]; fv[r] = jc[r]; }
                    } else {
                        if (nbin == 1 && p < 32) {   // a second edge on the pair: expand the first
#pragma unroll
                            for (int r = 0; r < 6; ++r)
#pragma unroll
                                for (int cc = 0; cc < 6; ++cc) O[6 * cc + r] = fu[r] * fv[cc];
                        }
#pragma unroll
                        for (int r = 0; r < 6; ++r)
#pragma unroll
                            for (int cc = 0; cc < 6; ++cc) O[6 * cc + r] += wr * jr[r] * jc[cc];
                    }
                    ++nbin;
                    ++nedges_pair;
                }
            }
            ++e;
        }
        if (SE3) {
            while (es < ns) {
                const int vi = (int)CS(es, 0), vj = (int)CS(es, 1);
                if ((vj > vi ? vj : vi) != p) break;
                const bool robust = CS(es, 2) != 0.0, i_is_cur = vi == p;
                double val[48], Xi[12], Xj[12];
#pragma unroll
                for (int k = 0; k < 48; ++k) val[k] = CS(es, 3 + k);
#pragma unroll
                for (int k = 0; k < 12; ++k) { Xi[k] = i_is_cur ? Xc[k] : Xp[k]; Xj[k] = i_is_cur ? Xp[k] : Xc[k]; }
                double Hii[21], Hjj[21], Hoff[36], bi[6], bj[6], rterm;
                const double chi = chain_se3_terms<FULL>(Xi, Xj, val, robust, !i_is_cur, Hii, Hjj, Hoff, bi, bj, rterm);
                rsum += rterm;
                csum += chi;
                if (FULL) {
#pragma unroll
                    for (int k = 0; k < 21; ++k) {
                        Dc[k] += i_is_cur ? Hii[k] : Hjj[k];
                        Dp[k] += i_is_cur ? Hjj[k] : Hii[k];
                    }
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        Dc[21 + k] += i_is_cur ? bi[k] : bj[k];
                        Dp[21 + k] += i_is_cur ? bj[k] : bi[k];
                    }
                    if (nbin == 1 && p < 32) {   // a range edge on the pair is waiting as two vectors: expand it
#pragma unroll
                        for (int r = 0; r < 6; ++r)
#pragma unroll
                            for (int cc = 0; cc < 6; ++cc) O[6 * cc + r] = fu[r] * fv[cc];
                    }
#pragma unroll
                    for (int k = 0; k < 36; ++k) O[k] += Hoff[k];
                    nbin += 2;   // (an SE3 coupling is never rank-1: the block is stored in full)
                    ++nedges_pair;
                }
                ++es;
            }
        }
        // unary priors on pose p
        while (q < np && (int)npv == p) {
            double Zi[12], Wd[6];
#pragma unroll
            for (int k = 0; k < 12; ++k) Zi[k] = CP(q, 1 + k);
#pragma unroll
            for (int k = 0; k < 6; ++k) Wd[k] = CP(q, 13 + k);
            double RE[9], tE[3], qq[4];
            mat_mul(Zi, Xc, RE);
            mat_vec(Zi, Xc + 9, tE);
            tE[0] += Zi[9]; tE[1] += Zi[10]; tE[2] += Zi[11];
            mat_to_quat(RE, qq);
            quat_normalize_sign(qq);
            const double err[6] = {tE[0], tE[1], tE[2], qq[1], qq[2], qq[3]};
            double chi = 0.0;
#pragma unroll
            for (int i = 0; i < 6; ++i) chi += err[i] * (Wd[i] * err[i]);
            rsum += chi;
            csum += chi;
            if (FULL) {
                double J[36];
#pragma unroll
                for (int i = 0; i < 36; ++i) J[i] = 0.0;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) J[i * 6 + j] = RE[i * 3 + j];
                quat_right_jac(qq, 1.0, J, 6);
#pragma unroll
                for (int r = 0; r < 6; ++r) {
#pragma unroll
                    for (int cc = 0; cc <= r; ++cc) {
                        double h = 0.0;
                        if ((r < 3) == (cc < 3)) {
#pragma unroll
                            for (int i = (r < 3 ? 0 : 3); i < (r < 3 ? 3 : 6); ++i) h += J[i * 6 + r] * Wd[i] * J[i * 6 + cc];
                        }
                        Dc[r * (r + 1) / 2 + cc] += h;
                    }
                    double bb = 0.0;
#pragma unroll
                    for (int i = (r < 3 ? 0 : 3); i < (r < 3 ? 3 : 6); ++i) bb += J[i * 6 + r] * (-Wd[i] * err[i]);
                    Dc[21 + r] += bb;
                }
            }
            ++q;
            if (q < np) npv = CP(q, 0);
        }
        if (FULL) {
            if (p > 0) {
#pragma unroll
                for (int k = 0; k < 21; ++k) CH(p - 1, HD, k) = Dp[k];
#pragma unroll
                for (int k = 0; k < 6; ++k) CH(p - 1, HB, k) = Dp[21 + k];
#pragma unroll
                for (int r = 0; r < 6; ++r) md = fmax(md, fabs(Dp[r * (r + 1) / 2 + r]));
                if (nbin == 1 && p < 32) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) { CH(p, HO, k) = fu[k]; CH(p, HO, 6 + k) = fv[k]; }
                    kinds |= 1ull << (2 * p);
                } else if (nbin >= 1 || p >= 32) {   // (poses from 32 on are always read as full blocks: zeros when uncoupled)
#pragma unroll
                    for (int k = 0; k < 36; ++k) CH(p, HO, k) = O[k];
                    if (p < 32) kinds |= 2ull << (2 * p);
                }
                nshared += nedges_pair >= 2 ? nedges_pair : 0;
            }
#pragma unroll
            for (int k = 0; k < 27; ++k) Dp[k] = Dc[k];
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) Xp[k] = Xc[k];
    }
    if (FULL && nv > 0) {
#pragma unroll
        for (int k = 0; k < 21; ++k) CH(nv - 1, HD, k) = Dp[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) CH(nv - 1, HB, k) = Dp[21 + k];
#pragma unroll
        for (int r = 0; r < 6; ++r) md = fmax(md, fabs(Dp[r * (r + 1) / 2 + r]));
    }
    robust_chi = rsum; plain_chi = csum; max_diag = md;
    if (FULL) { ho_kind = kinds; shared_edges = nshared; }
#undef CE
#undef CP
#undef CS
}

// pose p of the trial state: X (+) dx, read from pose buffer `buf`, written to the other one (accepting a step flips the
// window's buffer, rejecting it costs nothing); returns the pose's share of g2o's computeScale sum
__device__ __forceinline__ double chain_apply_step(double* slab, int p, int buf, const double* dx, double lambda) {
    using namespace chainw;
    double Xo[12], sc = 0.0;
#pragma unroll
    for (int k = 0; k < 12; ++k) Xo[k] = CH(p, P + 12 * buf, k);
#pragma unroll
    for (int k = 0; k < 6; ++k) sc += dx[k] * (lambda * dx[k] + CH(p, HB, k));
    double Rd[9];
    const double ww = 1.0 - (dx[3] * dx[3] + dx[4] * dx[4] + dx[5] * dx[5]);
    if (ww < 0) { Rd[0] = 1; Rd[1] = 0; Rd[2] = 0; Rd[3] = 0; Rd[4] = 1; Rd[5] = 0; Rd[6] = 0; Rd[7] = 0; Rd[8] = 1; }
    else { const double qd[4] = {sqrt(ww), dx[3], dx[4], dx[5]}; quat_to_mat(qd, Rd); }
    double Rn[9], tn[3];
    mat_mul(Xo, Rd, Rn);
    mat_vec(Xo, dx, tn);
    const int ob = P + 12 * (1 - buf);
#pragma unroll
    for (int k = 0; k < 9; ++k) CH(p, ob, k) = Rn[k];
    CH(p, ob, 9) = Xo[9] + tn[0]; CH(p, ob, 10) = Xo[10] + tn[1]; CH(p, ob, 11) = Xo[11] + tn[2];
    return sc;
}

// (H + lambda I) x = b for a block-tridiagonal H: forward sweep (Cholesky + forward substitution), then the back-substitution
// with the step applied pose by pose as its x comes out.  x is only written when every pivot was positive and finite (g2o
// leaves its x alone when the factorisation fails, and LM applies that stale x all the same).
__device__ __forceinline__ bool chain_factor_solve(double* slab, int nv, double lambda, int buf, unsigned long long ho_kind, double& scale_sum) {
    using namespace chainw;
    scale_sum = 0.0;
    // kind of pose p's coupling block (chain_sweep): 0 none, 1 rank-1 (12 doubles), 2 full (36)
    auto kind_of = [&](int p) { return p < 32 ? (int)((ho_kind >> (2 * p)) & 3ull) : (p > 0 ? 2 : 0); };
    bool ok = true;
    double Gp[6][6], igp[6], yp[6];   // the previous pose's factor (strict lower), inverse pivots, y
#pragma unroll
    for (int r = 0; r < 6; ++r) { igp[r] = 0.0; yp[r] = 0.0;
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) Gp[r][cc] = 0.0; }
    // (the next pose's H is requested before the current pose is worked on: one wave per SIMD has nobody else to hide the
    //  memory round trip behind)
    double nHd[21], nHb[6], nHo[36];
#pragma unroll
    for (int k = 0; k < 21; ++k) nHd[k] = nv > 0 ? CH(0, HD, k) : 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) nHb[k] = nv > 0 ? CH(0, HB, k) : 0.0;
#pragma unroll
    for (int k = 0; k < 36; ++k) nHo[k] = 0.0;
    for (int p = 0; p < nv; ++p) {
        double A[6][6], rhs[6], Ho[36];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
#pragma unroll
            for (int cc = 0; cc <= r; ++cc) A[r][cc] = nHd[r * (r + 1) / 2 + cc];
            A[r][r] += lambda;
            rhs[r] = nHb[r];
        }
#pragma unroll
        for (int k = 0; k < 36; ++k) Ho[k] = nHo[k];
        const int kd = kind_of(p);
        if (kd == 1) {   // rank-1: u v^T
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int cc = 0; cc < 6; ++cc) Ho[6 * cc + r] = nHo[r] * nHo[6 + cc];
        }
        if (p + 1 < nv) {
#pragma unroll
            for (int k = 0; k < 21; ++k) nHd[k] = CH(p + 1, HD, k);
#pragma unroll
            for (int k = 0; k < 6; ++k) nHb[k] = CH(p + 1, HB, k);
            const int kn = kind_of(p + 1);
            if (kn == 1) {
#pragma unroll
                for (int k = 0; k < 12; ++k) nHo[k] = CH(p + 1, HO, k);
            } else if (kn == 2) {
#pragma unroll
                for (int k = 0; k < 36; ++k) nHo[k] = CH(p + 1, HO, k);
            }
        }
        if (kd != 0) {
            // row by row: w = row r of W = H_p,p-1 G_{p-1}^-T; S -= w w^T; rhs_r -= w . y_{p-1}
            double Wm[36];   // W, entry (r, c) at 6 c + r
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                double w[6];
#pragma unroll
                for (int cc = 0; cc < 6; ++cc) w[cc] = Ho[6 * cc + r];
#pragma unroll
                for (int cc = 0; cc < 6; ++cc) {
                    w[cc] *= igp[cc];
#pragma unroll
                    for (int c2 = cc + 1; c2 < 6; ++c2) w[c2] = __builtin_fma(-w[cc], Gp[c2][cc], w[c2]);
                }
#pragma unroll
                for (int cc = 0; cc < 6; ++cc) Wm[6 * cc + r] = w[cc];   // (not stored: the back-substitution re-forms W^T x from H_p,p-1 and G)
                double acc = rhs[r];
#pragma unroll
                for (int cc = 0; cc < 6; ++cc) acc = __builtin_fma(-w[cc], yp[cc], acc);
                rhs[r] = acc;
            }
            // S -= W W^T (lower triangle)
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c2 = 0; c2 <= r; ++c2) {
                    double s2 = A[r][c2];
#pragma unroll
                    for (int k = 0; k < 6; ++k) s2 = __builtin_fma(-Wm[6 * k + r], Wm[6 * k + c2], s2);
                    A[r][c2] = s2;
                }
        }
        double ig[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const double g = pivot_rsqrt(A[j][j]);
            ig[j] = g;
#pragma unroll
            for (int i2 = j + 1; i2 < 6; ++i2) A[i2][j] *= g;
#pragma unroll
            for (int i2 = j + 1; i2 < 6; ++i2)
#pragma unroll
                for (int cc = j + 1; cc <= i2; ++cc) A[i2][cc] = __builtin_fma(-A[i2][j], A[cc][j], A[i2][cc]);
        }
        ok = ok && (((ig[0] + ig[1]) + (ig[2] + ig[3])) + (ig[4] + ig[5]) < DBL_MAX);
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) {
            rhs[cc] *= ig[cc];
#pragma unroll
            for (int c2 = cc + 1; c2 < 6; ++c2) rhs[c2] = __builtin_fma(-rhs[cc], A[c2][cc], rhs[c2]);
        }
        {
            int k = 0;
#pragma unroll
            for (int cc = 0; cc < 5; ++cc)
#pragma unroll
                for (int r = cc + 1; r < 6; ++r) { CH(p, G, k) = A[r][cc]; ++k; }
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) { CH(p, G, 15 + r) = ig[r]; CH(p, Y, r) = rhs[r]; igp[r] = ig[r]; yp[r] = rhs[r]; }
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int cc = 0; cc < 6; ++cc) Gp[r][cc] = cc < r ? A[r][cc] : 0.0;
    }
    if (!ok) {
        for (int p = 0; p < nv; ++p) {
            double dx[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) dx[k] = CH(p, X, k);
            scale_sum += chain_apply_step(slab, p, buf, dx, lambda);
        }
        return false;
    }
    double xn[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) xn[r] = 0.0;
    // x_p = G_p^-T (y_p - W_{p+1}^T x_{p+1}) with W_{p+1}^T x = G_p^-1 (H_{p+1,p}^T x): W itself is never stored (it was a fifth of the
    // bytes a trial moved)
    double nG[21], nY[6], nW[36];   // (requested one pose ahead, as in the forward sweep; nW: H_{p+1,p})
#pragma unroll
    for (int k = 0; k < 21; ++k) nG[k] = CH(nv - 1, G, k);
#pragma unroll
    for (int k = 0; k < 6; ++k) nY[k] = CH(nv - 1, Y, k);
#pragma unroll
    for (int k = 0; k < 36; ++k) nW[k] = 0.0;
    for (int p = nv - 1; p >= 0; --p) {
        double t[6], Gl[6][6], ig[6], Wn[36];
#pragma unroll
        for (int r = 0; r < 6; ++r) { t[r] = nY[r]; ig[r] = nG[15 + r]; }
        {
            int k = 0;
#pragma unroll
            for (int cc = 0; cc < 5; ++cc)
#pragma unroll
                for (int r = cc + 1; r < 6; ++r) { Gl[r][cc] = nG[k]; ++k; }
        }
#pragma unroll
        for (int k = 0; k < 36; ++k) Wn[k] = nW[k];   // W of pose p + 1
        if (p > 0) {
#pragma unroll
            for (int k = 0; k < 21; ++k) nG[k] = CH(p - 1, G, k);
#pragma unroll
            for (int k = 0; k < 6; ++k) nY[k] = CH(p - 1, Y, k);
        }
        const int kup = p < nv - 1 ? kind_of(p + 1) : 0;   // kind of the block that couples pose p + 1 to this one (held in Wn)
        {
            const int kme = kind_of(p);   // (pose p's coupling block is what pose p - 1 needs next)
            if (kme == 1) {
#pragma unroll
                for (int k = 0; k < 12; ++k) nW[k] = CH(p, HO, k);
            } else if (kme == 2) {
#pragma unroll
                for (int k = 0; k < 36; ++k) nW[k] = CH(p, HO, k);
            }
        }
        if (kup != 0) {
            double v[6];
            if (kup == 1) {   // (u v^T)^T x = v (u . x)
                double sx = 0.0;
#pragma unroll
                for (int r = 0; r < 6; ++r) sx = __builtin_fma(Wn[r], xn[r], sx);
#pragma unroll
                for (int cc = 0; cc < 6; ++cc) v[cc] = Wn[6 + cc] * sx;
            } else {
#pragma unroll
                for (int cc = 0; cc < 6; ++cc) {
                    double acc = 0.0;
#pragma unroll
                    for (int r = 0; r < 6; ++r) acc = __builtin_fma(Wn[6 * cc + r], xn[r], acc);
                    v[cc] = acc;
                }
            }
            // z = G_p^-1 v (forward substitution), t -= z
#pragma unroll
            for (int cc = 0; cc < 6; ++cc) {
                v[cc] *= ig[cc];
#pragma unroll
                for (int c2 = cc + 1; c2 < 6; ++c2) v[c2] = __builtin_fma(-v[cc], Gl[c2][cc], v[c2]);
                t[cc] -= v[cc];
            }
        }
#pragma unroll
        for (int rr = 5; rr >= 0; --rr) {
            xn[rr] = t[rr] * ig[rr];
#pragma unroll
            for (int q2 = 0; q2 < rr; ++q2) t[q2] = __builtin_fma(-Gl[rr][q2], xn[rr], t[q2]);
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) CH(p, X, r) = xn[r];
        scale_sum += chain_apply_step(slab, p, buf, xn, lambda);
    }
    return true;
}

template <int JAC, bool SE3>
__global__ void __launch_bounds__(64, 1) chain_lm_kernel(const WindowArgs a, double* ws) {
    using namespace chainw;
    const int lane = threadIdx.x;
    const long long inst = (long long)blockIdx.x * 64 + lane;
    const bool live = inst < a.B;
    const WindowCaps& c = a.caps;
    double* slab = ws + (size_t)blockIdx.x * 64 * chain_window_doubles(c) + lane;
    int nv = 0, nr = 0, np = 0, ns = 0;
    if (live) { nv = a.counts[inst * 4 + 0]; nr = a.counts[inst * 4 + 1]; np = a.counts[inst * 4 + 2]; ns = SE3 ? a.counts[inst * 4 + 3] : 0; }
    const double* gin = a.poses_in + (size_t)(live ? inst : 0) * c.nv_max * 12;
    double* gout = a.poses + (size_t)(live ? inst : 0) * c.nv_max * 12;
    for (int p = 0; p < nv; ++p) {
#pragma unroll
        for (int k = 0; k < 12; ++k) CH(p, P, k) = gin[p * 12 + k];
#pragma unroll
        for (int k = 0; k < 6; ++k) CH(p, X, k) = 0.0;   // the solver's x of a fresh optimize() call
    }
    {
        const size_t eoff = (size_t)c.nv_max * N, poff = eoff + (size_t)c.nr_max * 7;
        const int32_t* ridx = a.r_idx + (size_t)(live ? inst : 0) * c.nr_max * 2;
        const double* rval = a.r_val + (size_t)(live ? inst : 0) * c.nr_max * 5;
        for (int e = 0; e < nr; ++e) {
            slab[(eoff + (size_t)e * 7 + 0) * 64] = (double)ridx[2 * e];
            slab[(eoff + (size_t)e * 7 + 1) * 64] = (double)ridx[2 * e + 1];
#pragma unroll
            for (int k = 0; k < 5; ++k) slab[(eoff + (size_t)e * 7 + 2 + k) * 64] = rval[5 * e + k];
        }
        const int32_t* pidx = a.p_idx + (size_t)(live ? inst : 0) * c.np_max;
        const double* pval = a.p_val + (size_t)(live ? inst : 0) * c.np_max * 18;
        for (int e = 0; e < np; ++e) {
            slab[(poff + (size_t)e * 19) * 64] = (double)pidx[e];
#pragma unroll
            for (int k = 0; k < 18; ++k) slab[(poff + (size_t)e * 19 + 1 + k) * 64] = pval[18 * e + k];
        }
        if (SE3) {
            const size_t soff = poff + (size_t)c.np_max * 19;
            const int32_t* sidx = a.s_idx + (size_t)(live ? inst : 0) * c.ns_max * 4;
            const double* sval = a.s_val + (size_t)(live ? inst : 0) * c.ns_max * 48;
            for (int e = 0; e < ns; ++e) {
#pragma unroll
                for (int k = 0; k < 3; ++k) slab[(soff + (size_t)e * 51 + k) * 64] = (double)sidx[4 * e + k];
                for (int k = 0; k < 48; ++k) slab[(soff + (size_t)e * 51 + 3 + k) * 64] = sval[48 * e + k];
            }
        }
    }
    constexpr double tau = 1e-5, good_lo = 1.0 / 3.0, good_hi = 2.0 / 3.0;
    constexpr int max_trials = 10;
    double lambda = 0.0, ni = 2.0, cur_chi = 0.0, last_plain = 0.0;
    int it = 0, q = 0, trials = 0, terminated = 0, buf = 0, shared_edges = 0;
    unsigned long long ho_kind = 0;
    bool need_lin = true;
    bool done = !live || nv <= 0 || nr + np + ns <= 0 || a.iterations <= 0;
    while (__ballot(!done)) {
        if (!done) {
            if (need_lin) {
                double plain, md;
                chain_sweep<true, JAC, SE3>(a, slab, inst, nv, nr, np, ns, buf, cur_chi, plain, md, ho_kind, shared_edges);
                last_plain = plain;
                if (it == 0) { lambda = tau * md; ni = 2.0; }
                q = 0;
                need_lin = false;
            }
            // solve and apply the step (the trial state goes to the other pose buffer)
            double sc;
            const bool ok2 = chain_factor_solve(slab, nv, lambda, buf, ho_kind, sc);
            ++trials;
            double temp_chi, plain2, md2;
            unsigned long long unused_kind;
            int unused_shared;
            chain_sweep<false, JAC, SE3>(a, slab, inst, nv, nr, np, ns, 1 - buf, temp_chi, plain2, md2, unused_kind, unused_shared);
            last_plain = plain2;
            if (!ok2) temp_chi = DBL_MAX;
            const double scale = sc + 1e-3;
            const double rho = (cur_chi - temp_chi) / scale;
            bool iteration_over;
            if (rho > 0.0 && fabs(temp_chi) <= DBL_MAX) {
                const double r21 = 2.0 * rho - 1.0;
                double alpha = 1.0 - r21 * r21 * r21;
                alpha = fmin(alpha, good_hi);
                lambda *= fmax(good_lo, alpha);
                ni = 2.0;
                cur_chi = temp_chi;
                buf = 1 - buf;   // the trial state is the state
                ++q;
                iteration_over = true;
            } else {
                lambda *= ni;
                ni *= 2.0;      // (pop: the state was never overwritten)
                ++q;
                iteration_over = !(rho < 0.0 && q < max_trials);
            }
            if (iteration_over) {
                ++it;
                need_lin = true;
                if (q == max_trials || rho == 0.0) { terminated = 1; done = true; }
                if (it >= a.iterations) done = true;
            }
        }
    }
    if (live) {
        for (int p = 0; p < nv; ++p) {
#pragma unroll
            for (int k = 0; k < 12; ++k) gout[p * 12 + k] = CH(p, P + 12 * buf, k);
        }
        double* res = a.result + (size_t)inst * 8;
        res[0] = last_plain; res[1] = cur_chi; res[2] = lambda; res[3] = (double)it; res[4] = (double)trials;
        res[5] = (double)terminated; res[6] = (double)shared_edges; res[7] = nv > 0 ? (double)(nv * 65536 + 2 * nv - 1) : 0.0;
    }
#undef CH
}

}  // namespace

hipError_t launch_window_chain(const WindowArgs& a, double* chain_ws, hipStream_t stream) {
    if (a.B <= 0 || !chain_ws) return hipErrorInvalidValue;
    const unsigned blocks = (unsigned)((a.B + 63) / 64);
    // (the variant with EdgeSE3 factors between consecutive poses is a separate instantiation: the range-only windows keep their
    //  register budget)
    if (a.caps.ns_max > 0) {
        if (a.jacobian) hipLaunchKernelGGL((chain_lm_kernel<1, true>), dim3(blocks), dim3(64), 0, stream, a, chain_ws);
        else hipLaunchKernelGGL((chain_lm_kernel<0, true>), dim3(blocks), dim3(64), 0, stream, a, chain_ws);
    } else {
        if (a.jacobian) hipLaunchKernelGGL((chain_lm_kernel<1, false>), dim3(blocks), dim3(64), 0, stream, a, chain_ws);
        else hipLaunchKernelGGL((chain_lm_kernel<0, false>), dim3(blocks), dim3(64), 0, stream, a, chain_ws);
    }
    return hipGetLastError();
}


}  // namespace locamd
