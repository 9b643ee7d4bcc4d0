// gfx950 (MI355X / CDNA4): tree_lm_kernel / tree_wave_kernel — forest windows of ONE shared topology (BASELINE config 5; split out of
// window_kernel.hip; the general solver and the shared residual / g2o semantics are described there).
#include "se3_edge_device.h"

namespace locamd {

// =====================================================================================================================
// TREE windows of ONE shared topology, one LANE per window (BASELINE config 5: the key-frame star of addPoseEdge,
// localization.cpp:254-290 — every pose hangs on an older key pose by an EdgeSE3 — plus one anchor range per pose).
//
// A window whose pose-to-pose edges form a FOREST factors without fill when the leaves go first; when every window of the
// batch has the SAME structure (same counts and index tables: the batch is one graph replayed with different measurements),
// that order and the whole elimination schedule are computed ONCE on the host (capi_window.cpp: build_tree_sched) and every
// lane of a wave walks it in lock-step on its own window — chain_lm_kernel's scheme with "the previous pose" replaced by
// "the parent": no per-window ordering / symbolic work, no structure tables in LDS, no divergence, and the state in an
// [entry][lane] HBM workspace (every load / store of the wave is one 512-byte line) instead of per-instance workspaces read
// 8 bytes per cache line (window_lm_kernel moves 68 GB per 16 384 config-5 windows that way).
// Per node: H_nn (21), H_parent,n (36), b (6), G (15 + 6 inverse pivots), y, x, two pose buffers, and a spill slot for the
// sums a node collects from its children (27) — which stay in REGISTERS as long as consecutive nodes of the schedule share
// their parent (the host orders the children of a node heavy subtree first, leaves last: for config 5's caterpillar — eight
// keys in a row, seven leaves each — the spill slot is never touched).
namespace treew {
constexpr int HD = 0, HO = 21, HB = 57, G = 63, Y = 84, X = 90, P = 96, ACC = 120, N = 147;
}
__host__ __device__ inline size_t tree_window_doubles(const WindowCaps& c) {
    return (size_t)c.nv_max * treew::N + (size_t)c.nr_max * 5 + (size_t)c.np_max * 18 + (size_t)c.ns_max * 48;
}
size_t window_tree_workspace_doubles(const WindowCaps& c, long long B) { return (size_t)((B + 63) / 64) * 64 * tree_window_doubles(c); }

namespace {

#define TH(p, f, k) slab[((size_t)(p) * treew::N + (f) + (k)) * 64]
// wave-uniform schedule entries as scalars
__device__ __forceinline__ int sload(const int32_t* p) { return __builtin_amdgcn_readfirstlane(*p); }

// the sums a node collects from its children (27 doubles: 21 + 6): kept in registers while consecutive nodes share their parent
struct AccCache {
    double v[27];
    int owner;                  // position (in the schedule) of the node the registers belong to, -1: none
    unsigned long long spilled; // nodes (by position) whose sums sit in their ACC slot
};
__device__ __forceinline__ void acc_init(AccCache& c) {
#pragma unroll
    for (int k = 0; k < 27; ++k) c.v[k] = 0.0;
    c.owner = -1; c.spilled = 0;
}
// make `pos` (pose slot `slot`) the owner of the registers
__device__ __forceinline__ void acc_switch(AccCache& c, double* slab, const TreeSched& ts, int pos, int slot) {
    if (c.owner == pos) return;
    if (c.owner >= 0) {
        const int os = sload(ts.node + c.owner);
#pragma unroll
        for (int k = 0; k < 27; ++k) TH(os, treew::ACC, k) = c.v[k];
        c.spilled |= 1ull << c.owner;
    }
    if ((c.spilled >> pos) & 1ull) {
#pragma unroll
        for (int k = 0; k < 27; ++k) c.v[k] = TH(slot, treew::ACC, k);
        c.spilled &= ~(1ull << pos);
    } else {
#pragma unroll
        for (int k = 0; k < 27; ++k) c.v[k] = 0.0;
    }
    c.owner = pos;
}
// take the sums collected for node `pos` (zeros if none) and release the registers
__device__ __forceinline__ void acc_take(AccCache& c, double* slab, int pos, int slot, double* out) {
    if (c.owner == pos) {
#pragma unroll
        for (int k = 0; k < 27; ++k) out[k] = c.v[k];
        c.owner = -1;
    } else if ((c.spilled >> pos) & 1ull) {
#pragma unroll
        for (int k = 0; k < 27; ++k) out[k] = TH(slot, treew::ACC, k);
        c.spilled &= ~(1ull << pos);
    } else {
#pragma unroll
        for (int k = 0; k < 27; ++k) out[k] = 0.0;
    }
}

// one sweep over the nodes in elimination order: chi sums always; FULL: H, b and the coupling blocks as well
template <bool FULL, int JAC>
__device__ __forceinline__ void tree_sweep(const WindowArgs& a, const TreeSched& ts, double* slab, int nv, int buf,
                                           double& robust_chi, double& plain_chi, double& max_diag, int& shared_edges) {
    using namespace treew;
    const WindowCaps& c = a.caps;
    const size_t eoff = (size_t)c.nv_max * N, poff = eoff + (size_t)c.nr_max * 5, soff = poff + (size_t)c.np_max * 18;
#define TE(e, k) slab[(eoff + (size_t)(e) * 5 + (k)) * 64]
#define TP(e, k) slab[(poff + (size_t)(e) * 18 + (k)) * 64]
#define TS(e, k) slab[(soff + (size_t)(e) * 48 + (k)) * 64]
    double rsum = 0.0, csum = 0.0, md = 0.0;
    int nshared = 0;
    AccCache lacc;
    acc_init(lacc);
    int pown = -1;
    double Xpar[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) Xpar[k] = 0.0;
    for (int kpos = 0; kpos < nv; ++kpos) {
        const int n = sload(ts.node + kpos), ppos = sload(ts.par + kpos);
        const int pn = ppos >= 0 ? sload(ts.node + ppos) : -1;
        double Xc[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) Xc[k] = TH(n, P + 12 * buf, k);
        if (ppos >= 0 && pown != ppos) {
#pragma unroll
            for (int k = 0; k < 12; ++k) Xpar[k] = TH(pn, P + 12 * buf, k);
            pown = ppos;
        }
        double D[27], Dp[27], O[36];   // own block + b; the parent's share; coupling (rows: parent, columns: this node)
        if (FULL) {
#pragma unroll
            for (int k = 0; k < 27; ++k) { D[k] = 0.0; Dp[k] = 0.0; }
#pragma unroll
            for (int k = 0; k < 36; ++k) O[k] = 0.0;
        }
        int nbin = 0;
        // range edges attached to this node: to anchors (unary) or to the parent
        const int r0 = sload(ts.r_off + kpos), r1 = sload(ts.r_off + kpos + 1);
        for (int ri = r0; ri < r1; ++ri) {
            const int e = sload(ts.r_list + ri);
            const int v0 = sload(ts.r_idx + 2 * e), v1 = sload(ts.r_idx + 2 * e + 1);
            const double meas = TE(ri, 0), info = TE(ri, 1);
            const double off[3] = {TE(ri, 2), TE(ri, 3), TE(ri, 4)};
            const bool first_is_cur = v0 == n;   // endpoint 0 (the lever arm) is this node (else the parent)
            double X0[12], X1[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) { X0[k] = first_is_cur ? Xc[k] : Xpar[k]; X1[k] = first_is_cur ? Xpar[k] : Xc[k]; }
            double p0[3], p1[3];
            mat_vec(X0, off, p0);
            p0[0] += X0[9]; p0[1] += X0[10]; p0[2] += X0[11];
            if (v1 >= 0) { p1[0] = X1[9]; p1[1] = X1[10]; p1[2] = X1[11]; }
            else { const double* an = a.anchors + (size_t)(-1 - v1) * 3; p1[0] = an[0]; p1[1] = an[1]; p1[2] = an[2]; }
            double u[3] = {p0[0] - p1[0], p0[1] - p1[1], p0[2] - p1[2]};
            const double nn = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
            const double err = JAC == 0 ? meas - nn : range_error_plain(X0, X0 + 9, off, p1, meas);
            const double chi = err * (info * err);
            const double aux = 1.0 + chi;
            rsum += fast_log_ge1(aux);
            csum += chi;
            if (FULL) {
                double J0[6], J1[3];
                if (JAC == 0) {
                    const double inv = nn > 0.0 ? 1.0 / nn : 0.0;
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
#pragma unroll
                for (int r = 0; r < 6; ++r) {
#pragma unroll
                    for (int cc = 0; cc <= r; ++cc) {
                        const double h = wr * J0[r] * J0[cc];
                        D[r * (r + 1) / 2 + cc] += first_is_cur ? h : 0.0;
                        Dp[r * (r + 1) / 2 + cc] += first_is_cur ? 0.0 : h;
                    }
                    const double bb = J0[r] * wre;
                    D[21 + r] += first_is_cur ? bb : 0.0;
                    Dp[21 + r] += first_is_cur ? 0.0 : bb;
                }
                if (v1 >= 0) {
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
#pragma unroll
                        for (int cc = 0; cc <= r; ++cc) {
                            const double h = wr * J1[r] * J1[cc];
                            D[r * (r + 1) / 2 + cc] += first_is_cur ? 0.0 : h;
                            Dp[r * (r + 1) / 2 + cc] += first_is_cur ? h : 0.0;
                        }
                        const double bb = J1[r] * wre;
                        D[21 + r] += first_is_cur ? 0.0 : bb;
                        Dp[21 + r] += first_is_cur ? bb : 0.0;
                    }
                    // rows: the parent's J, columns: this node's
                    double jr[6], jc[6];
#pragma unroll
                    for (int r = 0; r < 6; ++r) {
                        jr[r] = first_is_cur ? (r < 3 ? J1[r] : 0.0) : J0[r];
                        jc[r] = first_is_cur ? J0[r] : (r < 3 ? J1[r] : 0.0);
                    }
#pragma unroll
                    for (int r = 0; r < 6; ++r)
#pragma unroll
                        for (int cc = 0; cc < 6; ++cc) O[6 * cc + r] += wr * jr[r] * jc[cc];
                    ++nbin;
                }
            }
        }
        // EdgeSE3 factors between this node and its parent
        const int s0 = sload(ts.s_off + kpos), s1 = sload(ts.s_off + kpos + 1);
        for (int si = s0; si < s1; ++si) {
            const int e = sload(ts.s_list + si);
            const int vi = sload(ts.s_idx + 4 * e);
            const bool robust = sload(ts.s_idx + 4 * e + 2) != 0, i_is_cur = vi == n;
            double val[48], Xi[12], Xj[12];
#pragma unroll
            for (int k = 0; k < 48; ++k) val[k] = TS(si, k);
#pragma unroll
            for (int k = 0; k < 12; ++k) { Xi[k] = i_is_cur ? Xc[k] : Xpar[k]; Xj[k] = i_is_cur ? Xpar[k] : Xc[k]; }
            double Hii[21], Hjj[21], Hoff[36], bi[6], bj[6], rterm;
            // (the "later" pose of the pair = the parent: its rows)
            const double chi = chain_se3_terms<FULL>(Xi, Xj, val, robust, i_is_cur, Hii, Hjj, Hoff, bi, bj, rterm);
            rsum += rterm;
            csum += chi;
            if (FULL) {
#pragma unroll
                for (int k = 0; k < 21; ++k) { D[k] += i_is_cur ? Hii[k] : Hjj[k]; Dp[k] += i_is_cur ? Hjj[k] : Hii[k]; }
#pragma unroll
                for (int k = 0; k < 6; ++k) { D[21 + k] += i_is_cur ? bi[k] : bj[k]; Dp[21 + k] += i_is_cur ? bj[k] : bi[k]; }
#pragma unroll
                for (int k = 0; k < 36; ++k) O[k] += Hoff[k];
                ++nbin;
            }
        }
        // unary priors on this node
        const int q0 = sload(ts.p_off + kpos), q1 = sload(ts.p_off + kpos + 1);
        for (int qi = q0; qi < q1; ++qi) {
            double Zi[12], Wd[6];
#pragma unroll
            for (int k = 0; k < 12; ++k) Zi[k] = TP(qi, k);
#pragma unroll
            for (int k = 0; k < 6; ++k) Wd[k] = TP(qi, 12 + k);
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
                        D[r * (r + 1) / 2 + cc] += h;
                    }
                    double bb = 0.0;
#pragma unroll
                    for (int i = (r < 3 ? 0 : 3); i < (r < 3 ? 3 : 6); ++i) bb += J[i * 6 + r] * (-Wd[i] * err[i]);
                    D[21 + r] += bb;
                }
            }
        }
        if (FULL) {
            double fromkids[27];
            acc_take(lacc, slab, kpos, n, fromkids);
#pragma unroll
            for (int k = 0; k < 27; ++k) D[k] += fromkids[k];
#pragma unroll
            for (int k = 0; k < 21; ++k) TH(n, HD, k) = D[k];
#pragma unroll
            for (int k = 0; k < 6; ++k) TH(n, HB, k) = D[21 + k];
#pragma unroll
            for (int r = 0; r < 6; ++r) md = fmax(md, fabs(D[r * (r + 1) / 2 + r]));
            if (ppos >= 0) {
#pragma unroll
                for (int k = 0; k < 36; ++k) TH(n, HO, k) = O[k];
                acc_switch(lacc, slab, ts, ppos, pn);
#pragma unroll
                for (int k = 0; k < 27; ++k) lacc.v[k] += Dp[k];
                nshared += nbin >= 2 ? nbin : 0;
            }
        }
    }
    robust_chi = rsum; plain_chi = csum; max_diag = md;
    if (FULL) shared_edges = nshared;
}

// pose `n` of the trial state: X (+) dx, read from pose buffer `buf`, written to the other one
__device__ __forceinline__ double tree_apply_step(double* slab, int n, int buf, const double* dx, double lambda) {
    using namespace treew;
    double Xo[12], sc = 0.0;
#pragma unroll
    for (int k = 0; k < 12; ++k) Xo[k] = TH(n, P + 12 * buf, k);
#pragma unroll
    for (int k = 0; k < 6; ++k) sc += dx[k] * (lambda * dx[k] + TH(n, HB, k));
    double Rd[9];
    const double ww = 1.0 - (dx[3] * dx[3] + dx[4] * dx[4] + dx[5] * dx[5]);
    if (ww < 0) { Rd[0] = 1; Rd[1] = 0; Rd[2] = 0; Rd[3] = 0; Rd[4] = 1; Rd[5] = 0; Rd[6] = 0; Rd[7] = 0; Rd[8] = 1; }
    else { const double qd[4] = {sqrt(ww), dx[3], dx[4], dx[5]}; quat_to_mat(qd, Rd); }
    double Rn[9], tn[3];
    mat_mul(Xo, Rd, Rn);
    mat_vec(Xo, dx, tn);
    const int ob = P + 12 * (1 - buf);
#pragma unroll
    for (int k = 0; k < 9; ++k) TH(n, ob, k) = Rn[k];
    TH(n, ob, 9) = Xo[9] + tn[0]; TH(n, ob, 10) = Xo[10] + tn[1]; TH(n, ob, 11) = Xo[11] + tn[2];
    return sc;
}

// (H + lambda I) x = b over the forest: leaves first (Cholesky + forward substitution, a node's Schur update handed to its
// parent), then roots first (back-substitution, the step applied node by node).  x is only written when every pivot was positive
// and finite.
__device__ __forceinline__ bool tree_factor_solve(const TreeSched& ts, double* slab, int nv, double lambda, int buf, double& scale_sum) {
    using namespace treew;
    scale_sum = 0.0;
    bool ok = true;
    AccCache acc;
    acc_init(acc);
    for (int kpos = 0; kpos < nv; ++kpos) {
        const int n = sload(ts.node + kpos), ppos = sload(ts.par + kpos);
        double A[6][6], rhs[6], kids[27];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
#pragma unroll
            for (int cc = 0; cc <= r; ++cc) A[r][cc] = TH(n, HD, r * (r + 1) / 2 + cc);
            A[r][r] += lambda;
            rhs[r] = TH(n, HB, r);
        }
        double Ho[36];
        if (ppos >= 0) {
#pragma unroll
            for (int k = 0; k < 36; ++k) Ho[k] = TH(n, HO, k);
        }
        acc_take(acc, slab, kpos, n, kids);
#pragma unroll
        for (int r = 0; r < 6; ++r) {
#pragma unroll
            for (int cc = 0; cc <= r; ++cc) A[r][cc] -= kids[r * (r + 1) / 2 + cc];
            rhs[r] -= kids[21 + r];
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
                for (int r = cc + 1; r < 6; ++r) { TH(n, G, k) = A[r][cc]; ++k; }
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) { TH(n, G, 15 + r) = ig[r]; TH(n, Y, r) = rhs[r]; }
        if (ppos >= 0) {
            // W = H_parent,n G_n^-T (rows: the parent's); its Schur update W W^T, W y goes to the parent
            double Wm[36];   // entry (r, c) at 6 c + r
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                double w[6];
#pragma unroll
                for (int cc = 0; cc < 6; ++cc) w[cc] = Ho[6 * cc + r];
#pragma unroll
                for (int cc = 0; cc < 6; ++cc) {
                    w[cc] *= ig[cc];
#pragma unroll
                    for (int c2 = cc + 1; c2 < 6; ++c2) w[c2] = __builtin_fma(-w[cc], A[c2][cc], w[c2]);
                }
#pragma unroll
                for (int cc = 0; cc < 6; ++cc) Wm[6 * cc + r] = w[cc];
            }
            acc_switch(acc, slab, ts, ppos, sload(ts.node + ppos));
#pragma unroll
            for (int r = 0; r < 6; ++r) {
#pragma unroll
                for (int c2 = 0; c2 <= r; ++c2) {
                    double s2 = acc.v[r * (r + 1) / 2 + c2];
#pragma unroll
                    for (int k = 0; k < 6; ++k) s2 = __builtin_fma(Wm[6 * k + r], Wm[6 * k + c2], s2);
                    acc.v[r * (r + 1) / 2 + c2] = s2;
                }
                double s3 = acc.v[21 + r];
#pragma unroll
                for (int k = 0; k < 6; ++k) s3 = __builtin_fma(Wm[6 * k + r], rhs[k], s3);
                acc.v[21 + r] = s3;
            }
        }
    }
    if (!ok) {
        for (int kpos = 0; kpos < nv; ++kpos) {
            const int n = sload(ts.node + kpos);
            double dx[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) dx[k] = TH(n, X, k);
            scale_sum += tree_apply_step(slab, n, buf, dx, lambda);
        }
        return false;
    }
    // roots first: x_n = G_n^-T (y_n - W_n^T x_parent), W_n^T x = G_n^-1 (H_parent,n^T x)
    int xown = -1;
    double xpar[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) xpar[k] = 0.0;
    for (int kpos = nv - 1; kpos >= 0; --kpos) {
        const int n = sload(ts.node + kpos), ppos = sload(ts.par + kpos);
        double t[6], Gl[6][6], ig[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) { t[r] = TH(n, Y, r); ig[r] = TH(n, G, 15 + r); }
        {
            int k = 0;
#pragma unroll
            for (int cc = 0; cc < 5; ++cc)
#pragma unroll
                for (int r = cc + 1; r < 6; ++r) { Gl[r][cc] = TH(n, G, k); ++k; }
        }
        if (ppos >= 0) {
            if (xown != ppos) {
                const int pn = sload(ts.node + ppos);
#pragma unroll
                for (int k = 0; k < 6; ++k) xpar[k] = TH(pn, X, k);
                xown = ppos;
            }
            double v[6];
#pragma unroll
            for (int cc = 0; cc < 6; ++cc) {
                double a2 = 0.0;
#pragma unroll
                for (int r = 0; r < 6; ++r) a2 = __builtin_fma(TH(n, HO, 6 * cc + r), xpar[r], a2);
                v[cc] = a2;
            }
#pragma unroll
            for (int cc = 0; cc < 6; ++cc) {
                v[cc] *= ig[cc];
#pragma unroll
                for (int c2 = cc + 1; c2 < 6; ++c2) v[c2] = __builtin_fma(-v[cc], Gl[c2][cc], v[c2]);
                t[cc] -= v[cc];
            }
        }
        double xn[6];
#pragma unroll
        for (int rr = 5; rr >= 0; --rr) {
            xn[rr] = t[rr] * ig[rr];
#pragma unroll
            for (int q2 = 0; q2 < rr; ++q2) t[q2] = __builtin_fma(-Gl[rr][q2], xn[rr], t[q2]);
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) TH(n, X, r) = xn[r];
        scale_sum += tree_apply_step(slab, n, buf, xn, lambda);
    }
    return true;
}

template <int JAC>
__global__ void __launch_bounds__(64, 1) tree_lm_kernel(const WindowArgs a, const TreeSched ts, double* ws) {
    using namespace treew;
    const int lane = threadIdx.x;
    const long long inst = (long long)blockIdx.x * 64 + lane;
    const bool live = inst < a.B;
    const WindowCaps& c = a.caps;
    double* slab = ws + (size_t)blockIdx.x * 64 * tree_window_doubles(c) + lane;
    const int nv = ts.nv, nr = ts.nr, np = ts.np, ns = ts.ns;   // (one topology for the whole batch)
    const size_t src = (size_t)(live ? inst : 0);               // (lanes past the batch replay window 0: their results are discarded)
    const double* gin = a.poses_in + src * c.nv_max * 12;
    double* gout = a.poses + src * c.nv_max * 12;
    for (int p = 0; p < nv; ++p) {
#pragma unroll
        for (int k = 0; k < 12; ++k) TH(p, P, k) = gin[p * 12 + k];
#pragma unroll
        for (int k = 0; k < 6; ++k) TH(p, X, k) = 0.0;
    }
    {
        const size_t eoff = (size_t)c.nv_max * N, poff = eoff + (size_t)c.nr_max * 5, soff = poff + (size_t)c.np_max * 18;
        const double* rval = a.r_val + src * c.nr_max * 5;
        for (int i = 0; i < nr; ++i) {   // in the order the schedule walks them
            const int e = sload(ts.r_list + i);
#pragma unroll
            for (int k = 0; k < 5; ++k) slab[(eoff + (size_t)i * 5 + k) * 64] = rval[5 * e + k];
        }
        const double* pval = a.p_val + src * c.np_max * 18;
        for (int i = 0; i < np; ++i) {
            const int e = sload(ts.p_list + i);
#pragma unroll
            for (int k = 0; k < 18; ++k) slab[(poff + (size_t)i * 18 + k) * 64] = pval[18 * e + k];
        }
        const double* sval = a.s_val + src * c.ns_max * 48;
        for (int i = 0; i < ns; ++i) {
            const int e = sload(ts.s_list + i);
            for (int k = 0; k < 48; ++k) slab[(soff + (size_t)i * 48 + k) * 64] = sval[48 * e + k];
        }
    }
    constexpr double tau = 1e-5, good_lo = 1.0 / 3.0, good_hi = 2.0 / 3.0;
    constexpr int max_trials = 10;
    double lambda = 0.0, ni = 2.0, cur_chi = 0.0, last_plain = 0.0;
    int it = 0, q = 0, trials = 0, terminated = 0, buf = 0, shared_edges = 0;
    bool need_lin = true;
    bool done = nv <= 0 || nr + np + ns <= 0 || a.iterations <= 0;
    while (__ballot(!done)) {
        if (!done) {
            if (need_lin) {
                double plain, md;
                tree_sweep<true, JAC>(a, ts, slab, nv, buf, cur_chi, plain, md, shared_edges);
                last_plain = plain;
                if (it == 0) { lambda = tau * md; ni = 2.0; }
                q = 0;
                need_lin = false;
            }
            double sc;
            const bool ok2 = tree_factor_solve(ts, slab, nv, lambda, buf, sc);
            ++trials;
            double temp_chi, plain2, md2;
            int unused_shared;
            tree_sweep<false, JAC>(a, ts, slab, nv, 1 - buf, temp_chi, plain2, md2, unused_shared);
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
                buf = 1 - buf;
                ++q;
                iteration_over = true;
            } else {
                lambda *= ni;
                ni *= 2.0;
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
            for (int k = 0; k < 12; ++k) gout[p * 12 + k] = TH(p, P + 12 * buf, k);
        }
        double* res = a.result + (size_t)inst * 8;
        res[0] = last_plain; res[1] = cur_chi; res[2] = lambda; res[3] = (double)it; res[4] = (double)trials;
        res[5] = (double)terminated; res[6] = (double)shared_edges; res[7] = nv > 0 ? (double)(ts.depth * 65536 + 2 * nv - ts.nroots) : 0.0;
    }
}
#undef TH
#undef TE
#undef TP
#undef TS

}  // namespace

hipError_t launch_window_tree(const WindowArgs& a, const TreeSched& ts, double* ws, hipStream_t stream) {
    if (a.B <= 0 || !ws || ts.nv <= 0 || ts.nv > 64) return hipErrorInvalidValue;
    const unsigned blocks = (unsigned)((a.B + 63) / 64);
    if (a.jacobian) hipLaunchKernelGGL((tree_lm_kernel<1>), dim3(blocks), dim3(64), 0, stream, a, ts, ws);
    else hipLaunchKernelGGL((tree_lm_kernel<0>), dim3(blocks), dim3(64), 0, stream, a, ts, ws);
    return hipGetLastError();
}

}  // namespace locamd
