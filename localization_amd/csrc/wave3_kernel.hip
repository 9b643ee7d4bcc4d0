// gfx950 (MI355X / CDNA4): TRANSLATION-ONLY chain windows, one WAVE per window — the drop-in node's own solve (one window per
// range message, cfg/uwb_only.yaml) and small batches of such windows.
//
// What it solves (reference file:line): the graph Localization::addRangeEdge builds per range message
// (localization.cpp:297-376) — per pose one EdgeSE3Range to an anchor (:331) and the zero-range smoothness edge to the previous
// pose (:338-340), Cauchy kernels (:608-627) — solved by Localization::solve() = g2o Levenberg-Marquardt (:164-170), chi2() (:197).
// The 3-DoF form is exact for these graphs (chain3_kernel.hip's header: identity antenna offsets, identity rotations, no rotation
// information — every dropped term of the 6x6-block system is an exact zero); the host takes this kernel under the same
// conditions as chain3_lm_kernel, for batches too small for one lane per window.
//
// MI355X mapping.  A single window is a latency problem: one wave issues one f64 instruction per ~8 cycles whatever its 64 lanes
// hold, so the general wave-per-window kernel's 6x6-block schedule (~9 k instructions per LM iteration on a ten-pose window,
// 0.36 ms per solve) is cut to what this graph needs:
//   * lane = EDGE for everything per edge (residual, robust weight, the twelve perturbed norms of g2o's numeric Jacobian): one
//     pass of ~300 instructions whatever the edge count (<= 64 per pass); an edge leaves (w, -w e, J0, J1) in LDS;
//   * lane = POSE for the normal equations: pose p sums its own edges' records (host order = g2o's accumulation order) into
//     H_pp (6), b_p (3) and the 3x3 coupling block with pose p - 1, all in that lane's registers for every trial of the iteration;
//   * the block-tridiagonal Cholesky is inherently sequential: step p runs on lane p alone (3x3 blocks: ~100 instructions) and
//     hands G_p, y_p to step p + 1 through v_readlane (SGPRs, no LDS round trip); the back-substitution hands O_p^T x_p back the
//     same way;
//   * nothing but the poses and the per-edge records ever leaves registers; no barriers (one wave per workgroup), no HBM
//     workspace, 4 KB of LDS for a ten-pose window.
#include "window_kernel.h"
#include "device_math.h"

#include <float.h>
#include <math.h>

namespace locamd {

namespace {

extern __shared__ double w3lds[];

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double w3_dpp(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
// value of lane l (wave-uniform l) in every lane — through SGPRs
__device__ __forceinline__ double w3_bcast(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double w3_sum(double v) {   // DPP row shifts + row broadcasts: one fixed order, every lane gets the same bits
    v += w3_dpp<0x111, 0xF>(v);
    v += w3_dpp<0x112, 0xF>(v);
    v += w3_dpp<0x114, 0xF>(v);
    v += w3_dpp<0x118, 0xF>(v);
    v += w3_dpp<0x142, 0xA>(v);
    v += w3_dpp<0x143, 0xC>(v);
    return w3_bcast(v, 63);
}
__device__ __forceinline__ double w3_max(double v) {   // non-negative inputs
    v = fmax(v, w3_dpp<0x111, 0xF>(v));
    v = fmax(v, w3_dpp<0x112, 0xF>(v));
    v = fmax(v, w3_dpp<0x114, 0xF>(v));
    v = fmax(v, w3_dpp<0x118, 0xF>(v));
    v = fmax(v, w3_dpp<0x142, 0xA>(v));
    v = fmax(v, w3_dpp<0x143, 0xC>(v));
    return w3_bcast(v, 63);
}
// one wave per workgroup: the LDS operations of a wave execute in order; the fences keep the compiler from moving them
__device__ __forceinline__ void w3_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ double w3_pivot_rsqrt(double d) {   // window_kernel.hip: pivot_rsqrt
    const double y = __builtin_amdgcn_rsq(d);
    const double t = d * y;
    const double e = __builtin_fma(-t, y, 1.0);
    const double pq = __builtin_fma(0.375, e, 0.5);
    const double ye = y * e;
    return __builtin_fma(ye, pq, y);
}

#pragma clang fp contract(off)
// ||d|| the way a plain CPU build of computeError evaluates it (numeric_jacobian.h: range_error_plain with a zero lever arm)
__device__ __forceinline__ double w3_norm_plain(double dx, double dy, double dz) { return sqrt_ieee_unscaled(dx * dx + dy * dy + dz * dz); }
// g2o's central difference of e = meas - ||p0 - p1|| along axis D of endpoint `which`'s translation (R = I, zero lever arm:
// X * fromVectorMQT(+-delta e_D) = (I, t +- delta e_D); chain3_kernel.hip: range_jac_numeric3)
template <int D>
__device__ __forceinline__ double w3_jac_numeric(const double* p0, const double* p1, int which, double meas) {
    constexpr double delta = 1e-9;
    constexpr double scalar = 1.0 / (2 * delta);
    double a[3] = {p0[0], p0[1], p0[2]}, b[3] = {p1[0], p1[1], p1[2]}, am[3] = {p0[0], p0[1], p0[2]}, bm[3] = {p1[0], p1[1], p1[2]};
    if (which == 0) { a[D] = delta + p0[D]; am[D] = -delta + p0[D]; }
    else { b[D] = delta + p1[D]; bm[D] = -delta + p1[D]; }
    const double ep = meas - w3_norm_plain(a[0] - b[0], a[1] - b[1], a[2] - b[2]);
    const double em = meas - w3_norm_plain(am[0] - bm[0], am[1] - bm[1], am[2] - bm[2]);
    double bak = ep;
    bak -= em;
    return scalar * bak;
}
#pragma clang fp contract(fast)

// LDS of one window (doubles first, then ints)
struct W3Lds {
    double* T;      // [2][nv_max][3] translations: state / trial state
    double* rec;    // [nr_max][8]    w, -w e, J0 (3), J1 (3) of the last linearisation
    double* ev;     // [nr_max][2]    measurement, information
    double* fix;    // [nr_max][3]    the fixed endpoint of an anchor edge
    double* pv;     // [np_max][6]    priors: Z^-1 t (3), information diagonal (3)
    int* eidx;      // [nr_max][2]
    int* pidx;      // [np_max]
    int* list;      // [2 nr_max]     the poses' incident edges, pose by pose (each pose keeps its range in registers):
                    //                (edge << 2) | (other endpoint is the previous pose) << 1 | (this pose is endpoint 1)
};
__device__ __forceinline__ W3Lds w3_carve(const WindowCaps& c) {
    W3Lds l;
    double* p = w3lds;
    l.T = p; p += 2 * c.nv_max * 3;
    l.rec = p; p += (size_t)c.nr_max * 8;
    l.ev = p; p += (size_t)c.nr_max * 2;
    l.fix = p; p += (size_t)c.nr_max * 3;
    l.pv = p; p += (size_t)c.np_max * 6;
    int* q = reinterpret_cast<int*>(p);
    l.eidx = q; q += (size_t)c.nr_max * 2;
    l.pidx = q; q += c.np_max;
    l.list = q;
    return l;
}

// every edge and prior of the window at the translations of buffer `buf`: the robust and plain chi2 sums; FULL: the edges'
// linearisation records as well
template <bool FULL, int JAC>
__device__ __forceinline__ void w3_edges(const W3Lds& l, int nvm, int nr, int np, int buf, int lane, double& robust_chi, double& plain_chi) {
    double rsum = 0.0, csum = 0.0;
    const double* T = l.T + (size_t)buf * nvm * 3;
    for (int e = lane; e < nr; e += 64) {
        const int v0 = l.eidx[2 * e], v1 = l.eidx[2 * e + 1];
        const double meas = l.ev[2 * e], info = l.ev[2 * e + 1];
        const double* s0 = T + v0 * 3;
        const double* s1 = v1 >= 0 ? T + v1 * 3 : l.fix + (size_t)e * 3;
        const double p0[3] = {s0[0], s0[1], s0[2]}, p1[3] = {s1[0], s1[1], s1[2]};
        double u[3] = {p0[0] - p1[0], p0[1] - p1[1], p0[2] - p1[2]};
        double err, inv = 0.0;
        if (JAC == 0) {
            const double x = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
            double n;
            sqrt_and_rsqrt(x, n, inv);
            if (!(x > 0.0)) { n = 0.0; inv = 0.0; }   // coincident endpoints: J = 0, what the central difference gives
            err = meas - n;
        } else {
            err = meas - w3_norm_plain(u[0], u[1], u[2]);
        }
        const double chi = err * (info * err);
        const double aux = 1.0 + chi;
        rsum += fast_log_ge1(aux);
        csum += chi;
        if (FULL) {
            double J0[3], J1[3];
            if (JAC == 0) {
                u[0] *= inv; u[1] *= inv; u[2] *= inv;
                J0[0] = -u[0]; J0[1] = -u[1]; J0[2] = -u[2];
                J1[0] = u[0]; J1[1] = u[1]; J1[2] = u[2];
            } else {
                J0[0] = w3_jac_numeric<0>(p0, p1, 0, meas);
                J0[1] = w3_jac_numeric<1>(p0, p1, 0, meas);
                J0[2] = w3_jac_numeric<2>(p0, p1, 0, meas);
                J1[0] = w3_jac_numeric<0>(p0, p1, 1, meas);
                J1[1] = w3_jac_numeric<1>(p0, p1, 1, meas);
                J1[2] = w3_jac_numeric<2>(p0, p1, 1, meas);
            }
            const double wr = info * fast_rcp(aux), wre = -wr * err;
            double* r = l.rec + (size_t)e * 8;
            r[0] = wr; r[1] = wre;
#pragma unroll
            for (int k = 0; k < 3; ++k) { r[2 + k] = J0[k]; r[5 + k] = v1 >= 0 ? J1[k] : 0.0; }
        }
    }
    // unary priors: e = t + Z^-1.t (identity rotations), diagonal information on the translation, no robust kernel
    for (int q = lane; q < np; q += 64) {
        const double* s = T + l.pidx[q] * 3;
        const double* v = l.pv + (size_t)q * 6;
        double chi = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) { const double er = s[k] + v[k]; chi += er * (v[3 + k] * er); }
        rsum += chi;
        csum += chi;
    }
    robust_chi = w3_sum(rsum);
    plain_chi = w3_sum(csum);
}

template <int JAC>
__global__ void __launch_bounds__(64) wave3_lm_kernel(const WindowArgs a) {
    const int lane = threadIdx.x;
    const long long inst = blockIdx.x;
    const WindowCaps& cp = a.caps;
    const W3Lds l = w3_carve(cp);
    const int nvm = cp.nv_max;
    const int nv = a.counts[inst * 4 + 0], nr = a.counts[inst * 4 + 1], np = a.counts[inst * 4 + 2];
    const double* gin = a.poses_in + (size_t)inst * nvm * 12;
    double* gout = a.poses + (size_t)inst * nvm * 12;
    // ---- set-up: translations, edges (with their fixed endpoints), priors into LDS; the poses' incidence lists ---------------
    double rot[9];   // this lane's pose keeps its rotation (it never moves)
#pragma unroll
    for (int k = 0; k < 9; ++k) rot[k] = 0.0;
    if (lane < nv) {
#pragma unroll
        for (int k = 0; k < 9; ++k) rot[k] = gin[lane * 12 + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) l.T[lane * 3 + k] = gin[lane * 12 + 9 + k];
    }
    {
        const int32_t* ridx = a.r_idx + (size_t)inst * cp.nr_max * 2;
        const double* rval = a.r_val + (size_t)inst * cp.nr_max * 5;
        for (int e = lane; e < nr; e += 64) {
            const int v0 = ridx[2 * e], v1 = ridx[2 * e + 1];
            l.eidx[2 * e] = v0; l.eidx[2 * e + 1] = v1;
            l.ev[2 * e] = rval[5 * e]; l.ev[2 * e + 1] = rval[5 * e + 1];
            if (v1 < 0) {
                const double* an = a.anchors + (size_t)(-1 - v1) * 3;
                l.fix[3 * e] = an[0]; l.fix[3 * e + 1] = an[1]; l.fix[3 * e + 2] = an[2];
            }
        }
        const int32_t* pidx = a.p_idx + (size_t)inst * cp.np_max;
        const double* pval = a.p_val + (size_t)inst * cp.np_max * 18;
        for (int q = lane; q < np; q += 64) {
            l.pidx[q] = pidx[q];
#pragma unroll
            for (int k = 0; k < 3; ++k) { l.pv[6 * q + k] = pval[18 * q + 9 + k]; l.pv[6 * q + 3 + k] = pval[18 * q + 12 + k]; }
        }
    }
    w3_sync();
    int deg = 0, nbin = 0;
    for (int e = 0; e < nr; ++e) {
        const int v0 = l.eidx[2 * e], v1 = l.eidx[2 * e + 1];
        deg += (v0 == lane) + (v1 == lane);
        nbin += (v1 >= 0 && (v0 > v1 ? v0 : v1) == lane);
    }
    int lst = 0;   // first entry of this pose's list: exclusive prefix sum of deg over the lanes
    {
        int incl = deg;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        lst = incl - deg;
    }
    {
        int k = lst;
        for (int e = 0; e < nr; ++e) {
            const int v0 = l.eidx[2 * e], v1 = l.eidx[2 * e + 1];
            if (v0 == lane) l.list[k++] = (e << 2) | ((v1 == lane - 1 && v1 >= 0) ? 2 : 0);
            if (v1 == lane) l.list[k++] = (e << 2) | (v0 == lane - 1 ? 2 : 0) | 1;
        }
    }
    const int shared_edges = (int)w3_sum(lane < nv && nbin >= 2 ? (double)nbin : 0.0);
    w3_sync();

    // ---- Levenberg-Marquardt (g2o: OptimizationAlgorithmLevenberg::solve), wave-uniform control flow ---------------
    constexpr double tau = 1e-5, good_lo = 1.0 / 3.0, good_hi = 2.0 / 3.0;
    constexpr int max_trials = 10;
    double lambda = 0.0, ni = 2.0, cur_chi = 0.0, last_plain = 0.0;
    int it = 0, q = 0, trials = 0, terminated = 0, buf = 0;
    bool need_lin = true;
    bool done = nv <= 0 || nr + np <= 0 || a.iterations <= 0;
    double D[6], b[3], O[9], X[3] = {0.0, 0.0, 0.0};   // this pose's H_pp, b_p, H_p,p-1 (entry (r, c) at 3 c + r), x_p
#pragma unroll
    for (int k = 0; k < 6; ++k) D[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) b[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) O[k] = 0.0;
    while (!done) {
        if (need_lin) {
            double plain;
            w3_edges<true, JAC>(l, nvm, nr, np, buf, lane, cur_chi, plain);
            last_plain = plain;
            w3_sync();
            // pose p: its edges' blocks in creation order
#pragma unroll
            for (int k = 0; k < 6; ++k) D[k] = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) b[k] = 0.0;
#pragma unroll
            for (int k = 0; k < 9; ++k) O[k] = 0.0;
            if (lane < nv) {
                const int k1 = lst + deg;
                for (int k = lst; k < k1; ++k) {
                    const int code = l.list[k];
                    const double* r = l.rec + (size_t)(code >> 2) * 8;
                    const bool second = code & 1;
                    const double wr = r[0], wre = r[1];
                    double Jm[3], Jo[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) { const double j0 = r[2 + c], j1 = r[5 + c]; Jm[c] = second ? j1 : j0; Jo[c] = second ? j0 : j1; }
                    const double wj[3] = {wr * Jm[0], wr * Jm[1], wr * Jm[2]};
#pragma unroll
                    for (int rr = 0; rr < 3; ++rr) {
#pragma unroll
                        for (int cc = 0; cc <= rr; ++cc) D[rr * (rr + 1) / 2 + cc] = __builtin_fma(wj[rr], Jm[cc], D[rr * (rr + 1) / 2 + cc]);
                        b[rr] = __builtin_fma(Jm[rr], wre, b[rr]);
                    }
                    if (code & 2) {
#pragma unroll
                        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                            for (int cc = 0; cc < 3; ++cc) O[3 * cc + rr] = __builtin_fma(wj[rr], Jo[cc], O[3 * cc + rr]);
                    }
                }
                for (int pq = 0; pq < np; ++pq) {
                    if (l.pidx[pq] != lane) continue;
                    const double* v = l.pv + (size_t)pq * 6;
                    const double* s = l.T + ((size_t)buf * nvm + lane) * 3;
#pragma unroll
                    for (int k = 0; k < 3; ++k) { const double er = s[k] + v[k]; D[k * (k + 1) / 2 + k] += v[3 + k]; b[k] += -v[3 + k] * er; }
                }
            }
            if (it == 0) {
                const double md = fmax(fmax(fabs(D[0]), fabs(D[2])), fabs(D[5]));
                lambda = tau * w3_max(lane < nv ? md : 0.0);
                ni = 2.0;
            }
            q = 0;
            need_lin = false;
        }
        // ---- (H + lambda I) x = b: forward sweep, pose p on lane p, its factor handed on through SGPRs ---------------------------
        double g10 = 0.0, g20 = 0.0, g21 = 0.0, ig[3] = {0.0, 0.0, 0.0}, y[3] = {0.0, 0.0, 0.0};   // this pose's factor (strict lower part, inverse pivots), y
        double pg10 = 0.0, pg20 = 0.0, pg21 = 0.0, pig[3] = {0.0, 0.0, 0.0}, py[3] = {0.0, 0.0, 0.0};   // the previous pose's, wave-uniform
        bool ok = true;
        for (int p = 0; p < nv; ++p) {
            if (lane == p) {
                double A[3][3], rhs[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int c = 0; c <= r; ++c) A[r][c] = D[r * (r + 1) / 2 + c];
                    A[r][r] += lambda;
                    rhs[r] = b[r];
                }
                if (p > 0) {
                    // row by row: w = row r of W = H_p,p-1 G_{p-1}^-T; S -= w w^T; rhs_r -= w . y_{p-1}
                    double Wm[9];
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        double w0 = O[r], w1 = O[3 + r], w2 = O[6 + r];
                        w0 *= pig[0];
                        w1 = __builtin_fma(-w0, pg10, w1);
                        w2 = __builtin_fma(-w0, pg20, w2);
                        w1 *= pig[1];
                        w2 = __builtin_fma(-w1, pg21, w2);
                        w2 *= pig[2];
                        Wm[r] = w0; Wm[3 + r] = w1; Wm[6 + r] = w2;
                        double acc = rhs[r];
                        acc = __builtin_fma(-w0, py[0], acc);
                        acc = __builtin_fma(-w1, py[1], acc);
                        acc = __builtin_fma(-w2, py[2], acc);
                        rhs[r] = acc;
                    }
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c2 = 0; c2 <= r; ++c2) {
                            double s2 = A[r][c2];
#pragma unroll
                            for (int k = 0; k < 3; ++k) s2 = __builtin_fma(-Wm[3 * k + r], Wm[3 * k + c2], s2);
                            A[r][c2] = s2;
                        }
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const double g = w3_pivot_rsqrt(A[j][j]);
                    ig[j] = g;
#pragma unroll
                    for (int i2 = j + 1; i2 < 3; ++i2) A[i2][j] *= g;
#pragma unroll
                    for (int i2 = j + 1; i2 < 3; ++i2)
#pragma unroll
                        for (int cc = j + 1; cc <= i2; ++cc) A[i2][cc] = __builtin_fma(-A[i2][j], A[cc][j], A[i2][cc]);
                }
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    rhs[cc] *= ig[cc];
#pragma unroll
                    for (int c2 = cc + 1; c2 < 3; ++c2) rhs[c2] = __builtin_fma(-rhs[cc], A[c2][cc], rhs[c2]);
                }
                g10 = A[1][0]; g20 = A[2][0]; g21 = A[2][1];
#pragma unroll
                for (int r = 0; r < 3; ++r) y[r] = rhs[r];
            }
            pg10 = w3_bcast(g10, p); pg20 = w3_bcast(g20, p); pg21 = w3_bcast(g21, p);
#pragma unroll
            for (int r = 0; r < 3; ++r) { pig[r] = w3_bcast(ig[r], p); py[r] = w3_bcast(y[r], p); }
            ok = ok && ((pig[0] + pig[1]) + pig[2] < DBL_MAX);
        }
        // back-substitution, x only written when every pivot was positive and finite (g2o leaves its x alone when the factorisation
        // fails, and LM applies that stale x all the same: SURVEY A.6)
        if (ok) {
            double v[3] = {0.0, 0.0, 0.0};   // H_{p+1,p}^T x_{p+1}, wave-uniform
            for (int p = nv - 1; p >= 0; --p) {
                double vn[3] = {0.0, 0.0, 0.0};
                if (lane == p) {
                    double t[3] = {y[0], y[1], y[2]};
                    // z = G_p^-1 v (forward substitution), t -= z
                    double z0 = v[0] * ig[0];
                    double z1 = __builtin_fma(-z0, g10, v[1]);
                    double z2 = __builtin_fma(-z0, g20, v[2]);
                    t[0] -= z0;
                    z1 *= ig[1];
                    z2 = __builtin_fma(-z1, g21, z2);
                    t[1] -= z1;
                    z2 *= ig[2];
                    t[2] -= z2;
                    X[2] = t[2] * ig[2];
                    t[0] = __builtin_fma(-g20, X[2], t[0]);
                    t[1] = __builtin_fma(-g21, X[2], t[1]);
                    X[1] = t[1] * ig[1];
                    t[0] = __builtin_fma(-g10, X[1], t[0]);
                    X[0] = t[0] * ig[0];
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) {
                        double acc = 0.0;
#pragma unroll
                        for (int r = 0; r < 3; ++r) acc = __builtin_fma(O[3 * cc + r], X[r], acc);
                        vn[cc] = acc;
                    }
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) v[k] = w3_bcast(vn[k], p);
            }
        }
        ++trials;
        // the trial state t + x (VertexSE3::oplus with R = I) and g2o's computeScale sum
        double sc = 0.0;
        if (lane < nv) {
            const double* s = l.T + ((size_t)buf * nvm + lane) * 3;
            double* d = l.T + ((size_t)(1 - buf) * nvm + lane) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) { sc += X[k] * (lambda * X[k] + b[k]); d[k] = s[k] + X[k]; }
        }
        const double scale = w3_sum(sc) + 1e-3;
        w3_sync();
        double temp_chi, plain2;
        w3_edges<false, JAC>(l, nvm, nr, np, 1 - buf, lane, temp_chi, plain2);
        last_plain = plain2;
        if (!ok) temp_chi = DBL_MAX;
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
    if (lane < nv) {
        const double* s = l.T + ((size_t)buf * nvm + lane) * 3;
#pragma unroll
        for (int k = 0; k < 9; ++k) gout[lane * 12 + k] = rot[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) gout[lane * 12 + 9 + k] = s[k];
    }
    if (lane == 0) {
        double* res = a.result + (size_t)inst * 8;
        res[0] = last_plain; res[1] = cur_chi; res[2] = lambda; res[3] = (double)it; res[4] = (double)trials;
        res[5] = (double)terminated; res[6] = (double)shared_edges; res[7] = nv > 0 ? (double)(nv * 65536 + 2 * nv - 1) : 0.0;
    }
}

}  // namespace

size_t window_wave3_lds_bytes(const WindowCaps& c) {
    const size_t doubles = (size_t)2 * c.nv_max * 3 + (size_t)c.nr_max * 13 + (size_t)c.np_max * 6;
    const size_t ints = (size_t)c.nr_max * 4 + c.np_max;
    return doubles * sizeof(double) + ((ints + 1) & ~(size_t)1) * sizeof(int);
}

hipError_t launch_window_wave3(const WindowArgs& a, hipStream_t stream) {
    if (a.B <= 0 || a.caps.nv_max > 64 || a.caps.nv_max <= 0) return hipErrorInvalidValue;
    const size_t lds = window_wave3_lds_bytes(a.caps);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    if (a.jacobian) hipLaunchKernelGGL((wave3_lm_kernel<1>), dim3((unsigned)a.B), dim3(64), lds, stream, a);
    else hipLaunchKernelGGL((wave3_lm_kernel<0>), dim3((unsigned)a.B), dim3(64), lds, stream, a);
    return hipGetLastError();
}

}  // namespace locamd
