// gfx950 (MI355X / CDNA4): TRANSLATION-ONLY chain windows, one LANE per window — the exact 3-DoF reduction of the reference's
// own sliding window (cfg/uwb_only.yaml) for large batches.
//
// What it solves (reference file:line): the graph Localization::addRangeEdge builds per range message
// (localization.cpp:297-376) — per pose one EdgeSE3Range to an anchor (:331) and the zero-range smoothness edge to the previous
// pose (:338-340), Cauchy kernels (:608-627) — solved by Localization::solve() = g2o Levenberg-Marquardt (:164-170), chi2() (:197).
//
// Why 3-DoF is EXACT here (SURVEY.md §8(a) note; types_edge_se3range.cpp:105-114): the residual is
// e = d - ||(X0 O0).t - (X1 O1).t||.  With identity antenna offsets (the default, localization.h:170; the example bag's case)
// the rotation of a pose does not enter e at all: the analytic rotation Jacobian is 0, and g2o's numeric one is EXACTLY 0 too
// (both perturbed evaluations compute the same number).  The rotation rows of b are 0, LM adds lambda to their diagonal, so the
// rotation part of every step is 0/lambda = 0: R never changes, computeLambdaInit's max runs over the translation diagonals
// (the rotation ones are 0), computeScale's rotation terms are 0 (lambda 0 + 0).  With R = I (Robot::init, robot.cpp:47: the poses
// start from /uwb/nodesPos with identity rotation and nothing ever turns them) VertexSE3's t <- t + R dt is t + dt, so the
// 3x3-block system below reproduces the 6x6-block one operation for operation: every dropped term is an exact zero.
// The host takes this kernel only when that holds for the whole batch (capi_window.cpp: chain3_eligible): chain topology, no
// EdgeSE3, every lever arm zero, every rotation the identity, priors (if any) with translation-only diagonal information and
// an identity measurement rotation (addLidarEdge's z prior on such a pose, localization.cpp:476-486).
//
// MI355X mapping.  The 6-DoF lane-per-window kernel (window_kernel.hip: chain_lm_kernel) is bound by HBM: 31 GB of workspace
// traffic per 65 536 ten-pose windows.  Here a pose is 3 numbers and a block 3x3: the per-trial state shrinks from ~190 to ~36
// doubles per pose and lane (~7 GB per 65 536 windows).  Everything is laid out [entry][lane] (every load of the wave is one
// 512-byte line; a wave's ds_read_b64 is 512 contiguous bytes: conflict-free), in an HBM workspace or — for the factor (G, y) and /
// or the two translation buffers (state / trial state) — in LDS, as far as that still lets every wave of the batch be resident at
// once (launch_window_chain3: the kernel is bound by memory latency first, one wave per SIMD, so residency beats LDS; 65 536
// ten-pose windows: translations in LDS 1.71 ms = 3.8e7 windows/s, all in HBM 1.98 ms, (G, y) + translations in LDS — two waves
// per CU, two rounds — 2.33 ms; 16 384 windows fit with everything in LDS: 0.99 ms against 1.30 ms from HBM).  H (6 + 3 per
// pose), the coupling block of each consecutive pair (rank-1 (w J_p) J_{p-1}^T as two 3-vectors when one edge joins the pair,
// else 9 numbers) and the packed edges are requested one pose ahead.  Tried and dropped: scoring the trial state inside the
// back-substitution (two dependent sweeps per trial instead of three) with H requested two poses ahead — 144 bytes of scratch
// spills and no gain (1.77 ms): at 1 024 resident waves the kernel moves ~4 TB/s and is bound by that, not by the sweep count.
#include "window_kernel.h"
#include "device_math.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>

#include <atomic>

namespace locamd {

namespace chain3w {
constexpr int HD = 0, HB = 6, HO = 9, X = 18, NB = 21;   // per pose, always in the HBM workspace
constexpr int G = 0, Y = 6, NGY = 9;                      // per pose: factor (3 strict-lower entries + 3 inverse pivots), y
constexpr int NT = 6;                                      // per pose: two translation buffers
constexpr int ER = 3, EP = 7;                              // per range edge: packed endpoints, measurement, information; per prior: pose, Z^-1 t, information
constexpr long long V1_BIAS = 1 << 19, V0_SHIFT = 20;
}

// doubles of HBM workspace per window
__host__ __device__ inline size_t chain3_window_doubles(const WindowCaps& c) {
    return (size_t)c.nv_max * (chain3w::NB + chain3w::NGY + chain3w::NT) + (size_t)c.nr_max * chain3w::ER + (size_t)c.np_max * chain3w::EP;
}
size_t window_chain3_workspace_doubles(const WindowCaps& c, long long B) {
    return (size_t)((B + 63) / 64) * 64 * chain3_window_doubles(c);
}

namespace {

extern __shared__ double lds3[];

__device__ __forceinline__ double pivot_rsqrt3(double d) {   // window_kernel.hip: pivot_rsqrt
    const double y = __builtin_amdgcn_rsq(d);
    const double t = d * y;
    const double e = __builtin_fma(-t, y, 1.0);
    const double pq = __builtin_fma(0.375, e, 0.5);
    const double ye = y * e;
    return __builtin_fma(ye, pq, y);
}

#pragma clang fp contract(off)
// ||d|| the way a plain CPU build of computeError evaluates it (numeric_jacobian.h: range_error_plain with a zero lever arm)
__device__ __forceinline__ double sq3_plain(double dx, double dy, double dz) { return dx * dx + dy * dy + dz * dz; }
// g2o's central difference of e = meas - ||p0 - p1|| along axis D of endpoint `which`'s translation (numeric_jacobian.h /
// window_kernel.hip: range_jac_numeric with R = I and a zero lever arm: X * fromVectorMQT(+-delta e_D) = (I, t +- delta e_D))
// NEAR: the perturbed norms from the central one n0 (device_math.h: sqrt_ieee_near_c — the same correctly rounded numbers)
template <int D, bool NEAR>
__device__ __forceinline__ double range_jac_numeric3(const double* p0, const double* p1, int which, double meas, double n0, double h0) {
    constexpr double delta = 1e-9;
    constexpr double scalar = 1.0 / (2 * delta);
    double a[3] = {p0[0], p0[1], p0[2]}, b[3] = {p1[0], p1[1], p1[2]}, am[3] = {p0[0], p0[1], p0[2]}, bm[3] = {p1[0], p1[1], p1[2]};
    if (which == 0) { a[D] = delta + p0[D]; am[D] = -delta + p0[D]; }
    else { b[D] = delta + p1[D]; bm[D] = -delta + p1[D]; }
    const double xp = sq3_plain(a[0] - b[0], a[1] - b[1], a[2] - b[2]), xm = sq3_plain(am[0] - bm[0], am[1] - bm[1], am[2] - bm[2]);
    const double ep = meas - (NEAR ? sqrt_ieee_near_c(xp, n0, h0, (h0 * h0) * (h0 + h0)) : sqrt_ieee_unscaled(xp));
    const double em = meas - (NEAR ? sqrt_ieee_near_c(xm, n0, h0, (h0 * h0) * (h0 + h0)) : sqrt_ieee_unscaled(xm));
    double bak = ep;
    bak -= em;
    return scalar * bak;
}
#pragma clang fp contract(fast)

// where a window's arrays live: H, coupling blocks, x, edges in the HBM slab; (G, y) and the translations in LDS or in the slab
struct Ctx3 {
    double* slab;      // this lane's column of the wave's [entry][lane] workspace
    size_t gy_off;     // entry offsets inside the slab (used when the array is not in LDS)
    size_t t_off;
    size_t e_off, p_off;
    int lane;
    int lds_t;         // first LDS entry of the translation buffers
};

template <bool LG> __device__ __forceinline__ double& gy_ref(const Ctx3& c, int p, int k) {
    if (LG) return lds3[(size_t)(p * chain3w::NGY + k) * 64 + c.lane];
    return c.slab[(c.gy_off + (size_t)p * chain3w::NGY + k) * 64];
}
template <bool LT> __device__ __forceinline__ double& t_ref(const Ctx3& c, int p, int buf, int k) {
    if (LT) return lds3[(size_t)(c.lds_t + p * chain3w::NT + 3 * buf + k) * 64 + c.lane];
    return c.slab[(c.t_off + (size_t)p * chain3w::NT + 3 * buf + k) * 64];
}
#define S3(p, f) c.slab[((size_t)(p) * chain3w::NB + (f)) * 64]
#define E3(e, k) c.slab[(c.e_off + (size_t)(e) * chain3w::ER + (k)) * 64]
#define P3(e, k) c.slab[(c.p_off + (size_t)(e) * chain3w::EP + (k)) * 64]

// one sweep over the window's edges in pose order: chi sums always; FULL: H, b and the coupling blocks as well
template <bool FULL, int JAC, bool LT>
__device__ __forceinline__ void chain3_sweep(const WindowArgs& a, const Ctx3& c, int nv, int nr, int np, int buf,
                                             double& robust_chi, double& plain_chi, double& max_diag, unsigned long long& ho_kind, int& shared_edges) {
    using namespace chain3w;
    double rsum = 0.0, csum = 0.0, md = 0.0;
    double Dp[9], Dc[9], O[9], tp[3] = {0.0, 0.0, 0.0}, tc[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < 9; ++k) { Dp[k] = 0.0; Dc[k] = 0.0; O[k] = 0.0; }
    // ho_kind: two bits per pose (0 no coupling, 1 rank-1 as two vectors, 2 full block); poses from 32 on always store the block
    unsigned long long kinds = 0;
    int nbin = 0, nshared = 0;
    double fu[3] = {0.0, 0.0, 0.0}, fv[3] = {0.0, 0.0, 0.0};
    int e = 0, q = 0;
    double ne[3], nT[3], npv = -1.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) ne[k] = nr > 0 ? E3(0, k) : 0.0;
    if (np > 0) npv = P3(0, 0);
#pragma unroll
    for (int k = 0; k < 3; ++k) nT[k] = nv > 0 ? t_ref<LT>(c, 0, buf, k) : 0.0;
    for (int p = 0; p < nv; ++p) {
#pragma unroll
        for (int k = 0; k < 3; ++k) tc[k] = nT[k];
        if (p + 1 < nv) {
#pragma unroll
            for (int k = 0; k < 3; ++k) nT[k] = t_ref<LT>(c, p + 1, buf, k);
        }
        if (FULL) {
#pragma unroll
            for (int k = 0; k < 9; ++k) { Dc[k] = 0.0; O[k] = 0.0; }
            nbin = 0;
        }
        // range edges whose later pose is p
        while (e < nr) {
            const long long key = (long long)ne[0];
            const int v0 = (int)(key >> V0_SHIFT), v1 = (int)(key & ((1ll << V0_SHIFT) - 1)) - (int)V1_BIAS;
            if ((v1 > v0 ? v1 : v0) != p) break;
            const double meas = ne[1], info = ne[2];
            if (e + 1 < nr) {
#pragma unroll
                for (int k = 0; k < 3; ++k) ne[k] = E3(e + 1, k);
            }
            const bool first_is_cur = v0 == p;   // endpoint 0 is pose p (else p - 1)
            double p0[3], p1[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) p0[k] = first_is_cur ? tc[k] : tp[k];
            if (v1 >= 0) {
#pragma unroll
                for (int k = 0; k < 3; ++k) p1[k] = first_is_cur ? tp[k] : tc[k];
            } else {
                const double* an = a.anchors + (size_t)(-1 - v1) * 3;
                p1[0] = an[0]; p1[1] = an[1]; p1[2] = an[2];
            }
            double u[3] = {p0[0] - p1[0], p0[1] - p1[1], p0[2] - p1[2]};
            double n = 0.0, x0 = 0.0, h0 = 0.0;
            if (JAC == 0) n = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
            else { x0 = sq3_plain(u[0], u[1], u[2]); n = sqrt_ieee_unscaled_h(x0, h0); }
            const double err = meas - n;
            const double chi = err * (info * err);
            const double aux = 1.0 + chi;
            rsum += fast_log_ge1(aux);
            csum += chi;
            if (FULL) {
                double J0[3], J1[3];
                if (JAC == 0) {
                    const double inv = n > 0.0 ? 1.0 / n : 0.0;   // coincident endpoints: J = 0, what the central difference gives
                    u[0] *= inv; u[1] *= inv; u[2] *= inv;
                    J0[0] = -u[0]; J0[1] = -u[1]; J0[2] = -u[2];
                    if (v1 >= 0) { J1[0] = u[0]; J1[1] = u[1]; J1[2] = u[2]; } else { J1[0] = 0; J1[1] = 0; J1[2] = 0; }
                } else {
                    J1[0] = 0; J1[1] = 0; J1[2] = 0;
                    if (x0 >= 1e-5 && x0 < 1e300) {   // endpoints more than ~3 mm apart
                        J0[0] = range_jac_numeric3<0, true>(p0, p1, 0, meas, n, h0);
                        J0[1] = range_jac_numeric3<1, true>(p0, p1, 0, meas, n, h0);
                        J0[2] = range_jac_numeric3<2, true>(p0, p1, 0, meas, n, h0);
                        if (v1 >= 0) {
                            J1[0] = range_jac_numeric3<0, true>(p0, p1, 1, meas, n, h0);
                            J1[1] = range_jac_numeric3<1, true>(p0, p1, 1, meas, n, h0);
                            J1[2] = range_jac_numeric3<2, true>(p0, p1, 1, meas, n, h0);
                        }
                    } else {
                        J0[0] = range_jac_numeric3<0, false>(p0, p1, 0, meas, n, h0);
                        J0[1] = range_jac_numeric3<1, false>(p0, p1, 0, meas, n, h0);
                        J0[2] = range_jac_numeric3<2, false>(p0, p1, 0, meas, n, h0);
                        if (v1 >= 0) {
                            J1[0] = range_jac_numeric3<0, false>(p0, p1, 1, meas, n, h0);
                            J1[1] = range_jac_numeric3<1, false>(p0, p1, 1, meas, n, h0);
                            J1[2] = range_jac_numeric3<2, false>(p0, p1, 1, meas, n, h0);
                        }
                    }
                }
                const double wr = info / aux, wre = -wr * err;
                // endpoint 0's block and b: into pose p (Dc) or pose p - 1 (Dp)
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int cc = 0; cc <= r; ++cc) {
                        const double h = wr * J0[r] * J0[cc];
                        Dc[r * (r + 1) / 2 + cc] += first_is_cur ? h : 0.0;
                        Dp[r * (r + 1) / 2 + cc] += first_is_cur ? 0.0 : h;
                    }
                    const double bb = J0[r] * wre;
                    Dc[6 + r] += first_is_cur ? bb : 0.0;
                    Dp[6 + r] += first_is_cur ? 0.0 : bb;
                }
                if (v1 >= 0) {
                    // endpoint 1: into the OTHER pose; coupling block rows = pose p, columns = p - 1
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
#pragma unroll
                        for (int cc = 0; cc <= r; ++cc) {
                            const double h = wr * J1[r] * J1[cc];
                            Dc[r * (r + 1) / 2 + cc] += first_is_cur ? 0.0 : h;
                            Dp[r * (r + 1) / 2 + cc] += first_is_cur ? h : 0.0;
                        }
                        const double bb = J1[r] * wre;
                        Dc[6 + r] += first_is_cur ? 0.0 : bb;
                        Dp[6 + r] += first_is_cur ? bb : 0.0;
                    }
                    double jr[3], jc[3];
#pragma unroll
                    for (int r = 0; r < 3; ++r) { jr[r] = first_is_cur ? J0[r] : J1[r]; jc[r] = first_is_cur ? J1[r] : J0[r]; }
                    if (nbin == 0 && p < 32) {
#pragma unroll
                        for (int r = 0; r < 3; ++r) { fu[r] = wr * jr[r]; fv[r] = jc[r]; }
                    } else {
                        if (nbin == 1 && p < 32) {   // a second edge on the pair: expand the first
#pragma unroll
                            for (int r = 0; r < 3; ++r)
#pragma unroll
                                for (int cc = 0; cc < 3; ++cc) O[3 * cc + r] = fu[r] * fv[cc];
                        }
#pragma unroll
                        for (int r = 0; r < 3; ++r)
#pragma unroll
                            for (int cc = 0; cc < 3; ++cc) O[3 * cc + r] += wr * jr[r] * jc[cc];
                    }
                    ++nbin;
                }
            }
            ++e;
        }
        // unary priors on pose p: e = t + Z^-1.t (identity rotations), diagonal information on the translation, no robust kernel
        while (q < np && (int)npv == p) {
            double chi = 0.0, er[3], wd[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { er[k] = tc[k] + P3(q, 1 + k); wd[k] = P3(q, 4 + k); }
#pragma unroll
            for (int k = 0; k < 3; ++k) chi += er[k] * (wd[k] * er[k]);
            rsum += chi;
            csum += chi;
            if (FULL) {
#pragma unroll
                for (int r = 0; r < 3; ++r) { Dc[r * (r + 1) / 2 + r] += wd[r]; Dc[6 + r] += -wd[r] * er[r]; }
            }
            ++q;
            if (q < np) npv = P3(q, 0);
        }
        if (FULL) {
            if (p > 0) {
#pragma unroll
                for (int k = 0; k < 6; ++k) S3(p - 1, HD + k) = Dp[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) S3(p - 1, HB + k) = Dp[6 + k];
#pragma unroll
                for (int r = 0; r < 3; ++r) md = fmax(md, fabs(Dp[r * (r + 1) / 2 + r]));
                if (nbin == 1 && p < 32) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) { S3(p, HO + k) = fu[k]; S3(p, HO + 3 + k) = fv[k]; }
                    kinds |= 1ull << (2 * p);
                } else if (nbin >= 1 || p >= 32) {   // (poses from 32 on are always read as full blocks: zeros when uncoupled)
#pragma unroll
                    for (int k = 0; k < 9; ++k) S3(p, HO + k) = O[k];
                    if (p < 32) kinds |= 2ull << (2 * p);
                }
                nshared += nbin >= 2 ? nbin : 0;
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) Dp[k] = Dc[k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) tp[k] = tc[k];
    }
    if (FULL && nv > 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) S3(nv - 1, HD + k) = Dp[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) S3(nv - 1, HB + k) = Dp[6 + k];
#pragma unroll
        for (int r = 0; r < 3; ++r) md = fmax(md, fabs(Dp[r * (r + 1) / 2 + r]));
    }
    robust_chi = rsum; plain_chi = csum; max_diag = md;
    if (FULL) { ho_kind = kinds; shared_edges = nshared; }
}

// pose p of the trial state: t + dx (VertexSE3::oplus with R = I), read from buffer `buf`, written to the other one; returns the
// pose's share of g2o's computeScale sum
template <bool LT>
__device__ __forceinline__ double chain3_apply_step(const Ctx3& c, int p, int buf, const double* dx, double lambda) {
    using namespace chain3w;
    double sc = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) sc += dx[k] * (lambda * dx[k] + S3(p, HB + k));
#pragma unroll
    for (int k = 0; k < 3; ++k) t_ref<LT>(c, p, 1 - buf, k) = t_ref<LT>(c, p, buf, k) + dx[k];
    return sc;
}

// (H + lambda I) x = b for the block-tridiagonal H (3x3 blocks): forward sweep (Cholesky + forward substitution), then the
// back-substitution with the step applied pose by pose.  x is only written when every pivot was positive and finite (g2o leaves
// its x alone when the factorisation fails, and LM applies that stale x all the same: SURVEY A.6).
template <bool LG, bool LT>
__device__ __forceinline__ bool chain3_factor_solve(const Ctx3& c, int nv, double lambda, int buf, unsigned long long ho_kind, double& scale_sum) {
    using namespace chain3w;
    scale_sum = 0.0;
    auto kind_of = [&](int p) { return p < 32 ? (int)((ho_kind >> (2 * p)) & 3ull) : (p > 0 ? 2 : 0); };
    bool ok = true;
    double g10 = 0.0, g20 = 0.0, g21 = 0.0, igp[3] = {0.0, 0.0, 0.0}, yp[3] = {0.0, 0.0, 0.0};   // the previous pose's factor, inverse pivots, y
    double nHd[6], nHb[3], nHo[9];
#pragma unroll
    for (int k = 0; k < 6; ++k) nHd[k] = nv > 0 ? S3(0, HD + k) : 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) nHb[k] = nv > 0 ? S3(0, HB + k) : 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) nHo[k] = 0.0;
    for (int p = 0; p < nv; ++p) {
        double A[3][3], rhs[3], Ho[9];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int cc = 0; cc <= r; ++cc) A[r][cc] = nHd[r * (r + 1) / 2 + cc];
            A[r][r] += lambda;
            rhs[r] = nHb[r];
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) Ho[k] = nHo[k];
        const int kd = kind_of(p);
        if (kd == 1) {   // rank-1: u v^T
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) Ho[3 * cc + r] = nHo[r] * nHo[3 + cc];
        }
        if (p + 1 < nv) {
#pragma unroll
            for (int k = 0; k < 6; ++k) nHd[k] = S3(p + 1, HD + k);
#pragma unroll
            for (int k = 0; k < 3; ++k) nHb[k] = S3(p + 1, HB + k);
            const int kn = kind_of(p + 1);
            if (kn == 1) {
#pragma unroll
                for (int k = 0; k < 6; ++k) nHo[k] = S3(p + 1, HO + k);
            } else if (kn == 2) {
#pragma unroll
                for (int k = 0; k < 9; ++k) nHo[k] = S3(p + 1, HO + k);
            }
        }
        if (kd != 0) {
            // row by row: w = row r of W = H_p,p-1 G_{p-1}^-T; S -= w w^T; rhs_r -= w . y_{p-1}
            double Wm[9];   // W, entry (r, c) at 3 c + r
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                double w0 = Ho[r], w1 = Ho[3 + r], w2 = Ho[6 + r];
                w0 *= igp[0];
                w1 = __builtin_fma(-w0, g10, w1);
                w2 = __builtin_fma(-w0, g20, w2);
                w1 *= igp[1];
                w2 = __builtin_fma(-w1, g21, w2);
                w2 *= igp[2];
                Wm[r] = w0; Wm[3 + r] = w1; Wm[6 + r] = w2;
                double acc = rhs[r];
                acc = __builtin_fma(-w0, yp[0], acc);
                acc = __builtin_fma(-w1, yp[1], acc);
                acc = __builtin_fma(-w2, yp[2], acc);
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
        double ig[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double g = pivot_rsqrt3(A[j][j]);
            ig[j] = g;
#pragma unroll
            for (int i2 = j + 1; i2 < 3; ++i2) A[i2][j] *= g;
#pragma unroll
            for (int i2 = j + 1; i2 < 3; ++i2)
#pragma unroll
                for (int cc = j + 1; cc <= i2; ++cc) A[i2][cc] = __builtin_fma(-A[i2][j], A[cc][j], A[i2][cc]);
        }
        ok = ok && ((ig[0] + ig[1]) + ig[2] < DBL_MAX);
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            rhs[cc] *= ig[cc];
#pragma unroll
            for (int c2 = cc + 1; c2 < 3; ++c2) rhs[c2] = __builtin_fma(-rhs[cc], A[c2][cc], rhs[c2]);
        }
        gy_ref<LG>(c, p, G + 0) = A[1][0]; gy_ref<LG>(c, p, G + 1) = A[2][0]; gy_ref<LG>(c, p, G + 2) = A[2][1];
#pragma unroll
        for (int r = 0; r < 3; ++r) { gy_ref<LG>(c, p, G + 3 + r) = ig[r]; gy_ref<LG>(c, p, Y + r) = rhs[r]; igp[r] = ig[r]; yp[r] = rhs[r]; }
        g10 = A[1][0]; g20 = A[2][0]; g21 = A[2][1];
    }
    if (!ok) {
        for (int p = 0; p < nv; ++p) {
            double dx[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) dx[k] = S3(p, X + k);
            scale_sum += chain3_apply_step<LT>(c, p, buf, dx, lambda);
        }
        return false;
    }
    double xn[3] = {0.0, 0.0, 0.0};
    // x_p = G_p^-T (y_p - W_{p+1}^T x_{p+1}) with W_{p+1}^T x = G_p^-1 (H_{p+1,p}^T x): W itself is never stored
    double nW[9];   // H_{p+1,p}, requested one pose ahead
#pragma unroll
    for (int k = 0; k < 9; ++k) nW[k] = 0.0;
    for (int p = nv - 1; p >= 0; --p) {
        double t[3], ig[3], Wn[9];
        const double l10 = gy_ref<LG>(c, p, G + 0), l20 = gy_ref<LG>(c, p, G + 1), l21 = gy_ref<LG>(c, p, G + 2);
#pragma unroll
        for (int r = 0; r < 3; ++r) { t[r] = gy_ref<LG>(c, p, Y + r); ig[r] = gy_ref<LG>(c, p, G + 3 + r); }
#pragma unroll
        for (int k = 0; k < 9; ++k) Wn[k] = nW[k];   // coupling block of pose p + 1
        const int kup = p < nv - 1 ? kind_of(p + 1) : 0;
        {
            const int kme = kind_of(p);   // (pose p's coupling block is what pose p - 1 needs next)
            if (kme == 1) {
#pragma unroll
                for (int k = 0; k < 6; ++k) nW[k] = S3(p, HO + k);
            } else if (kme == 2) {
#pragma unroll
                for (int k = 0; k < 9; ++k) nW[k] = S3(p, HO + k);
            }
        }
        if (kup != 0) {
            double v[3];
            if (kup == 1) {   // (u v^T)^T x = v (u . x)
                double sx = 0.0;
#pragma unroll
                for (int r = 0; r < 3; ++r) sx = __builtin_fma(Wn[r], xn[r], sx);
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) v[cc] = Wn[3 + cc] * sx;
            } else {
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    double acc = 0.0;
#pragma unroll
                    for (int r = 0; r < 3; ++r) acc = __builtin_fma(Wn[3 * cc + r], xn[r], acc);
                    v[cc] = acc;
                }
            }
            // z = G_p^-1 v (forward substitution), t -= z
            v[0] *= ig[0];
            v[1] = __builtin_fma(-v[0], l10, v[1]);
            v[2] = __builtin_fma(-v[0], l20, v[2]);
            t[0] -= v[0];
            v[1] *= ig[1];
            v[2] = __builtin_fma(-v[1], l21, v[2]);
            t[1] -= v[1];
            v[2] *= ig[2];
            t[2] -= v[2];
        }
        xn[2] = t[2] * ig[2];
        t[0] = __builtin_fma(-l20, xn[2], t[0]);
        t[1] = __builtin_fma(-l21, xn[2], t[1]);
        xn[1] = t[1] * ig[1];
        t[0] = __builtin_fma(-l10, xn[1], t[0]);
        xn[0] = t[0] * ig[0];
#pragma unroll
        for (int r = 0; r < 3; ++r) S3(p, X + r) = xn[r];
        scale_sum += chain3_apply_step<LT>(c, p, buf, xn, lambda);
    }
    return true;
}

template <int JAC, bool LG, bool LT>
__global__ void __launch_bounds__(64, 1) chain3_lm_kernel(const WindowArgs a, double* ws) {
    using namespace chain3w;
    const int lane = threadIdx.x;
    const long long inst = (long long)blockIdx.x * 64 + lane;
    const bool live = inst < a.B;
    const WindowCaps& cp = a.caps;
    Ctx3 c;
    c.lane = lane;
    c.slab = ws + (size_t)blockIdx.x * 64 * chain3_window_doubles(cp) + lane;
    c.gy_off = (size_t)cp.nv_max * NB;
    c.t_off = c.gy_off + (size_t)cp.nv_max * NGY;
    c.e_off = c.t_off + (size_t)cp.nv_max * NT;
    c.p_off = c.e_off + (size_t)cp.nr_max * ER;
    c.lds_t = LG ? cp.nv_max * NGY : 0;
    int nv = 0, nr = 0, np = 0;
    if (live) { nv = a.counts[inst * 4 + 0]; nr = a.counts[inst * 4 + 1]; np = a.counts[inst * 4 + 2]; }
    const double* gin = a.poses_in + (size_t)(live ? inst : 0) * cp.nv_max * 12;
    double* gout = a.poses + (size_t)(live ? inst : 0) * cp.nv_max * 12;
    for (int p = 0; p < nv; ++p) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { t_ref<LT>(c, p, 0, k) = gin[p * 12 + 9 + k]; S3(p, X + k) = 0.0; }   // x of a fresh optimize() call
    }
    {
        const int32_t* ridx = a.r_idx + (size_t)(live ? inst : 0) * cp.nr_max * 2;
        const double* rval = a.r_val + (size_t)(live ? inst : 0) * cp.nr_max * 5;
        for (int e = 0; e < nr; ++e) {
            E3(e, 0) = (double)(((long long)ridx[2 * e] << V0_SHIFT) + ((long long)ridx[2 * e + 1] + V1_BIAS));
            E3(e, 1) = rval[5 * e];
            E3(e, 2) = rval[5 * e + 1];
        }
        const int32_t* pidx = a.p_idx + (size_t)(live ? inst : 0) * cp.np_max;
        const double* pval = a.p_val + (size_t)(live ? inst : 0) * cp.np_max * 18;
        for (int e = 0; e < np; ++e) {
            P3(e, 0) = (double)pidx[e];
#pragma unroll
            for (int k = 0; k < 3; ++k) { P3(e, 1 + k) = pval[18 * e + 9 + k]; P3(e, 4 + k) = pval[18 * e + 12 + k]; }
        }
    }
    constexpr double tau = 1e-5, good_lo = 1.0 / 3.0, good_hi = 2.0 / 3.0;
    constexpr int max_trials = 10;
    double lambda = 0.0, ni = 2.0, cur_chi = 0.0, last_plain = 0.0;
    int it = 0, q = 0, trials = 0, terminated = 0, buf = 0, shared_edges = 0;
    unsigned long long ho_kind = 0;
    bool need_lin = true;
    bool done = !live || nv <= 0 || nr + np <= 0 || a.iterations <= 0;
    while (__ballot(!done)) {
        if (!done) {
            if (need_lin) {
                double plain, md;
                chain3_sweep<true, JAC, LT>(a, c, nv, nr, np, buf, cur_chi, plain, md, ho_kind, shared_edges);
                last_plain = plain;
                if (it == 0) { lambda = tau * md; ni = 2.0; }
                q = 0;
                need_lin = false;
            }
            double sc;
            const bool ok2 = chain3_factor_solve<LG, LT>(c, nv, lambda, buf, ho_kind, sc);
            ++trials;
            double temp_chi, plain2, md2;
            unsigned long long unused_kind;
            int unused_shared;
            chain3_sweep<false, JAC, LT>(a, c, nv, nr, np, 1 - buf, temp_chi, plain2, md2, unused_kind, unused_shared);
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
            for (int k = 0; k < 9; ++k) gout[p * 12 + k] = gin[p * 12 + k];   // the rotations never move (identity)
#pragma unroll
            for (int k = 0; k < 3; ++k) gout[p * 12 + 9 + k] = t_ref<LT>(c, p, buf, k);
        }
        double* res = a.result + (size_t)inst * 8;
        res[0] = last_plain; res[1] = cur_chi; res[2] = lambda; res[3] = (double)it; res[4] = (double)trials;
        res[5] = (double)terminated; res[6] = (double)shared_edges; res[7] = nv > 0 ? (double)(nv * 65536 + 2 * nv - 1) : 0.0;
    }
}
#undef S3
#undef E3
#undef P3

template <int JAC, bool LG, bool LT>
hipError_t launch_chain3_t(const WindowArgs& a, double* ws, size_t lds, hipStream_t stream) {
    static std::atomic<uint64_t> attr_set{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_set.load(std::memory_order_acquire) & bit)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&chain3_lm_kernel<JAC, LG, LT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
        if (e != hipSuccess) return e;
        attr_set.fetch_or(bit, std::memory_order_release);
    }
    const unsigned blocks = (unsigned)((a.B + 63) / 64);
    hipLaunchKernelGGL((chain3_lm_kernel<JAC, LG, LT>), dim3(blocks), dim3(64), lds, stream, a, ws);
    return hipGetLastError();
}

}  // namespace

// LDS plan.  The kernel is bound by memory latency (one wave per SIMD, nothing else to hide a round trip behind), so what matters
// first is that EVERY wave of the batch is resident at once; LDS comes second.  Candidates, richest first: (G, y) + translations
// (15 doubles per pose and lane), (G, y) alone (9), the translations alone (6), nothing.  The first one whose footprint still lets
// all ceil(B / 64) waves onto the chip's CUs at once is taken (measured on 65 536 ten-pose windows, 1 024 waves on 256 CUs:
// everything in HBM 1.99 ms; (G, y) + translations in LDS — two waves per CU, so two rounds — 2.33 ms; (G, y) alone, three per
// CU, 2.81 ms).  mode bits: 1 = (G, y) in LDS, 2 = translations in LDS.
int window_chain3_lds_mode(const WindowCaps& c, long long B, int n_cus) {
    const long long waves = (B + 63) / 64;
    const size_t lds_cu = 160 * 1024;
    const int cand[4] = {3, 1, 2, 0};
    for (int m : cand) {
        const size_t per_pose = (m & 1 ? chain3w::NGY : 0) + (m & 2 ? chain3w::NT : 0);
        const size_t bytes = (size_t)c.nv_max * per_pose * 64 * sizeof(double);
        if (bytes > lds_cu - 1024) continue;
        const long long per_cu = bytes ? (long long)(lds_cu / (bytes + 256)) : 8;   // (8: two waves per SIMD at this kernel's register count)
        if ((per_cu < 8 ? per_cu : 8) * n_cus >= waves) return m;
    }
    return 0;
}

hipError_t launch_window_chain3(const WindowArgs& a, double* ws, hipStream_t stream) {
    if (a.B <= 0 || !ws || a.caps.ns_max < 0) return hipErrorInvalidValue;
    static std::atomic<int> n_cus{0};
    if (!n_cus.load()) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        n_cus.store(n);
    }
    int mode = window_chain3_lds_mode(a.caps, a.B, n_cus.load());
    if (const char* v = getenv("LOCAMD_CHAIN3_LDS")) mode = atoi(v) & 3;   // A/B runs (the footprint must fit: checked below)
    const size_t per_pose = (mode & 1 ? chain3w::NGY : 0) + (mode & 2 ? chain3w::NT : 0);
    const size_t lds = (size_t)a.caps.nv_max * per_pose * 64 * sizeof(double);
    if (lds > 160 * 1024 - 1024) return hipErrorInvalidValue;
    const int sel = (a.jacobian ? 4 : 0) + mode;
    switch (sel) {
        case 0: return launch_chain3_t<0, false, false>(a, ws, lds, stream);
        case 1: return launch_chain3_t<0, true, false>(a, ws, lds, stream);
        case 2: return launch_chain3_t<0, false, true>(a, ws, lds, stream);
        case 3: return launch_chain3_t<0, true, true>(a, ws, lds, stream);
        case 4: return launch_chain3_t<1, false, false>(a, ws, lds, stream);
        case 5: return launch_chain3_t<1, true, false>(a, ws, lds, stream);
        case 6: return launch_chain3_t<1, false, true>(a, ws, lds, stream);
        default: return launch_chain3_t<1, true, true>(a, ws, lds, stream);
    }
}

}  // namespace locamd
