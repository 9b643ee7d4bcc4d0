// gfx950 (MI355X / CDNA4) snapshot solver kernel: fused residual + Jacobian + normal equations + LM.
//
// What it computes (reference call sites in brackets):
//   per tag and epoch, M anchor ranges -> Cauchy-robust range factors
//       e_m = d_m - ||p - a_m||                    [types_edge_se3range.cpp:105-114]
//       Omega_m = 1 / err_m^2, RobustKernelCauchy   [localization.cpp:318, :608-627]
//       gate |‖p_prior - a_m‖ - d_m| > outlier      [localization.cpp:306-313]
//   solved with g2o's Levenberg-Marquardt as run by Localization::solve()
//       initializeOptimization(); optimize(maximum_iteration)   [localization.cpp:164-170]
//   (H = sum J^T rho' Omega J, b = -sum J^T rho' Omega e, lambda0 = 1e-5 max diag,
//    +lambda I, gain ratio with the 1e-3 guard, (1/3, 2/3) clamp, nu doubling, <= 10 trials, no convergence test)
//   and emits the estimate and optimizer.chi2()               [localization.cpp:197]
//
// MI355X mapping
//   * LPI lanes cooperate on one tag (LPI = 1, 2, 4 or 8; APL = M_PAD / LPI anchors per lane).  The 9 entries of
//     (H, b) plus the two chi sums are combined with DPP butterflies (quad_perm / row_half_mirror: VALU rate,
//     no LDS); every lane of a tag then holds bit-identical sums, so the per-tag LM state machine needs no
//     broadcast.  LPI trades redundant 3x3 solves for occupancy (B = 65 536 tags is only 1 wave per SIMD at LPI 1).
//   * The nested g2o loops (outer iterations x LM trials) are flattened into one wave-uniform loop of
//     "solve -> trial point -> evaluate residuals AND normal equations there"; an accepted trial's (H, b) is the
//     next iteration's system (g2o recomputes exactly those numbers), a rejected trial only bumps lambda.
//     Lanes carry their own (iteration, trial, lambda, nu, done) state; the wave runs until all are done.
//   * sum_m log(1 + chi_m) is evaluated as log(prod_m (1 + chi_m)): one f64 log per pass instead of M.
//   * HBM layout: ranges as float4 tiles [K][M4][B] so a wave reads 16 B per lane, unit stride over tags;
//     position state double [3][B] stays in registers across the K epochs of a launch; outputs are SoA doubles.
//   * Every workgroup does identical, independent work on a private slice of B, and the only shared data is the
//     192-byte anchor table, so there is no L2 reuse to steer: blockIdx -> XCD mapping is left to the dispatcher.
#include "snapshot_kernel.h"

#include <float.h>
#include <math.h>

namespace locamd {

namespace {

// ---- DPP lane exchange for doubles -------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
constexpr int DPP_QUAD_XOR1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;  // lane i <-> 7 - i inside each group of 8

template <int LPI>
__device__ __forceinline__ double group_sum(double v) {
    if constexpr (LPI >= 2) v += dpp_move<DPP_QUAD_XOR1>(v);
    if constexpr (LPI >= 4) v += dpp_move<DPP_QUAD_XOR2>(v);
    if constexpr (LPI >= 8) v += dpp_move<DPP_ROW_HALF_MIRROR>(v);
    return v;
}
template <int LPI>
__device__ __forceinline__ double group_prod(double v) {
    if constexpr (LPI >= 2) v *= dpp_move<DPP_QUAD_XOR1>(v);
    if constexpr (LPI >= 4) v *= dpp_move<DPP_QUAD_XOR2>(v);
    if constexpr (LPI >= 8) v *= dpp_move<DPP_ROW_HALF_MIRROR>(v);
    return v;
}

struct System {
    double h00, h01, h02, h11, h12, h22;  // J^T (rho' Omega) J
    double b0, b1, b2;                    // -J^T rho' Omega e
    double rchi;                          // activeRobustChi2 = sum log(1 + chi)
    double chi;                           // sum chi (what optimizer.chi2() reports)
};

// g2o's numeric Jacobian evaluates e(p +- delta e_d) with delta = 1e-9; keep those few operations un-contracted so
// the difference quotient sees the same roundings as the CPU restatement.
#pragma clang fp contract(off)
__device__ __forceinline__ double range_norm_plain(double dx, double dy, double dz) {
    return sqrt(dx * dx + dy * dy + dz * dz);
}
#pragma clang fp contract(fast)

// Residuals, robust weights, normal equations and both chi sums at point p, for this lane's APL anchors,
// then combined over the LPI lanes of the tag.
template <int LPI, int APL, int JAC>
__device__ __forceinline__ System evaluate(const double px, const double py, const double pz,
                                           const double (&ax)[APL], const double (&ay)[APL], const double (&az)[APL],
                                           const double (&d)[APL], const double (&w)[APL]) {
    System s = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    double prod = 1.0;
#pragma unroll
    for (int j = 0; j < APL; ++j) {
        const double dx = px - ax[j], dy = py - ay[j], dz = pz - az[j];
        double n, jx, jy, jz;  // J = de/dp
        if constexpr (JAC == 0) {
            const double n2 = dx * dx + dy * dy + dz * dz;
            n = sqrt(n2);
            const double inv = n > 0.0 ? 1.0 / n : 0.0;  // coincident endpoints: J = 0
            jx = -dx * inv; jy = -dy * inv; jz = -dz * inv;
        } else {
            constexpr double delta = 1e-9;
            constexpr double scalar = 1.0 / (2 * delta);
            n = range_norm_plain(dx, dy, dz);
            const double xp = (px + delta) - ax[j], xm = (px - delta) - ax[j];
            const double yp = (py + delta) - ay[j], ym = (py - delta) - ay[j];
            const double zp = (pz + delta) - az[j], zm = (pz - delta) - az[j];
            jx = scalar * ((d[j] - range_norm_plain(xp, dy, dz)) - (d[j] - range_norm_plain(xm, dy, dz)));
            jy = scalar * ((d[j] - range_norm_plain(dx, yp, dz)) - (d[j] - range_norm_plain(dx, ym, dz)));
            jz = scalar * ((d[j] - range_norm_plain(dx, dy, zp)) - (d[j] - range_norm_plain(dx, dy, zm)));
        }
        const double e = d[j] - n;
        const double chi = e * (w[j] * e);
        const double aux = 1.0 + chi;
        const double rho1 = 1.0 / aux;
        const double wr = rho1 * w[j];
        const double wre = -wr * e;  // omega_r
        s.h00 += wr * jx * jx; s.h01 += wr * jx * jy; s.h02 += wr * jx * jz;
        s.h11 += wr * jy * jy; s.h12 += wr * jy * jz; s.h22 += wr * jz * jz;
        s.b0 += jx * wre; s.b1 += jy * wre; s.b2 += jz * wre;
        prod *= aux;
        s.chi += chi;
    }
    s.h00 = group_sum<LPI>(s.h00); s.h01 = group_sum<LPI>(s.h01); s.h02 = group_sum<LPI>(s.h02);
    s.h11 = group_sum<LPI>(s.h11); s.h12 = group_sum<LPI>(s.h12); s.h22 = group_sum<LPI>(s.h22);
    s.b0 = group_sum<LPI>(s.b0); s.b1 = group_sum<LPI>(s.b1); s.b2 = group_sum<LPI>(s.b2);
    s.chi = group_sum<LPI>(s.chi);
    s.rchi = log(group_prod<LPI>(prod));
    return s;
}

// (H + lambda I) x = b by LDL^T; false if not positive definite (g2o: Cholesky failure).
__device__ __forceinline__ bool solve3(const System& s, double lambda, double& x0, double& x1, double& x2) {
    const double a00 = s.h00 + lambda, a11 = s.h11 + lambda, a22 = s.h22 + lambda;
    const double d0 = a00;
    const double i0 = 1.0 / d0;
    const double l10 = s.h01 * i0, l20 = s.h02 * i0;
    const double d1 = a11 - l10 * s.h01;
    const double i1 = 1.0 / d1;
    const double l21 = (s.h12 - l20 * s.h01) * i1;
    const double d2 = a22 - l20 * s.h02 - l21 * (s.h12 - l20 * s.h01);
    const bool ok = (d0 > 0.0) && (d1 > 0.0) && (d2 > 0.0) && (d2 < DBL_MAX);
    const double y0 = s.b0;
    const double y1 = s.b1 - l10 * y0;
    const double y2 = s.b2 - l20 * y0 - l21 * y1;
    x2 = y2 / d2;
    x1 = y1 * i1 - l21 * x2;
    x0 = y0 * i0 - l10 * x1 - l20 * x2;
    return ok;
}

template <int LPI, int APL, int JAC>
__global__ void __launch_bounds__(256) snapshot_lm_kernel(const SnapshotArgs a) {
    constexpr int M_PAD = LPI * APL;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long inst = tid / LPI;
    const int g = (int)(tid % LPI);
    const bool live = inst < a.B;
    if (!live) inst = a.B - 1;  // keep the lane in the DPP butterflies; it stores nothing
    const long long B = a.B;

    // anchors of this lane (uniform -> SGPRs when LPI == 1)
    double ax[APL], ay[APL], az[APL];
    const int m0 = g * APL;
#pragma unroll
    for (int j = 0; j < APL; ++j) {
        ax[j] = a.anchors[(m0 + j) * 3 + 0];
        ay[j] = a.anchors[(m0 + j) * 3 + 1];
        az[j] = a.anchors[(m0 + j) * 3 + 2];
    }
    (void)M_PAD;

    double px = a.pos[0 * B + inst], py = a.pos[1 * B + inst], pz = a.pos[2 * B + inst];

    constexpr double tau = 1e-5, good_lo = 1.0 / 3.0, good_hi = 2.0 / 3.0;
    constexpr int max_trials = 10;

    for (int k = 0; k < a.K; ++k) {
        // ---- ingest this epoch's ranges: 16 B (or 8/4 B) per lane, unit stride over tags -------------
        float df[APL], sf[APL];
        if constexpr (APL % 4 == 0) {
#pragma unroll
            for (int q4 = 0; q4 < APL / 4; ++q4) {
                const long long grp = m0 / 4 + q4;
                const long long idx = ((long long)k * a.M4 + grp) * B + inst;
                const float4 dv = reinterpret_cast<const float4*>(a.dist)[idx];
                const float4 sv = reinterpret_cast<const float4*>(a.err)[idx];
                df[4 * q4 + 0] = dv.x; df[4 * q4 + 1] = dv.y; df[4 * q4 + 2] = dv.z; df[4 * q4 + 3] = dv.w;
                sf[4 * q4 + 0] = sv.x; sf[4 * q4 + 1] = sv.y; sf[4 * q4 + 2] = sv.z; sf[4 * q4 + 3] = sv.w;
            }
        } else if constexpr (APL == 2) {
            const long long idx = (((long long)k * a.M4 + m0 / 4) * B + inst) * 2 + (m0 % 4) / 2;
            const float2 dv = reinterpret_cast<const float2*>(a.dist)[idx];
            const float2 sv = reinterpret_cast<const float2*>(a.err)[idx];
            df[0] = dv.x; df[1] = dv.y; sf[0] = sv.x; sf[1] = sv.y;
        } else {
            static_assert(APL == 1 || APL == 2 || APL % 4 == 0, "anchors per lane");
            const long long idx = (((long long)k * a.M4 + m0 / 4) * B + inst) * 4 + (m0 % 4);
            df[0] = a.dist[idx]; sf[0] = a.err[idx];
        }
        // ---- cost definition: information 1/err^2, outlier gate on the prior estimate -----------------
        double d[APL], w[APL];
        double wsum = 0.0;
        const bool gate_on = (a.gate > 0.0) && (k >= a.gate_from_epoch);
#pragma unroll
        for (int j = 0; j < APL; ++j) {
            const double dd = (double)df[j], ss = (double)sf[j];
            const bool valid = (ss > 0.0) && (ss < DBL_MAX) && (fabs(dd) < DBL_MAX);
            double wj = valid ? 1.0 / (ss * ss) : 0.0;
            const double dx = px - ax[j], dy = py - ay[j], dz = pz - az[j];
            const double dhat = sqrt(dx * dx + dy * dy + dz * dz);
            if (gate_on && fabs(dhat - dd) > a.gate) wj = 0.0;
            d[j] = valid ? dd : 0.0;
            w[j] = wj;
            wsum += wj;
        }
        wsum = group_sum<LPI>(wsum);

        // ---- g2o LM, flattened ---------------------------------------------------------------------------
        System cur = evaluate<LPI, APL, JAC>(px, py, pz, ax, ay, az, d, w);
        double cur_chi = cur.rchi;
        double last_chi = cur.chi;
        double lambda = tau * fmax(fabs(cur.h00), fmax(fabs(cur.h11), fabs(cur.h22)));  // computeLambdaInit
        double ni = 2.0;
        int it = 0, q = 0, trials = 0;
        bool done = (a.iterations <= 0) || !(wsum > 0.0);  // no active edge: "0 vertices to optimize"

        while (__any(!done)) {
            double x0, x1, x2;
            const bool ok2 = solve3(cur, lambda, x0, x1, x2);
            if (!ok2) { x0 = 0.0; x1 = 0.0; x2 = 0.0; }
            const double tx = px + x0, ty = py + x1, tz = pz + x2;  // oplus: t += R dt with R = I
            const System tr = evaluate<LPI, APL, JAC>(tx, ty, tz, ax, ay, az, d, w);
            const double temp_chi = ok2 ? tr.rchi : DBL_MAX;
            double scale = x0 * (lambda * x0 + cur.b0) + x1 * (lambda * x1 + cur.b1) + x2 * (lambda * x2 + cur.b2);
            scale += 1e-3;
            const double rho = (cur_chi - temp_chi) / scale;
            const bool accept = (rho > 0.0) && (fabs(temp_chi) < DBL_MAX) && ok2;
            if (!done) {
                ++trials;
                last_chi = tr.chi;
                if (accept) {
                    const double r21 = 2.0 * rho - 1.0;
                    double alpha = 1.0 - r21 * r21 * r21;
                    alpha = fmin(alpha, good_hi);
                    lambda *= fmax(good_lo, alpha);
                    ni = 2.0;
                    cur_chi = temp_chi;
                    px = tx; py = ty; pz = tz;
                    cur = tr;
                } else {
                    lambda *= ni;
                    ni *= 2.0;
                }
                ++q;
                const bool again = (rho < 0.0) && (q < max_trials);
                if (!again) {
                    ++it;
                    if (q == max_trials || rho == 0.0 || it >= a.iterations) done = true;
                    q = 0;
                }
            }
        }

        if (live && g == 0) {
            a.out_pos[((long long)k * 3 + 0) * B + inst] = px;
            a.out_pos[((long long)k * 3 + 1) * B + inst] = py;
            a.out_pos[((long long)k * 3 + 2) * B + inst] = pz;
            a.out_chi2[(long long)k * B + inst] = last_chi;
            if (a.out_trials) a.out_trials[(long long)k * B + inst] = (uint8_t)(trials > 255 ? 255 : trials);
        }
    }
    if (live && g == 0) {
        a.pos[0 * B + inst] = px; a.pos[1 * B + inst] = py; a.pos[2 * B + inst] = pz;
    }
}

template <int LPI, int APL, int JAC>
hipError_t launch_one(const SnapshotArgs& a, int block_threads, hipStream_t stream) {
    const long long lanes = a.B * LPI;
    const long long blocks = (lanes + block_threads - 1) / block_threads;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((snapshot_lm_kernel<LPI, APL, JAC>), dim3((unsigned)blocks), dim3((unsigned)block_threads), 0, stream, a);
    return hipGetLastError();
}

template <int LPI, int APL>
hipError_t launch_jac(const SnapshotArgs& a, int jac, int block_threads, hipStream_t stream) {
    return jac == 0 ? launch_one<LPI, APL, 0>(a, block_threads, stream) : launch_one<LPI, APL, 1>(a, block_threads, stream);
}

}  // namespace

bool snapshot_supported(int m_pad, int lpi) {
    if (lpi != 1 && lpi != 2 && lpi != 4 && lpi != 8) return false;
    if (m_pad != 4 && m_pad != 8 && m_pad != 12 && m_pad != 16) return false;
    if (m_pad % lpi) return false;
    const int apl = m_pad / lpi;
    if (!(apl == 1 || apl == 2 || apl % 4 == 0)) return false;
    if (m_pad == 12 && lpi != 1) return false;  // 12 = 3 float4 groups: one lane per tag only
    return true;
}

hipError_t launch_snapshot(const SnapshotArgs& a, int m_pad, int lpi, int jac, int block_threads, hipStream_t stream) {
    if (!snapshot_supported(m_pad, lpi)) return hipErrorInvalidValue;
    if (block_threads <= 0) block_threads = 256;
    if (block_threads % 64 || block_threads > 256) return hipErrorInvalidValue;
    switch (m_pad * 100 + lpi) {
        case 401: return launch_jac<1, 4>(a, jac, block_threads, stream);
        case 402: return launch_jac<2, 2>(a, jac, block_threads, stream);
        case 404: return launch_jac<4, 1>(a, jac, block_threads, stream);
        case 801: return launch_jac<1, 8>(a, jac, block_threads, stream);
        case 802: return launch_jac<2, 4>(a, jac, block_threads, stream);
        case 804: return launch_jac<4, 2>(a, jac, block_threads, stream);
        case 808: return launch_jac<8, 1>(a, jac, block_threads, stream);
        case 1201: return launch_jac<1, 12>(a, jac, block_threads, stream);
        case 1601: return launch_jac<1, 16>(a, jac, block_threads, stream);
        case 1602: return launch_jac<2, 8>(a, jac, block_threads, stream);
        case 1604: return launch_jac<4, 4>(a, jac, block_threads, stream);
        case 1608: return launch_jac<8, 2>(a, jac, block_threads, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace locamd
