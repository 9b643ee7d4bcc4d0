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
//     broadcast.
//   * ONE wave-uniform loop of "passes".  A pass = [3x3 solve -> trial point -> residuals AND normal equations at the
//     trial point -> per-lane LM bookkeeping].  The nested g2o loops (outer iterations x LM trials) become per-lane
//     state (iteration, trial, lambda, nu); an accepted trial's (H, b) is the next iteration's system (g2o recomputes
//     exactly those numbers), a rejected trial only bumps lambda.  The first evaluation of an update (at the prior,
//     which also applies the outlier gate) is just another pass with a zero step.
//   * Lanes are NOT synchronised per epoch: a tag that finishes its update stores its result and moves on to its next
//     epoch while its wave-mates are still iterating (LM trial counts differ per tag; synchronising per epoch costs
//     max-over-wave instead of mean: +22 % at 32 tags/wave, +39 % at 64).  A wave keeps a two-epoch window: the next
//     epoch's ranges sit converted in registers (a transition is a handful of register moves), the one after is in
//     flight from HBM as raw floats; float->(measurement, information) conversion runs once per epoch for the whole
//     wave in lockstep, when the slowest lane leaves the window's first epoch.
//   * sum_m log(1 + chi_m) is evaluated as log(prod_m (1 + chi_m)): one f64 log per pass instead of M.
//     1/x and (sqrt x, 1/sqrt x) come from v_rcp_f64 / v_rsq_f64 + FMA refinement instead of the IEEE division/sqrt
//     expansions (operands are ranges, 1 + chi, pivots: always normal); ranges keep the builtin sqrt's 1.1e-16 bound.
//   * HBM layout: ranges as float4 tiles [K][M4][B] (16 B per lane, unit stride over tags); position state double
//     [3][B] stays in registers across the K epochs of a launch; outputs are SoA doubles.
//   * Every workgroup does identical, independent work on a private slice of B, and the only shared data is the
//     192-byte anchor table, so there is no L2 reuse to steer: blockIdx -> XCD mapping is left to the dispatcher.
#include "snapshot_kernel.h"
#include "device_math.h"

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

// g2o's numeric Jacobian evaluates e(p +- delta e_d) with delta = 1e-9; keep those few operations un-contracted and
// on the IEEE sqrt so the difference quotient sees (nearly) the same roundings as a CPU build.
#pragma clang fp contract(off)
__device__ __forceinline__ double range_norm_plain(double dx, double dy, double dz) {
    return sqrt_ieee_unscaled(dx * dx + dy * dy + dz * dz);
}
// the squared distance in the same (plain) operation order
__device__ __forceinline__ double range_sq_plain(double dx, double dy, double dz) { return dx * dx + dy * dy + dz * dz; }
#pragma clang fp contract(fast)

// Residuals, robust weights, normal equations and both chi sums at point p, for this lane's APL anchors, combined
// over the LPI lanes of the tag.  `gate`: finite only on the first evaluation of an update (at the prior estimate) with
// the outlier gate armed: ranges with | ||p - a|| - d | > gate lose their weight for the whole update.
template <int LPI, int APL, int JAC>
__device__ __forceinline__ System evaluate(const double px, const double py, const double pz,
                                           const double (&ax)[APL], const double (&ay)[APL], const double (&az)[APL],
                                           const double (&d)[APL], double (&w)[APL], const double gate) {
    System s = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    double prod = 1.0;
    int kexp = 0;
#pragma unroll
    for (int j = 0; j < APL; ++j) {
        const double dx = px - ax[j], dy = py - ay[j], dz = pz - az[j];
        double n, jx, jy, jz;  // J = de/dp
        if constexpr (JAC == 0) {
            // coincident endpoints: n2 = 0 is clamped, dx = dy = dz = 0 -> J = 0
            const double n2 = fmax(dx * dx + dy * dy + dz * dz, 1e-300);
            double inv;
            sqrt_and_rsqrt_fast(n2, n, inv);
            jx = -dx * inv; jy = -dy * inv; jz = -dz * inv;
        } else {
            constexpr double delta = 1e-9;
            constexpr double scalar = 1.0 / (2 * delta);
            const double x0 = range_sq_plain(dx, dy, dz);
            double h0, c0;
            n = sqrt_ieee_unscaled_raw(x0, h0, c0);   // (no +-0 / inf pass-through here: only the slow branch below can see such an argument)
            const double xp = (px + delta) - ax[j], xm = (px - delta) - ax[j];
            const double yp = (py + delta) - ay[j], ym = (py - delta) - ay[j];
            const double zp = (pz + delta) - az[j], zm = (pz - delta) - az[j];
            if (x0 >= 1e-5 && x0 < 1e300) {
                // the six perturbed norms from the central one (device_math.h: sqrt_ieee_near_c — the same correctly rounded numbers)
                jx = scalar * ((d[j] - sqrt_ieee_near_c(range_sq_plain(xp, dy, dz), n, h0, c0)) - (d[j] - sqrt_ieee_near_c(range_sq_plain(xm, dy, dz), n, h0, c0)));
                jy = scalar * ((d[j] - sqrt_ieee_near_c(range_sq_plain(dx, yp, dz), n, h0, c0)) - (d[j] - sqrt_ieee_near_c(range_sq_plain(dx, ym, dz), n, h0, c0)));
                jz = scalar * ((d[j] - sqrt_ieee_near_c(range_sq_plain(dx, dy, zp), n, h0, c0)) - (d[j] - sqrt_ieee_near_c(range_sq_plain(dx, dy, zm), n, h0, c0)));
            } else {   // a tag within millimetres of an anchor (or an estimate that has run away)
                n = (x0 == 0.0 || x0 == __builtin_inf()) ? x0 : n;
                jx = scalar * ((d[j] - range_norm_plain(xp, dy, dz)) - (d[j] - range_norm_plain(xm, dy, dz)));
                jy = scalar * ((d[j] - range_norm_plain(dx, yp, dz)) - (d[j] - range_norm_plain(dx, ym, dz)));
                jz = scalar * ((d[j] - range_norm_plain(dx, dy, zp)) - (d[j] - range_norm_plain(dx, dy, zm)));
            }
        }
        const double e = d[j] - n;
        w[j] = (fabs(e) > gate) ? 0.0 : w[j];  // |d_hat - d| > distance_outlier, localization.cpp:309 (gate = inf: off)
        const double we = w[j] * e;
        const double chi = e * we;
        const double aux = 1.0 + chi;
        const double rho1 = fast_rcp_1nr(aux);
        const double wr = rho1 * w[j];
        const double wjx = wr * jx, wjy = wr * jy, wjz = wr * jz;
        s.h00 = __builtin_fma(wjx, jx, s.h00); s.h01 = __builtin_fma(wjx, jy, s.h01); s.h02 = __builtin_fma(wjx, jz, s.h02);
        s.h11 = __builtin_fma(wjy, jy, s.h11); s.h12 = __builtin_fma(wjy, jz, s.h12); s.h22 = __builtin_fma(wjz, jz, s.h22);
        s.b0 = __builtin_fma(-wjx, e, s.b0); s.b1 = __builtin_fma(-wjy, e, s.b1); s.b2 = __builtin_fma(-wjz, e, s.b2);
        prod *= aux;
        if (APL > 4 && j == 3) { kexp = __builtin_amdgcn_frexp_exp(prod); prod = __builtin_amdgcn_frexp_mant(prod); }
        s.chi += chi;
    }
    s.h00 = group_sum<LPI>(s.h00); s.h01 = group_sum<LPI>(s.h01); s.h02 = group_sum<LPI>(s.h02);
    s.h11 = group_sum<LPI>(s.h11); s.h12 = group_sum<LPI>(s.h12); s.h22 = group_sum<LPI>(s.h22);
    s.b0 = group_sum<LPI>(s.b0); s.b1 = group_sum<LPI>(s.b1); s.b2 = group_sum<LPI>(s.b2);
    s.chi = group_sum<LPI>(s.chi);
    // Sum_j log(1 + chi_j) as ONE log of the product, kept as (mantissa, exponent) with the exponent taken out after every four
    // factors: the product of four factors overflows only beyond |e| / sigma ~ 1e38 each — g2o's edge-by-edge sum of logs stays
    // finite there too, but float32 distance_err and ranges (the wire format) cannot produce it.
    if constexpr (LPI > 1) {
        kexp += __builtin_amdgcn_frexp_exp(prod);
        prod = __builtin_amdgcn_frexp_mant(prod);
        prod = group_prod<LPI>(prod);
        kexp = (int)group_sum<LPI>((double)kexp);
    }
    s.rchi = fast_log_ge1_scaled(prod, kexp);
    return s;
}

// (H + lambda I) x = b by LDL^T; false if not positive definite (g2o: Cholesky failure).
__device__ __forceinline__ bool solve3(const System& s, double lambda, double& x0, double& x1, double& x2) {
    const double d0 = s.h00 + lambda, a11 = s.h11 + lambda, a22 = s.h22 + lambda;
    const double i0 = fast_rcp(d0);
    const double l10 = s.h01 * i0, l20 = s.h02 * i0;
    const double d1 = __builtin_fma(-l10, s.h01, a11);
    const double i1 = fast_rcp(d1);
    const double t21 = __builtin_fma(-l20, s.h01, s.h12);
    const double l21 = t21 * i1;
    const double d2 = __builtin_fma(-l21, t21, __builtin_fma(-l20, s.h02, a22));
    const double i2 = fast_rcp(d2);
    const bool ok = (d0 > 0.0) && (d1 > 0.0) && (d2 > 0.0) && (d2 < DBL_MAX);
    const double y0 = s.b0;
    const double y1 = __builtin_fma(-l10, y0, s.b1);
    const double y2 = __builtin_fma(-l21, y1, __builtin_fma(-l20, y0, s.b2));
    x2 = y2 * i2;
    x1 = __builtin_fma(-l21, x2, y1 * i1);
    x0 = __builtin_fma(-l20, x2, __builtin_fma(-l10, x1, y0 * i0));
    return ok;
}

// float ranges of one epoch for this lane (16 / 8 / 4 B per lane, unit stride over tags)
template <int APL>
__device__ __forceinline__ void load_epoch(const SnapshotArgs& a, long long k, int m0, long long inst,
                                           float (&df)[APL], float (&sf)[APL]) {
    const long long B = a.B;
    if constexpr (APL % 4 == 0) {
#pragma unroll
        for (int q4 = 0; q4 < APL / 4; ++q4) {
            const long long idx = (k * a.M4 + (m0 / 4 + q4)) * B + inst;
            const float4 dv = reinterpret_cast<const float4*>(a.dist)[idx];
            const float4 sv = reinterpret_cast<const float4*>(a.err)[idx];
            df[4 * q4 + 0] = dv.x; df[4 * q4 + 1] = dv.y; df[4 * q4 + 2] = dv.z; df[4 * q4 + 3] = dv.w;
            sf[4 * q4 + 0] = sv.x; sf[4 * q4 + 1] = sv.y; sf[4 * q4 + 2] = sv.z; sf[4 * q4 + 3] = sv.w;
        }
    } else if constexpr (APL == 2) {
        const long long idx = ((k * a.M4 + m0 / 4) * B + inst) * 2 + (m0 % 4) / 2;
        const float2 dv = reinterpret_cast<const float2*>(a.dist)[idx];
        const float2 sv = reinterpret_cast<const float2*>(a.err)[idx];
        df[0] = dv.x; df[1] = dv.y; sf[0] = sv.x; sf[1] = sv.y;
    } else {
        static_assert(APL == 1 || APL == 2 || APL % 4 == 0, "anchors per lane");
        const long long idx = ((k * a.M4 + m0 / 4) * B + inst) * 4 + (m0 % 4);
        df[0] = a.dist[idx]; sf[0] = a.err[idx];
    }
}
// cost definition: measurement as double, information 1/err^2 (localization.cpp:318, :615-620); unusable slots get 0
template <int APL>
__device__ __forceinline__ void ingest(const float (&df)[APL], const float (&sf)[APL], double (&d)[APL], double (&w)[APL]) {
#pragma unroll
    for (int j = 0; j < APL; ++j) {
        const double dd = (double)df[j], ss = (double)sf[j];
        const bool valid = (ss > 0.0) && (ss < DBL_MAX) && (fabs(dd) < DBL_MAX);
        const double cov = valid ? ss * ss : 1.0;
        const double mask = valid ? 1.0 : 0.0;
        d[j] = valid ? dd : 0.0;
        w[j] = fast_rcp(cov) * mask;  // (multiply, not select: keeps the reciprocal out of a divergent branch)
    }
}

// LDS holds what a lane touches once per epoch, so the per-pass registers stay under the 256 architectural VGPRs:
//   s_next [APL][256] double2 : (measurement, information) of the window's second epoch, per lane
//   s_raw  [2*APL/4][256] float4 : raw float tiles of the epoch after that, landed by LDS-DMA (global_load_lds_dwordx4:
//            no VGPR destination, so an in-flight prefetch costs no registers and no loop-carried copies)
template <int APL> struct SnapshotLds {
    static constexpr int NEXT_BYTES = APL * 256 * 16;
    static constexpr int RAW_BYTES = (APL % 4 == 0) ? (2 * (APL / 4) * 256 * 16) : 0;
    static constexpr int BYTES = NEXT_BYTES + RAW_BYTES;
};

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

template <int LPI, int APL, int JAC>
__global__ void __launch_bounds__(256) snapshot_lm_kernel(const SnapshotArgs a) {
    __shared__ __attribute__((aligned(16))) char lds_bytes[SnapshotLds<APL>::BYTES];
    double2* const s_next = reinterpret_cast<double2*>(lds_bytes);
    float4* const s_raw = reinterpret_cast<float4*>(lds_bytes + SnapshotLds<APL>::NEXT_BYTES);
    const int tib = threadIdx.x;
    const long long tid = (long long)blockIdx.x * blockDim.x + tib;
    const long long inst = tid / LPI;
    const int g = (int)(tid % LPI);
    const long long B = a.B;
    const int K = a.K;
    // all LPI lanes of a tag share `inst`, so a group is live or dead as a whole
    const bool live = (inst < B) && K > 0;
    const long long ld_inst = live ? inst : 0;  // dead lanes load tag 0's ranges (in bounds) and never store
    bool exhausted = !live;

    // anchors of this lane (uniform -> SGPRs when LPI == 1)
    double ax[APL], ay[APL], az[APL];
    const int m0 = g * APL;
#pragma unroll
    for (int j = 0; j < APL; ++j) {
        ax[j] = a.anchors[(m0 + j) * 3 + 0];
        ay[j] = a.anchors[(m0 + j) * 3 + 1];
        az[j] = a.anchors[(m0 + j) * 3 + 2];
    }

    constexpr double tau = 1e-5, good_lo = 1.0 / 3.0, good_hi = 2.0 / 3.0;
    constexpr int max_trials = 10;

    // Epoch window [c, c+1]: every live lane of the wave works on epoch c or c+1.
    //   d, w   (registers): this lane's current epoch (k)
    //   s_next (LDS)      : epoch c+1, already converted to (measurement, information) for ALL lanes
    //   s_raw  (LDS)      : raw float ranges of epoch c+2, in flight from HBM
    // A lane that finishes epoch c takes its s_next entry; one that finishes c+1 early waits until the slowest lane
    // has left epoch c, at which point the whole wave converts c+2 in lockstep and prefetches c+3.
    auto put_next = [&](const double (&dd)[APL], const double (&ww)[APL]) {
#pragma unroll
        for (int j = 0; j < APL; ++j) s_next[j * 256 + tib] = make_double2(dd[j], ww[j]);
    };
    // Hand-issued ds_read_b128: hipcc cannot tell s_next from s_raw and would put an s_waitcnt vmcnt(0) (the LDS-DMA
    // and every store in flight) in front of compiler-generated LDS reads here, on every pass; s_next is only ever
    // written by this lane's own ds_write, so lgkmcnt is the only counter that matters.
    auto take_next = [&](double (&dd)[APL], double (&ww)[APL]) {
        typedef double v2f64 __attribute__((ext_vector_type(2)));
        const unsigned addr = (unsigned)(unsigned long long)(s_next + tib);
        v2f64 v[APL];
#pragma unroll
        for (int j = 0; j < APL; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[j]) : "v"(addr), "n"(j * 4096));
        if constexpr (APL == 8) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
        } else {
#pragma unroll
            for (int j = 0; j < APL; ++j) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[j]) :: "memory");
        }
#pragma unroll
        for (int j = 0; j < APL; ++j) { dd[j] = v[j].x; ww[j] = v[j].y; }
    };
    // start the fetch of epoch `ke` (no-op for lane mappings that cannot use 16-byte LDS-DMA: they load at consume time)
    auto prefetch = [&](int ke) {
        if constexpr (APL % 4 == 0) {
            const int wave_base = (tib / 64) * 64;
#pragma unroll
            for (int q4 = 0; q4 < APL / 4; ++q4) {
                const long long idx = ((long long)ke * a.M4 + (m0 / 4 + q4)) * B + ld_inst;
                __builtin_amdgcn_global_load_lds((glb_void_ptr)(reinterpret_cast<const float4*>(a.dist) + idx),
                                                 (lds_void_ptr)(s_raw + q4 * 256 + wave_base), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((glb_void_ptr)(reinterpret_cast<const float4*>(a.err) + idx),
                                                 (lds_void_ptr)(s_raw + (APL / 4 + q4) * 256 + wave_base), 16, 0, 0);
            }
        }
    };
    auto consume = [&](int ke, double (&dd)[APL], double (&ww)[APL]) {
        float df[APL], sf[APL];
        if constexpr (APL % 4 == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the DMA of epoch ke has landed (issued ~10 passes ago)
#pragma unroll
            for (int q4 = 0; q4 < APL / 4; ++q4) {
                const float4 dv = s_raw[q4 * 256 + tib];
                const float4 sv = s_raw[(APL / 4 + q4) * 256 + tib];
                df[4 * q4 + 0] = dv.x; df[4 * q4 + 1] = dv.y; df[4 * q4 + 2] = dv.z; df[4 * q4 + 3] = dv.w;
                sf[4 * q4 + 0] = sv.x; sf[4 * q4 + 1] = sv.y; sf[4 * q4 + 2] = sv.z; sf[4 * q4 + 3] = sv.w;
            }
        } else {
            load_epoch<APL>(a, ke, m0, ld_inst, df, sf);
        }
        ingest<APL>(df, sf, dd, ww);
    };

    double px = 0, py = 0, pz = 0;
    double d[APL], w[APL];
    if (live) { px = a.pos[0 * B + inst]; py = a.pos[1 * B + inst]; pz = a.pos[2 * B + inst]; }
    {
        float df0[APL], sf0[APL];
        load_epoch<APL>(a, 0, m0, ld_inst, df0, sf0);
        ingest<APL>(df0, sf0, d, w);
        if (K > 1) {
            double dn[APL], wn[APL];
            load_epoch<APL>(a, 1, m0, ld_inst, df0, sf0);
            ingest<APL>(df0, sf0, dn, wn);
            put_next(dn, wn);
        }
        if (K > 2) prefetch(2);
    }
    // the prior's loads retire here, not at their first use inside the loop (where the wait would sit on every pass)
    asm volatile("" : "+v"(px), "+v"(py), "+v"(pz));

    System cur = {1, 0, 0, 1, 0, 1, 0, 0, 0, 0, 0};
    double cur_chi = 0, last_chi = 0, lambda = 1.0, ni = 2.0;
    int c = 0;          // wave-uniform window base
    int k = 0;          // this lane's epoch: c or c+1
    int it = 0, q = 0, trials = 0;
    bool init = true;   // next pass is the first evaluation of this epoch's update
    bool waiting = false;

    while (__any(!exhausted)) {
        const bool active = !exhausted && !waiting;
        // ---- step: (H + lambda I) x = b; first pass of an update evaluates the prior itself -------------------
        double x0, x1, x2;
        const bool ok2 = solve3(cur, lambda, x0, x1, x2) && !init;
        if (!ok2) { x0 = 0.0; x1 = 0.0; x2 = 0.0; }
        const double tx = px + x0, ty = py + x1, tz = pz + x2;  // oplus: t += R dt with R = I
        // outlier gate armed only on the first pass of an update, after warm-up: an infinite threshold elsewhere
        const bool gate_now = active && init && (a.gate > 0.0) && (k >= a.gate_from_epoch);
        const System tr = evaluate<LPI, APL, JAC>(tx, ty, tz, ax, ay, az, d, w, gate_now ? a.gate : DBL_MAX);

        // ---- per-lane LM bookkeeping (g2o OptimizationAlgorithmLevenberg::solve + SparseOptimizer::optimize) ---
        // Written as selects on per-lane predicates (one exec region only, for the 11-double system copy): a lone wave
        // per SIMD pays a full issue slot for every branch-management instruction.
        const bool first = active && init;   // this pass evaluated the prior: computeActiveErrors + buildSystem, iteration 0
        const bool lm = active && !init;     // this pass was an LM trial
        const double temp_chi = ok2 ? tr.rchi : DBL_MAX;
        double scale = x0 * __builtin_fma(lambda, x0, cur.b0) + x1 * __builtin_fma(lambda, x1, cur.b1) +
                       x2 * __builtin_fma(lambda, x2, cur.b2);  // computeScale
        scale += 1e-3;
        const double rho = (cur_chi - temp_chi) * fast_rcp(scale);
        const bool accept = lm && (rho > 0.0) && (fabs(temp_chi) < DBL_MAX) && ok2;
        if (first || accept) {  // the trial's system is the next iteration's system (discardTop); first: it IS the system
            cur = tr;
            cur_chi = tr.rchi;
            px = tx; py = ty; pz = tz;  // (first: tx == px)
        }
        const double r21 = 2.0 * rho - 1.0;
        const double lam_acc = lambda * fmax(good_lo, fmin(1.0 - r21 * r21 * r21, good_hi));
        const double lam_first = tau * fmax(fabs(tr.h00), fmax(fabs(tr.h11), fabs(tr.h22)));  // computeLambdaInit
        lambda = first ? lam_first : (accept ? lam_acc : (lm ? lambda * ni : lambda));
        ni = (first || accept) ? 2.0 : (lm ? ni * 2.0 : ni);
        last_chi = active ? tr.chi : last_chi;
        trials = first ? 0 : trials + (lm ? 1 : 0);
        q = first ? 0 : q + (lm ? 1 : 0);
        const bool end_iter = lm && !((rho < 0.0) && (q < max_trials));
        it = first ? 0 : it + (end_iter ? 1 : 0);
        // no active edge ("0 vertices to optimize") or nothing to iterate; LM: Terminate (10 failed trials or rho == 0) or done
        const bool finished = (first && ((a.iterations <= 0) || !((tr.h00 + tr.h11 + tr.h22) > 0.0))) ||
                              (end_iter && ((q == max_trials) || (rho == 0.0) || (it >= a.iterations)));
        q = end_iter ? 0 : q;
        init = active ? false : init;

        // ---- lanes that just finished an update: emit, then step into the next epoch or wait for the window ------
        if (__any(finished)) {
            if (finished) {
                if (g == 0) {
                    a.out_pos[((long long)k * 3 + 0) * B + inst] = px;
                    a.out_pos[((long long)k * 3 + 1) * B + inst] = py;
                    a.out_pos[((long long)k * 3 + 2) * B + inst] = pz;
                    a.out_chi2[(long long)k * B + inst] = last_chi;
                    if (a.out_trials) a.out_trials[(long long)k * B + inst] = (uint8_t)(trials > 255 ? 255 : trials);
                }
                if (k + 1 >= K) {
                    exhausted = true;
                    if (g == 0) { a.pos[0 * B + inst] = px; a.pos[1 * B + inst] = py; a.pos[2 * B + inst] = pz; }
                } else if (k == c) {
                    take_next(d, w);
                    k = c + 1;
                    init = true;
                } else {
                    waiting = true;  // a full epoch ahead of the slowest lane of this wave
                }
            }
        }
        // ---- window advance: nobody is left in epoch c -> convert epoch c+2 for all lanes, prefetch c+3 --------------
        if (!__any(!exhausted && k == c)) {
            ++c;
            if (c + 1 < K) {
                double dn[APL], wn[APL];
                consume(c + 1, dn, wn);
                put_next(dn, wn);
                asm volatile("" ::: "memory");  // the LDS reads of s_raw above stay ahead of the DMA that refills it
                if (c + 2 < K) prefetch(c + 2);
                if (waiting) {  // k == c now: release into epoch c+1
#pragma unroll
                    for (int j = 0; j < APL; ++j) { d[j] = dn[j]; w[j] = wn[j]; }
                    k = c + 1;
                    init = true;
                    waiting = false;
                }
            }
        }
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

namespace {
__global__ void __launch_bounds__(256) pack_kmb_kernel(const float* __restrict__ src, float4* __restrict__ dst, long long B, int M, int M4,
                                                       long long total, float pad) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // (k, g, b), b fastest: reads and stores are coalesced
    if (i >= total) return;
    const long long b = i % B, kg = i / B;
    const int g = (int)(kg % M4);
    const long long k = kg / M4;
    const float* row = src + (k * M + 4 * g) * B + b;
    float4 v;
    v.x = 4 * g + 0 < M ? row[0] : pad;
    v.y = 4 * g + 1 < M ? row[B] : pad;
    v.z = 4 * g + 2 < M ? row[2 * B] : pad;
    v.w = 4 * g + 3 < M ? row[3 * B] : pad;
    dst[i] = v;
}
}  // namespace

hipError_t launch_pack_kmb(const float* src_kmb, float* dst_tiles, long long B, int M, int M4, int K, float pad, hipStream_t stream) {
    const long long total = (long long)K * M4 * B;
    if (total <= 0 || M <= 0 || M > 4 * M4) return hipErrorInvalidValue;
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pack_kmb_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src_kmb, reinterpret_cast<float4*>(dst_tiles), B, M, M4, total, pad);
    return hipGetLastError();
}

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
