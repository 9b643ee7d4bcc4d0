// gfx950 (MI355X / CDNA4): 6-DoF chain windows, one WAVE per window — the drop-in node's own solve when the poses turn
// (cfg/uwb_imu.yaml, cfg/uwb_imu_lidar.yaml: IMU orientation priors, an antenna lever arm) and small batches of such windows.
// The 6-DoF sibling of wave3_kernel.hip (read that header first: lane = edge for residuals and Jacobians, lane = pose for the
// normal equations, the sequential block-tridiagonal Cholesky handed from lane to lane by DPP, the next LM trials' lambdas solved
// speculatively in the idle lanes).
//
// What it solves (reference file:line): the graph Localization::addRangeEdge / addImuEdge / addLidarEdge build
// (localization.cpp:297-376, 499-535, 452-497): per pose an EdgeSE3Range to an anchor with the antenna lever arm on the pose
// (types_edge_se3range.cpp:105-114), the zero-range smoothness edge to the previous pose, unary EdgeSE3Prior factors with diagonal
// information; Cauchy kernels on the ranges (:608-627); solved by g2o Levenberg-Marquardt (:164-170), chi2() (:197).
//
// The host takes it (capi_window.cpp: pick_kernel) for chain batches below the lane-per-window batch size whose windows have <= 64
// poses, no EdgeSE3 factor, no lever arm on endpoint 1 and at most ONE range edge per pair of consecutive poses: the coupling block
// of a pair is then the rank-1 product (w J_p) J_{p-1}^T = u v^T of 6-vectors, the Schur complement of pose p is A_p - alpha_p u u^T,
// and Sherman-Morrison turns the sequential elimination into a recurrence on TWO SCALARS per pose while every pose factors its own
// 6x6 block at once (see "The solve" in the kernel).  Everything else stays on window_lm_kernel.  Measured: cfg/uwb_imu.yaml's
// twelve-pose window 0.083 ms (window_lm_kernel 0.40), 0.117 ms per range message through the node (the oracle: 0.26 ms).
//
// SE3 = true (round 4): the same windows WITH an EdgeSE3 factor between consecutive poses — Localization::addTwistEdge
// (localization.cpp:438-459, 560-605; cfg/uwb_twist.yaml), at most one per pair, <= 63 poses.  The coupling block of a pair is then a full
// 6x6 (the EdgeSE3's + the range edge's rank-1 part), so the elimination is the plain block-tridiagonal Cholesky — with q the pose
// eliminated just before p (its "sender"): W_p = G_q^-1 K_p^T, S_p = A_p - W_p^T W_p = G_p G_p^T — run from BOTH ends of the chain
// towards the middle pose in the same instructions (poses 0 .. m in lanes 0 .. m, poses n - 1 .. m + 1 in lanes m + 1 .. n - 1, the middle
// pose once more in lane n) and handed from lane to lane by DPP: every repetition every lane takes the factor of the lane before it
// (21 + 6 numbers) and redoes its own step; after ceil(n / 2) repetitions both chains are final and the middle pose's two lanes add
// their Schur complements.  The EdgeSE3 between a pose and its sender is linearised by the pose's lane (its record in LDS,
// [entry][edge]), the sender's share goes down one lane by DPP.  Everything else — records, gather, speculative trials (scored two per
// pass), LM — is shared with the rank-1 kernel.  Measured: cfg/uwb_twist.yaml's 15-pose window 0.29 ms (window_lm_kernel 0.64), 0.32 ms
// per range message through the node (the oracle: 0.34 ms), 4.5e6 windows/s in batches (chain_lm_kernel: 2.7e6).
#include "se3_edge_device.h"

#include <float.h>
#include <math.h>

#include <atomic>

namespace locamd {

namespace {

extern __shared__ double w6lds[];

#ifdef LOCAMD_WAVE6_TIMING   // diagnostic build: cycle stamps per phase instead of result[0 .. 7] (never benchmarked)
#define W6_T0() unsigned long long w6_tc = __builtin_readcyclecounter(), w6_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long w6_start = w6_tc
#define W6_T(k) do { const unsigned long long n_ = __builtin_readcyclecounter(); w6_ph[k] += n_ - w6_tc; w6_tc = n_; } while (0)
#else
#define W6_T0() do {} while (0)
#define W6_T(k) do {} while (0)
#endif

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double w6_dpp(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, CTRL == 0x138 || CTRL == 0x130);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, CTRL == 0x138 || CTRL == 0x130);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double w6_bcast(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
// the value of lane - 1 / lane + 1 (wave_shr:1 / wave_shl:1 cross the 16-lane rows on gfx9; no neighbour: 0)
__device__ __forceinline__ double w6_from_prev(double v) { return w6_dpp<0x138, 0xF>(v); }
__device__ __forceinline__ double w6_from_next(double v) { return w6_dpp<0x130, 0xF>(v); }
__device__ __forceinline__ double w6_sum(double v) {
    v += w6_dpp<0x111, 0xF>(v);
    v += w6_dpp<0x112, 0xF>(v);
    v += w6_dpp<0x114, 0xF>(v);
    v += w6_dpp<0x118, 0xF>(v);
    v += w6_dpp<0x142, 0xA>(v);
    v += w6_dpp<0x143, 0xC>(v);
    return w6_bcast(v, 63);
}
__device__ __forceinline__ double w6_max(double v) {   // non-negative inputs
    v = fmax(v, w6_dpp<0x111, 0xF>(v));
    v = fmax(v, w6_dpp<0x112, 0xF>(v));
    v = fmax(v, w6_dpp<0x114, 0xF>(v));
    v = fmax(v, w6_dpp<0x118, 0xF>(v));
    v = fmax(v, w6_dpp<0x142, 0xA>(v));
    v = fmax(v, w6_dpp<0x143, 0xC>(v));
    return w6_bcast(v, 63);
}
__device__ __forceinline__ void w6_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ double w6_pivot_rsqrt(double d) {   // window_kernel.hip: pivot_rsqrt
    const double y = __builtin_amdgcn_rsq(d);
    const double t = d * y;
    const double e = __builtin_fma(-t, y, 1.0);
    const double pq = __builtin_fma(0.375, e, 0.5);
    const double ye = y * e;
    return __builtin_fma(ye, pq, y);
}

// ---- small SE3 algebra (row-major 3x3; window_kernel.hip's) ------------------------------------------------------------------
__device__ __forceinline__ void w6_mat_mul(const double* A, const double* B, double* C) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j] + A[i * 3 + 2] * B[2 * 3 + j];
}
__device__ __forceinline__ void w6_mat_vec(const double* A, const double* v, double* o) {
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = A[i * 3 + 0] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2];
}
__device__ __forceinline__ void w6_mat_tvec(const double* A, const double* v, double* o) {  // A^T v
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = A[0 * 3 + i] * v[0] + A[1 * 3 + i] * v[1] + A[2 * 3 + i] * v[2];
}
// Eigen::Quaternion(Matrix3) — q = (w, x, y, z); the square root and its reciprocal from one v_rsq_f64 seed (device_math.h:
// sqrt_and_rsqrt, 1e-16 / 4e-15 relative: the prior's error and Jacobian carry them at that level)
__device__ __forceinline__ void w6_mat_to_quat(const double* R, double* q) {
    double qw, qx, qy, qz, n, inv;
    const double t = R[0] + R[4] + R[8];
    if (t > 0) {
        sqrt_and_rsqrt(t + 1.0, n, inv);
        qw = 0.5 * n; inv *= 0.5;
        qx = (R[7] - R[5]) * inv; qy = (R[2] - R[6]) * inv; qz = (R[3] - R[1]) * inv;
    } else if (R[0] >= R[4] && R[0] >= R[8]) {
        sqrt_and_rsqrt(R[0] - R[4] - R[8] + 1.0, n, inv);
        qx = 0.5 * n; inv *= 0.5;
        qw = (R[7] - R[5]) * inv; qy = (R[3] + R[1]) * inv; qz = (R[6] + R[2]) * inv;
    } else if (R[4] > R[0] && R[4] >= R[8]) {
        sqrt_and_rsqrt(R[4] - R[8] - R[0] + 1.0, n, inv);
        qy = 0.5 * n; inv *= 0.5;
        qw = (R[2] - R[6]) * inv; qz = (R[7] + R[5]) * inv; qx = (R[1] + R[3]) * inv;
    } else {
        sqrt_and_rsqrt(R[8] - R[0] - R[4] + 1.0, n, inv);
        qz = 0.5 * n; inv *= 0.5;
        qw = (R[3] - R[1]) * inv; qx = (R[2] + R[6]) * inv; qy = (R[5] + R[7]) * inv;
    }
    q[0] = qw; q[1] = qx; q[2] = qy; q[3] = qz;
}
// g2o internal::normalize: unit norm, w >= 0
__device__ __forceinline__ void w6_quat_normalize_sign(double* q) {
    double n, s;
    sqrt_and_rsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3], n, s);
    if (q[0] < 0) s = -s;
    q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s;
}
// Eigen toRotationMatrix (no normalisation)
__device__ __forceinline__ void w6_quat_to_mat(const double* q, double* R) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

#pragma clang fp contract(off)
// the squared distance of range_error_plain (numeric_jacobian.h), same operation order: |(R off + t) - q1|^2
__device__ __forceinline__ double w6_range_sq_plain(const double* R, const double* t, const double* off, const double* q1) {
    const double px = R[0] * off[0] + R[1] * off[1] + R[2] * off[2];
    const double py = R[3] * off[0] + R[4] * off[1] + R[5] * off[2];
    const double pz = R[6] * off[0] + R[7] * off[1] + R[8] * off[2];
    const double dx = (px + t[0]) - q1[0], dy = (py + t[1]) - q1[1], dz = (pz + t[2]) - q1[2];
    return dx * dx + dy * dy + dz * dz;
}
// one column of g2o's numeric Jacobian (window_kernel.hip: range_jac_numeric; delta = 1e-9 through VertexSE3::oplus) of endpoint
// `which` (0: the pose carrying the lever arm, 1: the other pose — no lever arm there: its point is its translation).
// NEAR: the perturbed norms from the central one (device_math.h: sqrt_ieee_near_c — the same correctly rounded numbers).
template <int D, bool NEAR>
__device__ __forceinline__ double w6_jac_numeric(const double* X0, const double* off, const double* X1, const double* q1, int which, double meas,
                                                 double n0, double h0) {
    constexpr double delta = 1e-9;
    constexpr double scalar = 1.0 / (2 * delta);
    double Rp[9], tp[3], Rm[9], tm[3];
    double xp, xm;
    if (which == 0) {
        oplus_axis_plain<D>(X0, X0 + 9, delta, Rp, tp);
        oplus_axis_plain<D>(X0, X0 + 9, -delta, Rm, tm);
        xp = w6_range_sq_plain(Rp, tp, off, q1);
        xm = w6_range_sq_plain(Rm, tm, off, q1);
    } else {
        oplus_axis_plain<D>(X1, X1 + 9, delta, Rp, tp);
        oplus_axis_plain<D>(X1, X1 + 9, -delta, Rm, tm);
        xp = w6_range_sq_plain(X0, X0 + 9, off, tp);
        xm = w6_range_sq_plain(X0, X0 + 9, off, tm);
    }
    const double ep = meas - (NEAR ? sqrt_ieee_near_c(xp, n0, h0, (h0 * h0) * (h0 + h0)) : sqrt_ieee_unscaled(xp));
    const double em = meas - (NEAR ? sqrt_ieee_near_c(xm, n0, h0, (h0 * h0) * (h0 + h0)) : sqrt_ieee_unscaled(xm));
    double bak = ep;
    bak -= em;
    return scalar * bak;
}
#pragma clang fp contract(fast)

constexpr int W6_NSLOT = 5;   // pose buffers in LDS: the state + up to four trial states
constexpr int W6_REC = 14;    // per (edge, moving endpoint): w, -w e, the pose's own J (6), the other endpoint's J (6) when that is the previous pose
constexpr int W6_PREC = 27;   // per prior: J^T W J (lower triangle, 21), -J^T W e (6)

// LDS of one window (doubles first, then ints)
struct W6Lds {
    double* pose;   // [5][nv_max][12] R (row-major), t
    double* rec;    // [2 nr_max][14]
    double* prec;   // [np_max][27]  the priors' records, grouped by pose
    double* ev;     // [nr_max][5]   measurement, information, lever arm of endpoint 0
    double* fix;    // [nr_max][3]   the fixed endpoint of an anchor edge
    double* pv;     // [np_max][18]  Z^-1 as R (9), t (3); information diagonal (6)
    double* sv;     // SE3: [48][64] the EdgeSE3 record of edge e in column e: Z^-1 as R t (12), information (36)
    int* eidx;      // [nr_max][2]
    int* epos;      // [nr_max][2]
    int* pidx;      // [np_max]
    int* ppos;      // [np_max]       where the prior's record goes (records grouped by pose)
};
template <bool SE3>
__device__ __forceinline__ W6Lds w6_carve(const WindowCaps& c) {
    W6Lds l;
    double* p = w6lds;
    l.pose = p; p += (size_t)W6_NSLOT * c.nv_max * 12;
    l.rec = p; p += (size_t)c.nr_max * 2 * W6_REC;
    l.prec = p; p += (size_t)c.np_max * W6_PREC;
    l.ev = p; p += (size_t)c.nr_max * 5;
    l.fix = p; p += (size_t)c.nr_max * 3;
    l.pv = p; p += (size_t)c.np_max * 18;
    l.sv = p; if (SE3) p += 48 * 64;
    int* q = reinterpret_cast<int*>(p);
    l.eidx = q; q += (size_t)c.nr_max * 2;
    l.epos = q; q += (size_t)c.nr_max * 2;
    l.pidx = q; q += c.np_max;
    l.ppos = q;
    return l;
}

struct W6Edge {
    int v0, v1, s0, s1;
    double meas, info, ox, oy, oz, fx, fy, fz;
};
__device__ __forceinline__ W6Edge w6_load_edge(const W6Lds& l, int e) {
    W6Edge E;
    E.v0 = l.eidx[2 * e]; E.v1 = l.eidx[2 * e + 1];
    E.s0 = l.epos[2 * e]; E.s1 = l.epos[2 * e + 1];
    E.meas = l.ev[5 * e]; E.info = l.ev[5 * e + 1];
    E.ox = l.ev[5 * e + 2]; E.oy = l.ev[5 * e + 3]; E.oz = l.ev[5 * e + 4];
    E.fx = l.fix[3 * e]; E.fy = l.fix[3 * e + 1]; E.fz = l.fix[3 * e + 2];
    return E;
}

// one range edge at the poses P: its robust / plain chi2; FULL: its linearisation records as well
template <bool FULL, int JAC>
__device__ __forceinline__ void w6_edge(const W6Lds& l, const double* P, const W6Edge& E, int tm, double& rsum, double& csum) {
    const int v0 = E.v0, v1 = E.v1;
    const double meas = E.meas, info = E.info;
    const double off[3] = {E.ox, E.oy, E.oz};
    double X0[12], X1[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) X0[k] = P[v0 * 12 + k];
    const int v1c = v1 >= 0 ? v1 : v0;
#pragma unroll
    for (int k = 0; k < 12; ++k) X1[k] = P[v1c * 12 + k];
    const double p1[3] = {v1 >= 0 ? X1[9] : E.fx, v1 >= 0 ? X1[10] : E.fy, v1 >= 0 ? X1[11] : E.fz};
    double err, inv = 0.0, u[3] = {0.0, 0.0, 0.0}, x0 = 0.0, n0 = 0.0, h0 = 0.0;
    if (JAC == 0) {
        double p0[3];
        w6_mat_vec(X0, off, p0);
        u[0] = (p0[0] + X0[9]) - p1[0]; u[1] = (p0[1] + X0[10]) - p1[1]; u[2] = (p0[2] + X0[11]) - p1[2];
        const double x = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
        double n;
        sqrt_and_rsqrt(x, n, inv);
        if (!(x > 0.0)) { n = 0.0; inv = 0.0; }   // coincident endpoints: J = 0, what the central difference gives
        err = meas - n;
    } else {
        x0 = w6_range_sq_plain(X0, X0 + 9, off, p1);
        n0 = sqrt_ieee_unscaled_h(x0, h0);
        err = meas - n0;
    }
    const double chi = err * (info * err);
    const double aux = 1.0 + chi;
    rsum += fast_log_ge1(aux);
    csum += chi;
    if (FULL) {
        double J0[6], J1[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (JAC == 0) {
            u[0] *= inv; u[1] *= inv; u[2] *= inv;
            double uR[3];
            w6_mat_tvec(X0, u, uR);   // (u^T R0)^T
            J0[0] = -uR[0]; J0[1] = -uR[1]; J0[2] = -uR[2];
            // dp0/dv = -2 R0 [o]x  =>  de/dv0 = 2 (uR x o)
            J0[3] = 2.0 * (uR[1] * off[2] - uR[2] * off[1]);
            J0[4] = 2.0 * (uR[2] * off[0] - uR[0] * off[2]);
            J0[5] = 2.0 * (uR[0] * off[1] - uR[1] * off[0]);
            if (v1 >= 0) {
                double uR1[3];
                w6_mat_tvec(X1, u, uR1);
                J1[0] = uR1[0]; J1[1] = uR1[1]; J1[2] = uR1[2];   // (no lever arm on endpoint 1: its rotation does not move its point)
            }
        } else {
            const bool near_ok = x0 >= 1e-5 && x0 < 1e300 && (off[0] * off[0] + off[1] * off[1] + off[2] * off[2]) <= 1.0;
            if (near_ok) {
                J0[0] = w6_jac_numeric<0, true>(X0, off, X1, p1, 0, meas, n0, h0);
                J0[1] = w6_jac_numeric<1, true>(X0, off, X1, p1, 0, meas, n0, h0);
                J0[2] = w6_jac_numeric<2, true>(X0, off, X1, p1, 0, meas, n0, h0);
                // With a zero lever arm a rotation of endpoint 0 does not move the ranged point at all: both perturbed evaluations of g2o's central
                // difference return the same number and the column is exactly +0 — the three columns are skipped when no lane here has a lever arm.
                if (__any(off[0] != 0.0 || off[1] != 0.0 || off[2] != 0.0)) {
                    J0[3] = w6_jac_numeric<3, true>(X0, off, X1, p1, 0, meas, n0, h0);
                    J0[4] = w6_jac_numeric<4, true>(X0, off, X1, p1, 0, meas, n0, h0);
                    J0[5] = w6_jac_numeric<5, true>(X0, off, X1, p1, 0, meas, n0, h0);
                } else { J0[3] = 0.0; J0[4] = 0.0; J0[5] = 0.0; }
                if (v1 >= 0) {
                    J1[0] = w6_jac_numeric<0, true>(X0, off, X1, p1, 1, meas, n0, h0);
                    J1[1] = w6_jac_numeric<1, true>(X0, off, X1, p1, 1, meas, n0, h0);
                    J1[2] = w6_jac_numeric<2, true>(X0, off, X1, p1, 1, meas, n0, h0);
                }
            } else {
                J0[0] = w6_jac_numeric<0, false>(X0, off, X1, p1, 0, meas, n0, h0);
                J0[1] = w6_jac_numeric<1, false>(X0, off, X1, p1, 0, meas, n0, h0);
                J0[2] = w6_jac_numeric<2, false>(X0, off, X1, p1, 0, meas, n0, h0);
                J0[3] = w6_jac_numeric<3, false>(X0, off, X1, p1, 0, meas, n0, h0);
                J0[4] = w6_jac_numeric<4, false>(X0, off, X1, p1, 0, meas, n0, h0);
                J0[5] = w6_jac_numeric<5, false>(X0, off, X1, p1, 0, meas, n0, h0);
                if (v1 >= 0) {
                    J1[0] = w6_jac_numeric<0, false>(X0, off, X1, p1, 1, meas, n0, h0);
                    J1[1] = w6_jac_numeric<1, false>(X0, off, X1, p1, 1, meas, n0, h0);
                    J1[2] = w6_jac_numeric<2, false>(X0, off, X1, p1, 1, meas, n0, h0);
                }
            }
        }
        const double wr = info * fast_rcp(aux), wre = -wr * err;
        // the endpoint that RECEIVES the other's Schur complement keeps the coupling (the other endpoint's J): the later pose of a consecutive
        // pair — in the two-sided elimination of the SE3 kernel (poses beyond the middle pose tm are eliminated downwards) the earlier one,
        // and the middle pose for both its neighbours (tm = INT_MAX: one-sided)
        const bool c0 = v1 >= 0 && ((v1 == v0 - 1 && v0 <= tm) || (v1 == v0 + 1 && v0 >= tm));
        const bool c1 = v1 >= 0 && ((v0 == v1 - 1 && v1 <= tm) || (v0 == v1 + 1 && v1 >= tm));
        double* r = l.rec + (size_t)E.s0 * W6_REC;
        r[0] = wr; r[1] = wre;
#pragma unroll
        for (int k = 0; k < 6; ++k) { r[2 + k] = J0[k]; r[8 + k] = c0 ? J1[k] : 0.0; }
        if (v1 >= 0) {
            double* r1 = l.rec + (size_t)E.s1 * W6_REC;
            r1[0] = wr; r1[1] = wre;
#pragma unroll
            for (int k = 0; k < 6; ++k) { r1[2 + k] = J1[k]; r1[8 + k] = c1 ? J0[k] : 0.0; }
        }
    }
}

// one EdgeSE3Prior (window_kernel.hip: evaluate_edges, unary priors): e = toVectorMQT(Z^-1 X), diagonal information, not robust
template <bool FULL>
__device__ __forceinline__ void w6_prior(const W6Lds& l, const double* P, int q, double& rsum, double& csum) {
    const int v = l.pidx[q];
    const double* valp = l.pv + (size_t)q * 18;
    double val[18], X[12];
#pragma unroll
    for (int k = 0; k < 18; ++k) val[k] = valp[k];
#pragma unroll
    for (int k = 0; k < 12; ++k) X[k] = P[v * 12 + k];
    double RE[9], tE[3], qq[4];
    w6_mat_mul(val, X, RE);
    w6_mat_vec(val, X + 9, tE);
    tE[0] += val[9]; tE[1] += val[10]; tE[2] += val[11];
    w6_mat_to_quat(RE, qq);
    w6_quat_normalize_sign(qq);
    const double err[6] = {tE[0], tE[1], tE[2], qq[1], qq[2], qq[3]};
    double chi = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) chi += err[i] * (val[12 + i] * err[i]);
    rsum += chi;
    csum += chi;
    if (FULL) {
        double* rec = l.prec + (size_t)l.ppos[q] * W6_PREC;
        // J = blockdiag(RE, Q) with Q = w I + [q_xyz]x (d vec(q (x) (sqrt(1 - |v|^2), v)) / dv at 0)
        double J[36];
#pragma unroll
        for (int i = 0; i < 36; ++i) J[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) J[i * 6 + j] = RE[i * 3 + j];
        J[3 * 6 + 3] = qq[0];  J[3 * 6 + 4] = -qq[3]; J[3 * 6 + 5] = qq[2];
        J[4 * 6 + 3] = qq[3];  J[4 * 6 + 4] = qq[0];  J[4 * 6 + 5] = -qq[1];
        J[5 * 6 + 3] = -qq[2]; J[5 * 6 + 4] = qq[1];  J[5 * 6 + 5] = qq[0];
        const double* W = val + 12;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
#pragma unroll
            for (int cc = 0; cc <= r; ++cc) {
                double h = 0.0;
                if ((r < 3) == (cc < 3)) {
#pragma unroll
                    for (int i = (r < 3 ? 0 : 3); i < (r < 3 ? 3 : 6); ++i) h += J[i * 6 + r] * W[i] * J[i * 6 + cc];
                }
                rec[r * (r + 1) / 2 + cc] = h;
            }
            double bb = 0.0;
#pragma unroll
            for (int i = (r < 3 ? 0 : 3); i < (r < 3 ? 3 : 6); ++i) bb += J[i * 6 + r] * (-W[i] * err[i]);
            rec[21 + r] = bb;
        }
    }
}

// every edge and prior of the window at the poses of buffer `buf`: the robust and plain chi2 sums; FULL: the linearisation records
template <bool FULL, int JAC>
__device__ __forceinline__ void w6_edges(const W6Lds& l, const W6Edge& E0, int nvm, int nr, int np, int buf, int lane, int tm, double& robust_chi, double& plain_chi) {
    double rsum = 0.0, csum = 0.0;
    const double* P = l.pose + (size_t)buf * nvm * 12;
    if (lane < nr) w6_edge<FULL, JAC>(l, P, E0, tm, rsum, csum);
    for (int e = lane + 64; e < nr; e += 64) w6_edge<FULL, JAC>(l, P, w6_load_edge(l, e), tm, rsum, csum);
    for (int q = lane; q < np; q += 64) w6_prior<FULL>(l, P, q, rsum, csum);
    robust_chi = w6_sum(rsum);
    plain_chi = w6_sum(csum);
}

// the sums of lanes 0 .. 31 and of lanes 32 .. 63: the DPP tree of w6_sum read after the row_bcast:15 step — bit for bit the whole-wave sum of
// values that sit in one half only
__device__ __forceinline__ void w6_half_sums(double v, double& lo, double& hi) {
    v += w6_dpp<0x111, 0xF>(v);
    v += w6_dpp<0x112, 0xF>(v);
    v += w6_dpp<0x114, 0xF>(v);
    v += w6_dpp<0x118, 0xF>(v);
    v += w6_dpp<0x142, 0xA>(v);
    lo = w6_bcast(v, 31); hi = w6_bcast(v, 63);
}
// TWO trial states scored in one pass (windows of <= 32 range edges and <= 32 priors): state A by lanes 0 .. 31, state B by lanes 32 .. 63, every
// lane with the edge of its position in its half (wave3_kernel.hip: w3_edges_dual)
template <int JAC>
__device__ __forceinline__ void w6_edges_dual(const W6Lds& l, const W6Edge& E0, int nvm, int nr, int np, int bufA, int bufB, int lane, int tm,
                                              double& chiA, double& plainA, double& chiB, double& plainB) {
    double rsum = 0.0, csum = 0.0;
    const int el = lane & 31;
    const double* P = l.pose + (size_t)(lane < 32 ? bufA : bufB) * nvm * 12;
    if (el < nr) w6_edge<false, JAC>(l, P, E0, tm, rsum, csum);
    if (el < np) w6_prior<false>(l, P, el, rsum, csum);
    w6_half_sums(rsum, chiA, chiB);
    w6_half_sums(csum, plainA, plainB);
}

#define W6_TRI(r, c) ((r) * ((r) + 1) / 2 + (c))

// the EdgeSE3 between a lane's pose and its sender, evaluated by that lane at the poses P: chi; FULL: its own / the sender's share of H
// and b (21 + 6 each), the coupling block KO (rows: the lane's pose, column-major)
struct W6Se3 { int e, i, j; bool robust; };
template <bool FULL>
__device__ __forceinline__ double w6_se3(const W6Lds& l, const double* P, const W6Se3& se, int col, int own_pose, double* own, double* oth, double* KO, double& rterm) {
    double Xi[12], Xj[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) { Xi[k] = P[se.i * 12 + k]; Xj[k] = P[se.j * 12 + k]; }
    const bool jl = se.j == own_pose;   // (chain_se3_terms: the coupling block's rows are pose j's when the flag is set, pose i's otherwise)
    if (FULL) {
        double Hii[21], Hjj[21], bi[6], bj[6];
        const double chi = chain_se3_terms<true, 64>(Xi, Xj, l.sv + col, se.robust, jl, Hii, Hjj, KO, bi, bj, rterm);
#pragma unroll
        for (int k = 0; k < 21; ++k) { own[k] = jl ? Hjj[k] : Hii[k]; oth[k] = jl ? Hii[k] : Hjj[k]; }
#pragma unroll
        for (int k = 0; k < 6; ++k) { own[21 + k] = jl ? bj[k] : bi[k]; oth[21 + k] = jl ? bi[k] : bj[k]; }
        return chi;
    }
    return chain_se3_terms<false, 64>(Xi, Xj, l.sv + col, se.robust, jl, nullptr, nullptr, nullptr, nullptr, nullptr, rterm);
}

template <int JAC, bool SE3>
__global__ void __launch_bounds__(64) wave6_lm_kernel(const WindowArgs a) {
    const int lane = threadIdx.x;
    const long long inst = blockIdx.x;
    const WindowCaps& cp = a.caps;
    const W6Lds l = w6_carve<SE3>(cp);
    const int nvm = cp.nv_max;
    const double* gin = a.poses_in + (size_t)inst * nvm * 12;
    double* gout = a.poses + (size_t)inst * nvm * 12;
    const int32_t* ridx = a.r_idx + (size_t)inst * cp.nr_max * 2;
    const double* rval = a.r_val + (size_t)inst * cp.nr_max * 5;
    // the first 64 poses are requested before the counts have arrived (wave3_kernel.hip)
    double pin[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) pin[k] = 0.0;
    if (lane < nvm) {
#pragma unroll
        for (int k = 0; k < 12; ++k) pin[k] = gin[lane * 12 + k];
    }
    const int nv = a.counts[inst * 4 + 0], nr = a.counts[inst * 4 + 1], np = a.counts[inst * 4 + 2];
    // speculation: G groups of W lanes solve the SAME normal equations with the lambdas of this and the next G - 1 trials; a group
    // keeps one idle lane after its last pose (the hand-over between lanes reads zeros there)
    const int W = nv <= 15 ? 16 : (nv <= 31 ? 32 : 64);
    const int G = 64 / W;
    const int grp = lane / W, lg = lane & (W - 1);
    // lane <-> pose.  Rank-1 kernel: lane lg of a group is pose lg, eliminated after pose lg - 1.  SE3: the block elimination runs from BOTH
    // ends of the chain towards the middle pose tm in the same instructions — poses 0 .. tm in lanes 0 .. tm, poses nv - 1 .. tm + 1 in lanes
    // tm + 1 .. nv - 1 and the middle pose ONCE MORE in lane nv (`dup`: the end of the second chain), so that in both chains a lane takes
    // its sender's factor from the lane before it; the two Schur complements of the middle pose are added at the end.  Half the
    // sequential steps.  (nv + 1 lanes per group: nv <= 63.)
    const int tm = SE3 ? (nv - 1) / 2 : 0x7fffffff;
    int pp = lg, qq = lg - 1;      // this lane's pose; the pose eliminated just before it in its chain (the sender), -1: none
    bool pose = lg < nv, dup = false;
    if (SE3) {
        if (lg > tm) { pp = nv + tm - lg; qq = lg > tm + 1 ? pp + 1 : -1; }
        pose = lg <= nv && nv > 0 && !(nv == 1 && lg == 1);
        dup = nv > 1 && lg == nv;
        if (!pose) { pp = 0; qq = -1; }
    }
    const bool mid = SE3 && pose && lg == tm;   // (its chain's last step; with `dup` the two halves of the middle pose)
    const unsigned long long group_mask = W == 64 ? ~0ull : ((1ull << W) - 1ull);
    W6_T0();
    // ---- set-up ------------------------------------------------------------------------------------------------------------------
    if (lane < nv) {
#pragma unroll
        for (int k = 0; k < 12; ++k) l.pose[lane * 12 + k] = pin[k];
    }
    for (int e = lane; e < nr; e += 64) {
        const int v0 = ridx[2 * e], v1 = ridx[2 * e + 1];
        l.eidx[2 * e] = v0; l.eidx[2 * e + 1] = v1;
#pragma unroll
        for (int k = 0; k < 5; ++k) l.ev[5 * e + k] = rval[5 * e + k];
        l.epos[2 * e + 1] = -1;
        const double* an = a.anchors + (size_t)(v1 < 0 ? -1 - v1 : 0) * 3;
        l.fix[3 * e] = an[0]; l.fix[3 * e + 1] = an[1]; l.fix[3 * e + 2] = an[2];
    }
    {
        const int32_t* pidx = a.p_idx + (size_t)inst * cp.np_max;
        const double* pval = a.p_val + (size_t)inst * cp.np_max * 18;
        for (int q = lane; q < np; q += 64) {
            l.pidx[q] = pidx[q];
#pragma unroll
            for (int k = 0; k < 18; ++k) l.pv[18 * q + k] = pval[18 * q + k];
        }
    }
    W6Se3 se = {-1, 0, 0, false};
    if (SE3) {
        // lane e takes edge e's index row (one round trip for all of them), then every lane looks for the edge between its pose and its sender
        const int32_t* sidx = a.s_idx + (size_t)inst * cp.ns_max * 4;
        int ei = -1, ej = -1, er = 0;
        if (lane < cp.ns_max) { ei = sidx[4 * lane]; ej = sidx[4 * lane + 1]; er = sidx[4 * lane + 2]; }   // (requested before the count has arrived)
        const int ns = a.counts[inst * 4 + 3];
        if (lane < ns) {   // edge e's record -> column e (one pass over it; which lane evaluates it is found out meanwhile)
            const double* val = a.s_val + ((size_t)inst * cp.ns_max + lane) * 48;
#pragma unroll
            for (int k = 0; k < 48; ++k) l.sv[k * 64 + lane] = val[k];
        }
        for (int e = 0; e < ns && e < 64; ++e) {
            const int i2 = __builtin_amdgcn_readlane(ei, e), j2 = __builtin_amdgcn_readlane(ej, e), r2 = __builtin_amdgcn_readlane(er, e);
            if (pose && qq >= 0 && ((i2 == pp && j2 == qq) || (i2 == qq && j2 == pp))) { se.e = e; se.i = i2; se.j = j2; se.robust = r2 != 0; }
        }
    }
    w6_sync();
    int deg = 0;   // (lanes 0 .. nv-1: pose = lane)
    for (int e = 0; e < nr; ++e) {
        const int v0 = l.eidx[2 * e], v1 = l.eidx[2 * e + 1];
        deg += (v0 == lane) + (v1 == lane);
    }
    int lst = 0;   // first record of this pose: exclusive prefix sum of deg over the lanes
    int kc = -1, kcr = -1;   // the record of the edge to the previous / (SE3) to the next pose
    {
        int incl = deg;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        lst = incl - deg;
        int k = lst;
        for (int e = 0; e < nr; ++e) {
            const int v0 = l.eidx[2 * e], v1 = l.eidx[2 * e + 1];
            if (v0 == lane) { if (v1 >= 0 && v1 == lane - 1) kc = k; if (v1 == lane + 1) kcr = k; l.epos[2 * e] = k++; }
            if (v1 == lane) { if (v0 == lane - 1) kc = k; if (v0 == lane + 1) kcr = k; l.epos[2 * e + 1] = k++; }
        }
    }
    int pdeg = 0, plst = 0;   // the same for the pose's priors
    for (int pq = 0; pq < np; ++pq) pdeg += l.pidx[pq] == lane;
    {
        int incl = pdeg;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        plst = incl - pdeg;
        int k = plst;
        for (int pq = 0; pq < np; ++pq)
            if (l.pidx[pq] == lane) l.ppos[pq] = k++;
    }
    deg = __shfl(deg, pp, 64);   // every group's lane of pose pp
    lst = __shfl(lst, pp, 64);
    kc = __shfl(kc, pp, 64);
    kcr = __shfl(kcr, pp, 64);
    if (SE3) kc = qq < 0 ? -1 : (qq == pp + 1 ? kcr : kc);   // the record of the edge to the SENDER
    pdeg = __shfl(pdeg, pp, 64);
    plst = __shfl(plst, pp, 64);
    w6_sync();
    W6Edge E0;
    E0.v0 = 0; E0.v1 = -1; E0.s0 = 0; E0.s1 = -1; E0.meas = 0.0; E0.info = 0.0; E0.ox = 0.0; E0.oy = 0.0; E0.oz = 0.0; E0.fx = 0.0; E0.fy = 0.0; E0.fz = 0.0;
    // (speculative trials are scored two at a time when a window's edges fit half a wave: the upper half keeps the same edges)
    const bool dual = G >= 2 && nr <= 32 && np <= 32;
    if ((dual ? (lane & 31) : lane) < nr) E0 = w6_load_edge(l, dual ? (lane & 31) : lane);
    W6_T(0);

    // ---- Levenberg-Marquardt (g2o: OptimizationAlgorithmLevenberg::solve), wave-uniform control flow ---------------
    constexpr double tau = 1e-5, good_lo = 1.0 / 3.0, good_hi = 2.0 / 3.0;
    constexpr int max_trials = 10;
    double lambda = 0.0, ni = 2.0, cur_chi = 0.0, last_plain = 0.0;
    int it = 0, q = 0, trials = 0, terminated = 0, cur = 0, jlast = 0;
    bool need_lin = true;
    bool done = nv <= 0 || nr + np <= 0 || a.iterations <= 0;
    double D[21], b[6], u[6], vnext[6], X[6];   // this pose's H_pp (lower), b_p; H_p,p-1 = u v^T; the v of pose p + 1's block; x_p of this group's last solve
    double K[SE3 ? 36 : 1];                      // SE3: the full coupling block H_p,p-1 (rows: pose p), column-major
    int shared_edges = 0;
#pragma unroll
    for (int k = 0; k < 21; ++k) D[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) { b[k] = 0.0; u[k] = 0.0; vnext[k] = 0.0; X[k] = 0.0; }
    while (!done) {
        if (need_lin) {
            double plain;
            w6_edges<true, JAC>(l, E0, nvm, nr, np, cur, lane, tm, cur_chi, plain);
            double own[SE3 ? 27 : 1], oth[SE3 ? 27 : 1];
            if (SE3) {
                // (every group's lane of pose p linearises the pair's EdgeSE3 — the same instructions for all groups; group 0's chi counts)
                double rs = 0.0, cs = 0.0;
#pragma unroll
                for (int k = 0; k < 27; ++k) { own[k] = 0.0; oth[k] = 0.0; }
#pragma unroll
                for (int k = 0; k < 36; ++k) K[k] = 0.0;
                if (se.e >= 0) {
                    double rterm;
                    const double chi = w6_se3<true>(l, l.pose + (size_t)cur * nvm * 12, se, se.e, pp, own, oth, K, rterm);
                    if (grp == 0) { rs = rterm; cs = chi; }
                }
                cur_chi += w6_sum(rs);
                plain += w6_sum(cs);
            }
            last_plain = plain;
            w6_sync();
            W6_T(1);
#pragma unroll
            for (int k = 0; k < 21; ++k) D[k] = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) { b[k] = 0.0; u[k] = 0.0; }
            double vv[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            if (pose) {
                const int k1 = lst + deg;
                for (int k = lst; k < k1; ++k) {
                    const double* r = l.rec + (size_t)k * W6_REC;
                    const double wr = r[0], wre = r[1];
                    double Jm[6], wj[6];
#pragma unroll
                    for (int c = 0; c < 6; ++c) { Jm[c] = r[2 + c]; wj[c] = wr * Jm[c]; }
                    if (!dup) {   // (the middle pose's second lane only takes the coupling with ITS sender)
#pragma unroll
                        for (int rr = 0; rr < 6; ++rr) {
#pragma unroll
                            for (int cc = 0; cc <= rr; ++cc) D[W6_TRI(rr, cc)] = __builtin_fma(wj[rr], Jm[cc], D[W6_TRI(rr, cc)]);
                            b[rr] = __builtin_fma(Jm[rr], wre, b[rr]);
                        }
                    }
                    if (k == kc) {
#pragma unroll
                        for (int c = 0; c < 6; ++c) { u[c] = wj[c]; vv[c] = r[8 + c]; }
                    }
                }
                for (int pq = plst; pq < (dup ? plst : plst + pdeg); ++pq) {
                    const double* r = l.prec + (size_t)pq * W6_PREC;
#pragma unroll
                    for (int k = 0; k < 21; ++k) D[k] += r[k];
#pragma unroll
                    for (int k = 0; k < 6; ++k) b[k] += r[21 + k];
                }
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) vnext[c] = w6_from_next(vv[c]);
            if (SE3) {
                // the EdgeSE3's shares: own, and the NEXT lane (whose sender this lane is) hands down what its edge gives this lane's pose
                // (zeros from a lane without one); the pair's coupling block = the EdgeSE3's + the range edge's rank-1 part.  The middle pose's
                // H_mm and b_m stay split over its two lanes (the second holds the share of ITS edge): every use below is a sum over both.
#pragma unroll
                for (int k = 0; k < 21; ++k) D[k] += own[k] + w6_from_next(oth[k]);
#pragma unroll
                for (int k = 0; k < 6; ++k) b[k] += own[21 + k] + w6_from_next(oth[21 + k]);
#pragma unroll
                for (int c = 0; c < 6; ++c)
#pragma unroll
                    for (int r = 0; r < 6; ++r) K[6 * c + r] = __builtin_fma(u[r], vv[c], K[6 * c + r]);
                if (it == 0) shared_edges = (int)w6_sum(grp == 0 && se.e >= 0 && kc >= 0 ? 2.0 : 0.0);
            }
            if (it == 0) {
                double md = 0.0;
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    double dg = D[W6_TRI(r, r)];
                    if (SE3 && nv > 1) { const double t = __shfl(dg, grp * W + nv, 64); dg = mid ? dg + t : (dup ? 0.0 : dg); }   // (the middle pose's diagonal: both lanes' shares)
                    md = fmax(md, fabs(dg));
                }
                lambda = tau * w6_max(pose ? md : 0.0);
                ni = 2.0;
            }
            q = 0;
            need_lin = false;
            W6_T(2);
        }
        // ---- one round: (H + lambda_g I) x = b for the lambdas of the next G trials --------------------------------------------------
        double lamv[4], niv[4];
        lamv[0] = lambda; niv[0] = ni;
#pragma unroll
        for (int g = 1; g < 4; ++g) { lamv[g] = lamv[g - 1] * niv[g - 1]; niv[g] = 2.0 * niv[g - 1]; }   // (a rejected trial: lambda *= ni, ni *= 2)
        const double mylam = grp == 0 ? lamv[0] : (grp == 1 ? lamv[1] : (grp == 2 ? lamv[2] : lamv[3]));
        // The solve.  H is block-tridiagonal with rank-1 couplings H_p,p-1 = u_p v_p^T, so the Schur complement of pose p is
        // S_p = A_p - alpha_p u_p u_p^T with A_p = H_pp + lambda I and the SCALAR alpha_p = v_p^T S_{p-1}^-1 v_p, its right-hand side
        // b_p - beta_p u_p with beta_p = v_p^T S_{p-1}^-1 (b_{p-1} - beta_{p-1} u_{p-1}).  Sherman-Morrison gives S_p^-1 from A_p^-1:
        //   S^-1 = A^-1 + k (A^-1 u)(A^-1 u)^T,  k = alpha / den,  den = 1 - alpha u^T A^-1 u   (S positive definite <=> A is and den > 0)
        // so everything that needs a 6x6 factorisation — A_p = G G^T, A^-1 u, A^-1 v, A^-1 b and their five dot products — is done by ALL
        // poses AT ONCE (one pass, no dependence between poses), and the sequential part of the block-tridiagonal elimination shrinks to
        // a recurrence on two scalars per pose (forward: alpha, beta; backward: gamma = u_{p+1} . x_{p+1}), ~15 instructions per pose
        // instead of a 6x6 Cholesky and two triangular solves.  Each repetition every lane takes its neighbour's scalars through DPP and
        // redoes its own step (wave3_kernel.hip): after repetition r the poses 0 .. r hold final values; all G groups ride along.
        unsigned long long bad = 0;
        double Xn[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if constexpr (!SE3) {
            double pA[6], qA[6], rA[6];   // A^-1 u, A^-1 v_next, A^-1 b
            double uu = 0.0, uv = 0.0, vvq = 0.0, ub = 0.0, vb = 0.0;
            bool piv_ok = true;
    #pragma unroll
            for (int k = 0; k < 6; ++k) { pA[k] = 0.0; qA[k] = 0.0; rA[k] = 0.0; }
            if (pose) {
                double A[6][6], ig[6];
    #pragma unroll
                for (int rr = 0; rr < 6; ++rr) {
    #pragma unroll
                    for (int c = 0; c <= rr; ++c) A[rr][c] = D[W6_TRI(rr, c)];
                    A[rr][rr] += mylam;
                }
    #pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const double g = w6_pivot_rsqrt(A[j][j]);
                    ig[j] = g;
    #pragma unroll
                    for (int i2 = j + 1; i2 < 6; ++i2) A[i2][j] *= g;
    #pragma unroll
                    for (int i2 = j + 1; i2 < 6; ++i2)
    #pragma unroll
                        for (int cc = j + 1; cc <= i2; ++cc) A[i2][cc] = __builtin_fma(-A[i2][j], A[cc][j], A[i2][cc]);
                }
                piv_ok = ((ig[0] + ig[1]) + (ig[2] + ig[3])) + (ig[4] + ig[5]) < DBL_MAX;
                // three right-hand sides through G and G^T
    #pragma unroll
                for (int k = 0; k < 6; ++k) { pA[k] = u[k]; qA[k] = vnext[k]; rA[k] = b[k]; }
    #pragma unroll
                for (int cc = 0; cc < 6; ++cc) {
                    pA[cc] *= ig[cc]; qA[cc] *= ig[cc]; rA[cc] *= ig[cc];
    #pragma unroll
                    for (int c2 = cc + 1; c2 < 6; ++c2) {
                        pA[c2] = __builtin_fma(-pA[cc], A[c2][cc], pA[c2]);
                        qA[c2] = __builtin_fma(-qA[cc], A[c2][cc], qA[c2]);
                        rA[c2] = __builtin_fma(-rA[cc], A[c2][cc], rA[c2]);
                    }
                }
    #pragma unroll
                for (int cc = 5; cc >= 0; --cc) {
                    double a0 = pA[cc], a1 = qA[cc], a2 = rA[cc];
    #pragma unroll
                    for (int c2 = cc + 1; c2 < 6; ++c2) {
                        a0 = __builtin_fma(-A[c2][cc], pA[c2], a0);
                        a1 = __builtin_fma(-A[c2][cc], qA[c2], a1);
                        a2 = __builtin_fma(-A[c2][cc], rA[c2], a2);
                    }
                    pA[cc] = a0 * ig[cc]; qA[cc] = a1 * ig[cc]; rA[cc] = a2 * ig[cc];
                }
    #pragma unroll
                for (int k = 0; k < 6; ++k) {
                    uu = __builtin_fma(u[k], pA[k], uu); uv = __builtin_fma(u[k], qA[k], uv); vvq = __builtin_fma(vnext[k], qA[k], vvq);
                    ub = __builtin_fma(u[k], rA[k], ub); vb = __builtin_fma(vnext[k], rA[k], vb);
                }
            }
            W6_T(3);
            // forward recurrence: (alpha, beta) of pose p from those of pose p - 1
            double bin = 0.0, kk = 0.0, rden = 1.0, den = 1.0;
            {
                double al = 0.0, be = 0.0;
                for (int r = 0; r < nv; ++r) {
                    const double pal = w6_from_prev(al), pbe = w6_from_prev(be);
                    if (pose) {
                        bin = pbe;
                        den = __builtin_fma(-pal, uu, 1.0);
                        rden = fast_rcp(den);
                        kk = pal * rden;
                        const double kuv = kk * uv;
                        al = __builtin_fma(kuv, uv, vvq);
                        be = __builtin_fma(kuv, __builtin_fma(-pbe, uu, ub), __builtin_fma(-pbe, uv, vb));
                    }
                }
            }
            // a group in which some A_p or some Schur complement is not positive definite (or not finite) has failed
            bad = __ballot(pose && !(piv_ok && den > 0.0 && den < DBL_MAX));
            // backward recurrence: gamma = u_{p+1} . x_{p+1};  x_p = S_p^-1 (b_p - beta_p u_p - gamma v_{p+1})
            {
                double ga = 0.0, gin = 0.0, cfin = 0.0;
                for (int r = 0; r < nv; ++r) {
                    const double pga = w6_from_next(ga);
                    if (pose) {
                        gin = pga;
                        cfin = __builtin_fma(-pga, uv, __builtin_fma(-bin, uu, ub));   // u^T A^-1 (b - beta u - gamma v)
                        ga = cfin * rden;
                    }
                }
                if (pose) {
                    const double cb = __builtin_fma(-kk, cfin, bin);   // x = A^-1 b - (beta - k c) A^-1 u - gamma A^-1 v
    #pragma unroll
                    for (int k = 0; k < 6; ++k) Xn[k] = __builtin_fma(-gin, qA[k], __builtin_fma(-cb, pA[k], rA[k]));
                }
            }
        } else {
            // The solve with full coupling blocks: block Cholesky of the block-tridiagonal H + lambda I, eliminated from both ends towards the
            // middle pose.  With q the sender of pose p (the pose before it in its chain) and K_p = H_p,q:
            //   W_p = G_q^-1 K_p^T,  S_p = A_p - W_p^T W_p = G_p G_p^T,  y_p = G_p^-1 (b_p - W_p^T y_q);   back: x_q = G_q^-T (y_q - W_p x_p).
            // Forward: every repetition every lane takes the factor and y of the lane before it through DPP (15 + 6 + 6 numbers) and redoes its
            // step; after repetition r the first r + 1 poses of both chains hold final values (a lane whose inputs are final recomputes the
            // same numbers).  The middle pose's two lanes then add their Schur complements and both factor the sum.  Backward the same way round.
            double Gl[15], ig[6], yv[6], Wm[36];   // strict lower triangle of G_p (column-major: (1,0) .. (5,0), (2,1) ..), reciprocal pivots, y_p, W_p (entry (k, c) at 6 k + c)
            double Su[27];                         // S_p and its right-hand side before the factorisation (the middle pose's lanes add theirs)
#pragma unroll
            for (int k = 0; k < 15; ++k) Gl[k] = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) { ig[k] = 0.0; yv[k] = 0.0; }
#pragma unroll
            for (int k = 0; k < 36; ++k) Wm[k] = 0.0;
#pragma unroll
            for (int k = 0; k < 27; ++k) Su[k] = 0.0;
            const int reps = (tm > nv - 1 - tm ? tm : nv - 1 - tm) + 1;
            const double lamd = dup ? 0.0 : mylam;   // (the middle pose's lambda rides in its first lane)
            auto factor = [&](double (&A)[6][6], double* rhs) __attribute__((always_inline)) {   // A = G G^T in place, ig, Gl; rhs <- G^-1 rhs = yv
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const double g = w6_pivot_rsqrt(A[j][j]);
                    ig[j] = g;
#pragma unroll
                    for (int i2 = j + 1; i2 < 6; ++i2) A[i2][j] *= g;
#pragma unroll
                    for (int i2 = j + 1; i2 < 6; ++i2)
#pragma unroll
                        for (int cc = j + 1; cc <= i2; ++cc) A[i2][cc] = __builtin_fma(-A[i2][j], A[cc][j], A[i2][cc]);
                }
                int kk2 = 0;
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    rhs[c] *= ig[c];
#pragma unroll
                    for (int c2 = c + 1; c2 < 6; ++c2) { rhs[c2] = __builtin_fma(-rhs[c], A[c2][c], rhs[c2]); Gl[kk2] = A[c2][c]; ++kk2; }
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) yv[k] = rhs[k];
            };
            for (int r = 0; r < reps; ++r) {
                double Gq[15], iq[6], yq[6];
#pragma unroll
                for (int k = 0; k < 15; ++k) Gq[k] = w6_from_prev(Gl[k]);
#pragma unroll
                for (int k = 0; k < 6; ++k) { iq[k] = w6_from_prev(ig[k]); yq[k] = w6_from_prev(yv[k]); }
                if (pose) {
                    // W = G_q^-1 K^T, one forward substitution per row rr of K.  (A chain's first pose has no sender: W stays 0, and what the
                    // lane before it holds — another chain's or group's end, possibly not finite — is not looked at.)
                    if (qq >= 0) {
#pragma unroll
                        for (int rr = 0; rr < 6; ++rr) {
                            double w[6];
#pragma unroll
                            for (int c = 0; c < 6; ++c) w[c] = K[6 * c + rr];
                            int kk2 = 0;
#pragma unroll
                            for (int c = 0; c < 6; ++c) {
                                w[c] *= iq[c];
#pragma unroll
                                for (int c2 = c + 1; c2 < 6; ++c2) { w[c2] = __builtin_fma(-w[c], Gq[kk2], w[c2]); ++kk2; }
                            }
#pragma unroll
                            for (int c = 0; c < 6; ++c) Wm[6 * c + rr] = w[c];
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 6; ++k) yq[k] = 0.0;
                    }
                    double A[6][6], rhs[6];
#pragma unroll
                    for (int rr = 0; rr < 6; ++rr) {
#pragma unroll
                        for (int c = 0; c <= rr; ++c) {
                            double s2 = D[W6_TRI(rr, c)];
#pragma unroll
                            for (int k = 0; k < 6; ++k) s2 = __builtin_fma(-Wm[6 * k + rr], Wm[6 * k + c], s2);
                            A[rr][c] = s2;
                        }
                        A[rr][rr] += lamd;
                        double s3 = b[rr];
#pragma unroll
                        for (int k = 0; k < 6; ++k) s3 = __builtin_fma(-Wm[6 * k + rr], yq[k], s3);
                        rhs[rr] = s3;
                    }
#pragma unroll
                    for (int rr = 0; rr < 6; ++rr) {
#pragma unroll
                        for (int c = 0; c <= rr; ++c) Su[W6_TRI(rr, c)] = A[rr][c];
                        Su[21 + rr] = rhs[rr];
                    }
                    factor(A, rhs);
                }
            }
            if (nv > 1) {   // the middle pose: S_m = (A_m - W^T W of its first chain) + (- W^T W of the second), factored by both its lanes
                const int partner = grp * W + (mid ? nv : tm);
                double Sp[27];
#pragma unroll
                for (int k = 0; k < 27; ++k) Sp[k] = __shfl(Su[k], partner, 64);
                if (mid || dup) {
                    double A[6][6], rhs[6];
#pragma unroll
                    for (int rr = 0; rr < 6; ++rr) {
#pragma unroll
                        for (int c = 0; c <= rr; ++c) A[rr][c] = Su[W6_TRI(rr, c)] + Sp[W6_TRI(rr, c)];
                        rhs[rr] = Su[21 + rr] + Sp[21 + rr];
                    }
                    factor(A, rhs);
                }
            }
            // a group in which some Schur complement is not positive definite (or not finite) has failed
            bad = __ballot(pose && !(((ig[0] + ig[1]) + (ig[2] + ig[3])) + (ig[4] + ig[5]) < DBL_MAX));
            {
                double tn[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // W_p x_p: what the sender of pose p subtracts
                auto back = [&](const double* tq) __attribute__((always_inline)) {
                    double t[6];
#pragma unroll
                    for (int k = 0; k < 6; ++k) t[k] = yv[k] - tq[k];
#pragma unroll
                    for (int rr = 5; rr >= 0; --rr) {
                        Xn[rr] = t[rr] * ig[rr];
#pragma unroll
                        for (int q2 = 0; q2 < rr; ++q2) t[q2] = __builtin_fma(-Gl[q2 * 5 - q2 * (q2 - 1) / 2 + (rr - q2 - 1)], Xn[rr], t[q2]);   // G(rr, q2)
                    }
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        double s2 = 0.0;
#pragma unroll
                        for (int c = 0; c < 6; ++c) s2 = __builtin_fma(Wm[6 * k + c], Xn[c], s2);
                        tn[k] = s2;
                    }
                };
                const double zero6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
                if (mid || dup) back(zero6);   // x of the middle pose (the last one eliminated), in both its lanes
                for (int r = 0; r + 1 < reps; ++r) {
                    double tq[6];
#pragma unroll
                    for (int k = 0; k < 6; ++k) tq[k] = w6_from_next(tn[k]);
                    if (pose && !mid && !dup) back(tq);
                }
            }
        }
        // x of a failed factorisation: g2o leaves its x alone, and LM applies that stale x all the same
        if (bad) {
            double st[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) st[k] = __shfl(X[k], jlast * W + lg, 64);
            for (int g = 0; g < G; ++g) {
                if (grp == g && ((bad >> (g * W)) & group_mask)) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) Xn[k] = st[k];
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) st[k] = __shfl(Xn[k], g * W + lg, 64);
            }
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) X[k] = Xn[k];
        W6_T(4);
        // the trial states X * fromVectorMQT(x) (VertexSE3::oplus) and g2o's computeScale sums, one per group
        double scv[4];
        {
            double sc = 0.0;
            if (dup) {   // (b_m's second share: x_m . b_m is a sum over both lanes, lambda |x_m|^2 rides in the first)
#pragma unroll
                for (int k = 0; k < 6; ++k) sc += X[k] * b[k];
            }
            if (pose && !dup) {
                const int slot = cur + 1 + grp - (cur + 1 + grp >= W6_NSLOT ? W6_NSLOT : 0);
                const double* s = l.pose + ((size_t)cur * nvm + pp) * 12;
                double* d = l.pose + ((size_t)slot * nvm + pp) * 12;
                double Xc[12];
#pragma unroll
                for (int k = 0; k < 12; ++k) Xc[k] = s[k];
#pragma unroll
                for (int k = 0; k < 6; ++k) sc += X[k] * (mylam * X[k] + b[k]);
                double Rd[9];
                const double ww = 1.0 - (X[3] * X[3] + X[4] * X[4] + X[5] * X[5]);
                if (ww < 0) { Rd[0] = 1; Rd[1] = 0; Rd[2] = 0; Rd[3] = 0; Rd[4] = 1; Rd[5] = 0; Rd[6] = 0; Rd[7] = 0; Rd[8] = 1; }
                else { const double qd[4] = {sqrt(ww), X[3], X[4], X[5]}; w6_quat_to_mat(qd, Rd); }
                double Rn[9], tn[3];
                w6_mat_mul(Xc, Rd, Rn);
                w6_mat_vec(Xc, X, tn);
#pragma unroll
                for (int k = 0; k < 9; ++k) d[k] = Rn[k];
                d[9] = Xc[9] + tn[0]; d[10] = Xc[10] + tn[1]; d[11] = Xc[11] + tn[2];
            }
            sc += w6_dpp<0x111, 0xF>(sc);
            sc += w6_dpp<0x112, 0xF>(sc);
            sc += w6_dpp<0x114, 0xF>(sc);
            sc += w6_dpp<0x118, 0xF>(sc);
            const double s16 = sc;
            sc += w6_dpp<0x142, 0xA>(sc);
            const double s32 = sc;
            sc += w6_dpp<0x143, 0xC>(sc);
#pragma unroll
            for (int g = 0; g < 4; ++g) scv[g] = W == 16 ? w6_bcast(s16, 16 * g + 15) : (W == 32 ? w6_bcast(s32, 32 * (g & 1) + 31) : w6_bcast(sc, 63));
        }
        w6_sync();
        // ---- consume the trials in LM's order until one is accepted (or the iteration ends) -------------------------------------------
        bool iteration_over = false;
        double rho = 0.0;
        double tchi[4] = {0.0, 0.0, 0.0, 0.0}, tplain[4] = {0.0, 0.0, 0.0, 0.0};
        bool scored[4] = {false, false, false, false};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < G && !iteration_over) {
                const int slot = cur + 1 + g - (cur + 1 + g >= W6_NSLOT ? W6_NSLOT : 0);
                if (!scored[g]) {
                    if (g < 3 && dual && g + 1 < G) {   // this trial and the next one in one pass
                        const int g1 = g < 3 ? g + 1 : 3;
                        const int slot2 = cur + 2 + g - (cur + 2 + g >= W6_NSLOT ? W6_NSLOT : 0);
                        w6_edges_dual<JAC>(l, E0, nvm, nr, np, slot, slot2, lane, tm, tchi[g], tplain[g], tchi[g1], tplain[g1]);
                        if (SE3) {   // the EdgeSE3 factors: the lanes of the first group of each half
                            double rs = 0.0, cs = 0.0;
                            if ((lane & 31) < W && se.e >= 0) cs = w6_se3<false>(l, l.pose + (size_t)(lane < 32 ? slot : slot2) * nvm * 12, se, se.e, pp, nullptr, nullptr, nullptr, rs);
                            double ra, rb, ca, cb;
                            w6_half_sums(rs, ra, rb);
                            w6_half_sums(cs, ca, cb);
                            tchi[g] += ra; tplain[g] += ca; tchi[g1] += rb; tplain[g1] += cb;
                        }
                        scored[g1] = true;
                    } else {
                        w6_edges<false, JAC>(l, E0, nvm, nr, np, slot, lane, tm, tchi[g], tplain[g]);
                        if (SE3) {   // the EdgeSE3 factors, group 0's lanes
                            double rs = 0.0, cs = 0.0;
                            if (grp == 0 && se.e >= 0) cs = w6_se3<false>(l, l.pose + (size_t)slot * nvm * 12, se, se.e, pp, nullptr, nullptr, nullptr, rs);
                            tchi[g] += w6_sum(rs);
                            tplain[g] += w6_sum(cs);
                        }
                    }
                }
                double temp_chi = tchi[g];
                const double plain2 = tplain[g];
                last_plain = plain2;
                ++trials;
                jlast = g;
                if ((bad >> (g * W)) & group_mask) temp_chi = DBL_MAX;
                const double scale = scv[g] + 1e-3;
                rho = (cur_chi - temp_chi) / scale;
                if (rho > 0.0 && fabs(temp_chi) <= DBL_MAX) {
                    const double r21 = 2.0 * rho - 1.0;
                    double alpha = 1.0 - r21 * r21 * r21;
                    alpha = fmin(alpha, good_hi);
                    lambda = lamv[g] * fmax(good_lo, alpha);
                    ni = 2.0;
                    cur_chi = temp_chi;
                    cur = slot;   // the trial state is the state
                    ++q;
                    iteration_over = true;
                } else {
                    lambda = lamv[g] * niv[g];
                    ni = 2.0 * niv[g];      // (pop: the state was never overwritten)
                    ++q;
                    iteration_over = !(rho < 0.0 && q < max_trials);
                }
            }
        }
        W6_T(5);
        if (iteration_over) {
            ++it;
            need_lin = true;
            if (q == max_trials || rho == 0.0) { terminated = 1; done = true; }
            if (it >= a.iterations) done = true;
        }
        W6_T(6);
    }
    if (lane < nv) {
        const double* s = l.pose + ((size_t)cur * nvm + lane) * 12;
#pragma unroll
        for (int k = 0; k < 12; ++k) gout[lane * 12 + k] = s[k];
    }
    if (lane == 0) {
        double* res = a.result + (size_t)inst * 8;
        res[0] = last_plain; res[1] = cur_chi; res[2] = lambda; res[3] = (double)it; res[4] = (double)trials;
        res[5] = (double)terminated; res[6] = (double)shared_edges; res[7] = nv > 0 ? (double)(nv * 65536 + 2 * nv - 1) : 0.0;
#ifdef LOCAMD_WAVE6_TIMING
        for (int k = 0; k < 7; ++k) res[k] = (double)w6_ph[k];
        res[7] = (double)(__builtin_readcyclecounter() - w6_start) + 1e12 * trials;
#endif
    }
}

}  // namespace

size_t window_wave6_lds_bytes(const WindowCaps& c, bool se3) {
    const size_t doubles = (size_t)W6_NSLOT * c.nv_max * 12 + (size_t)c.nr_max * (2 * W6_REC + 5 + 3) + (size_t)c.np_max * (W6_PREC + 18) + (se3 ? 48 * 64 : 0);
    const size_t ints = (size_t)c.nr_max * 4 + 2 * (size_t)c.np_max;
    return doubles * sizeof(double) + ((ints + 1) & ~(size_t)1) * sizeof(int);
}

namespace {
template <int JAC, bool SE3>
hipError_t launch_wave6_t(const WindowArgs& a, size_t lds, hipStream_t stream) {
    static std::atomic<uint64_t> attr_set{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_set.load(std::memory_order_acquire) & bit)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wave6_lm_kernel<JAC, SE3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWave6MaxLds);
        if (e != hipSuccess) return e;
        attr_set.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((wave6_lm_kernel<JAC, SE3>), dim3((unsigned)a.B), dim3(64), lds, stream, a);
    return hipGetLastError();
}
}  // namespace

hipError_t launch_window_wave6(const WindowArgs& a, bool se3, hipStream_t stream) {
    if (a.B <= 0 || a.caps.nv_max > 64 || a.caps.nv_max <= 0 || (se3 && (a.caps.ns_max > 64 || a.caps.nv_max > 63))) return hipErrorInvalidValue;
    const size_t lds = window_wave6_lds_bytes(a.caps, se3);
    if (lds > kWave6MaxLds) return hipErrorInvalidValue;
    if (se3) return a.jacobian ? launch_wave6_t<1, true>(a, lds, stream) : launch_wave6_t<0, true>(a, lds, stream);
    return a.jacobian ? launch_wave6_t<1, false>(a, lds, stream) : launch_wave6_t<0, false>(a, lds, stream);
}

}  // namespace locamd
