// Device helpers shared by the window kernels (window_kernel.hip: the general sparse solver; chain_kernel.hip: one lane per chain
// window; tree_kernel.hip: forest windows of one shared topology): wave / block reductions on the VALU, 3x3 / quaternion algebra,
// g2o's numeric range Jacobian column, the pivot reciprocal square root.  Everything is internal to the including translation unit.
#pragma once
#include "window_kernel.h"
#include "device_math.h"
#include "numeric_jacobian.h"

#include <float.h>
#include <math.h>

namespace locamd {
namespace {

// Wave-wide reductions on the VALU (DPP row shifts + row broadcasts, then one readlane): every lane gets the same
// bits, no LDS crossbar round trips (a __shfl_xor butterfly on doubles costs ~6 dependent ds_bpermute pairs).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_or_zero(double v, double identity) {
    const int ilo = __double2loint(identity), ihi = __double2hiint(identity);
    const int lo = __builtin_amdgcn_update_dpp(ilo, __double2loint(v), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(ihi, __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double read_lane63(double v) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_or_zero<0x111, 0xF>(v, 0.0);  // row_shr:1
    v += dpp_or_zero<0x112, 0xF>(v, 0.0);  // row_shr:2
    v += dpp_or_zero<0x114, 0xF>(v, 0.0);  // row_shr:4
    v += dpp_or_zero<0x118, 0xF>(v, 0.0);  // row_shr:8  -> lane 15 of each row holds the row sum
    v += dpp_or_zero<0x142, 0xA>(v, 0.0);  // row_bcast:15 into rows 1 and 3
    v += dpp_or_zero<0x143, 0xC>(v, 0.0);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return read_lane63(v);
}
__device__ __forceinline__ double wave_max(double v) {  // for non-negative inputs (identity 0)
    v = fmax(v, dpp_or_zero<0x111, 0xF>(v, 0.0));
    v = fmax(v, dpp_or_zero<0x112, 0xF>(v, 0.0));
    v = fmax(v, dpp_or_zero<0x114, 0xF>(v, 0.0));
    v = fmax(v, dpp_or_zero<0x118, 0xF>(v, 0.0));
    v = fmax(v, dpp_or_zero<0x142, 0xA>(v, 0.0));
    v = fmax(v, dpp_or_zero<0x143, 0xC>(v, 0.0));
    return read_lane63(v);
}

// ---- several waves per window (NW > 1: windows of 65 .. 512 poses, whose structure tables fill a CU's LDS so that ONE window
//      runs per CU — with one wave it used a quarter of one SIMD pair's issue slots and left three SIMDs idle) ------------------
// SOLO: a section executed by wave 0 alone (the set-up) orders its own memory operations without the workgroup barrier.
template <bool SOLO>
__device__ __forceinline__ void sync_() {
    if (SOLO) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    } else {
        __syncthreads();
    }
}
// Reductions over the NW waves of a window through `red` (NW doubles of LDS): every thread gets the same bits (the waves'
// partial results are combined in wave order).  Two barriers each; all threads must call.
template <int NW>
__device__ __forceinline__ double block_sum(double v, double* red, int tid) {
    const double w = wave_sum(v);
    if (NW == 1) return w;
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = w;
    __syncthreads();
    double t = red[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) t += red[i];
    return t;
}
template <int NW>
__device__ __forceinline__ double block_max(double v, double* red, int tid) {  // non-negative inputs
    const double w = wave_max(v);
    if (NW == 1) return w;
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = w;
    __syncthreads();
    double t = red[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) t = fmax(t, red[i]);
    return t;
}
template <int NW>
__device__ __forceinline__ bool block_any(bool pred, double* red, int tid) {
    const bool w = __ballot(pred) != 0;
    if (NW == 1) return w;
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = w ? 1.0 : 0.0;
    __syncthreads();
    bool t = false;
#pragma unroll
    for (int i = 0; i < NW; ++i) t = t || red[i] != 0.0;
    return t;
}

// ---- small SE3 algebra (row-major 3x3) -------------------------------------------------------------------------
__device__ __forceinline__ void mat_mul(const double* A, const double* B, double* C) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j] + A[i * 3 + 2] * B[2 * 3 + j];
}
__device__ __forceinline__ void mat_tmul(const double* A, const double* B, double* C) {  // A^T B
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[0 * 3 + i] * B[0 * 3 + j] + A[1 * 3 + i] * B[1 * 3 + j] + A[2 * 3 + i] * B[2 * 3 + j];
}
__device__ __forceinline__ void mat_vec(const double* A, const double* v, double* o) {
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = A[i * 3 + 0] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2];
}
__device__ __forceinline__ void mat_tvec(const double* A, const double* v, double* o) {  // A^T v
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = A[0 * 3 + i] * v[0] + A[1 * 3 + i] * v[1] + A[2 * 3 + i] * v[2];
}
// Eigen::Quaternion(Matrix3) — q = (w, x, y, z).  Scalars, not an array: the optimiser otherwise merges the branches
// into a computed index and the array lands in scratch memory.
__device__ __forceinline__ void mat_to_quat(const double* R, double* q) {
    double qw, qx, qy, qz;
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        qw = 0.5 * t; t = 0.5 / t;
        qx = (R[7] - R[5]) * t; qy = (R[2] - R[6]) * t; qz = (R[3] - R[1]) * t;
    } else if (R[0] >= R[4] && R[0] >= R[8]) {  // i = 0, j = 1, k = 2
        t = sqrt(R[0] - R[4] - R[8] + 1.0);
        qx = 0.5 * t; t = 0.5 / t;
        qw = (R[7] - R[5]) * t; qy = (R[3] + R[1]) * t; qz = (R[6] + R[2]) * t;
    } else if (R[4] > R[0] && R[4] >= R[8]) {   // i = 1, j = 2, k = 0
        t = sqrt(R[4] - R[8] - R[0] + 1.0);
        qy = 0.5 * t; t = 0.5 / t;
        qw = (R[2] - R[6]) * t; qz = (R[7] + R[5]) * t; qx = (R[1] + R[3]) * t;
    } else {                                      // i = 2, j = 0, k = 1
        t = sqrt(R[8] - R[0] - R[4] + 1.0);
        qz = 0.5 * t; t = 0.5 / t;
        qw = (R[3] - R[1]) * t; qx = (R[2] + R[6]) * t; qy = (R[5] + R[7]) * t;
    }
    q[0] = qw; q[1] = qx; q[2] = qy; q[3] = qz;
}
__device__ __forceinline__ void quat_mul(const double* a, const double* b, double* o) {
    o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}
// g2o internal::normalize: unit norm, w >= 0; returns the sign applied
__device__ __forceinline__ double quat_normalize_sign(double* q) {
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    double s = 1.0 / n, sg = 1.0;
    if (q[0] < 0) { s = -s; sg = -1.0; }
    q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s;
    return sg;
}
// Eigen toRotationMatrix (no normalisation)
__device__ __forceinline__ void quat_to_mat(const double* q, double* R) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
// rows 3..5 x cols 3..5 of a Jacobian: d vec(q (x) (sqrt(1-|v|^2), v)) / dv at 0 = w I + [q_xyz]x
__device__ __forceinline__ void quat_right_jac(const double* q, double sgn, double* J, int ldj) {
    const double w = q[0] * sgn, x = q[1] * sgn, y = q[2] * sgn, z = q[3] * sgn;
    J[3 * ldj + 3] = w;  J[3 * ldj + 4] = -z; J[3 * ldj + 5] = y;
    J[4 * ldj + 3] = z;  J[4 * ldj + 4] = w;  J[4 * ldj + 5] = -x;
    J[5 * ldj + 3] = -y; J[5 * ldj + 4] = x;  J[5 * ldj + 5] = w;
}

typedef unsigned long long u64;

#pragma clang fp contract(off)
// one column of the numeric Jacobian of endpoint `which` (0: the pose carrying the lever arm, 1: the other pose)
template <int D>
__device__ __forceinline__ double range_jac_numeric(const double* X0, const double* off, const double* X1, const double* q1, int which, double meas,
                                                    const double* off1 = nullptr) {
    // (the textbook form of the increment here: these kernels sit at 512 registers, and the short form of numeric_jacobian.h — more
    //  values shared between the twelve evaluations — costs tree_wave_kernel 112 B more scratch per lane and 10 % of its time)
    constexpr double delta = 1e-9;
    constexpr double scalar = 1.0 / (2 * delta);
    double Rp[9], tp[3], Rm[9], tm[3];
    double ep, em;
    if (which == 0) {
        oplus_axis_plain_reference<D>(X0, X0 + 9, delta, Rp, tp);
        oplus_axis_plain_reference<D>(X0, X0 + 9, -delta, Rm, tm);
        ep = range_error_plain(Rp, tp, off, q1, meas);
        em = range_error_plain(Rm, tm, off, q1, meas);
    } else if (off1) {   // endpoint 1 carries a lever arm too: its point is (X1 * fromVectorMQT(+-delta e_D)) * o1
        perturbed_point_plain<D>(X1, X1 + 9, off1, delta, tp);
        perturbed_point_plain<D>(X1, X1 + 9, off1, -delta, tm);
        ep = range_error_plain(X0, X0 + 9, off, tp, meas);
        em = range_error_plain(X0, X0 + 9, off, tm, meas);
    } else {
        oplus_axis_plain_reference<D>(X1, X1 + 9, delta, Rp, tp);   // endpoint 1 has no lever arm: its point is its translation
        oplus_axis_plain_reference<D>(X1, X1 + 9, -delta, Rm, tm);
        ep = range_error_plain(X0, X0 + 9, off, tp, meas);
        em = range_error_plain(X0, X0 + 9, off, tm, meas);
    }
    double bak = ep;
    bak -= em;
    return scalar * bak;
}
#pragma clang fp contract(fast)

// 1/sqrt(d) for a pivot d > 0: hardware seed (~2^-24) + one third-order step y (1 + e/2 + 3 e^2/8), e = 1 - d y^2: the error
// term e^3 is far below an ulp; four dependent operations after the seed (the Goldschmidt pair + Newton used elsewhere: eight)
__device__ __forceinline__ double pivot_rsqrt(double d) {
    const double y = __builtin_amdgcn_rsq(d);
    const double t = d * y;
    const double e = __builtin_fma(-t, y, 1.0);
    const double pq = __builtin_fma(0.375, e, 0.5);
    const double ye = y * e;
    return __builtin_fma(ye, pq, y);
}

}  // namespace
}  // namespace locamd
