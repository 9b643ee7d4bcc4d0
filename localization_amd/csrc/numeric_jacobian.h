// g2o's numeric Jacobian of EdgeSE3Range, shared by the window and fusion kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "device_math.h"

namespace locamd {
namespace {

// ---- g2o's numeric Jacobian of EdgeSE3Range (BaseBinaryEdge::linearizeOplus) ---------------------------
// The reference has no linearizeOplus for EdgeSE3Range (types_edge_se3range.h:45-74), so g2o differentiates
// computeError() (types_edge_se3range.cpp:105-114) by central differences, delta = 1e-9, through VertexSE3::oplus
// (X <- X * fromVectorMQT(+-delta e_d)).  The few operations of that evaluation are kept un-contracted and on the IEEE
// sqrt, in the operation order of a plain CPU build, so the difference quotient (which multiplies every last-bit
// difference by 5e8) sees the same roundings.
#pragma clang fp contract(off)
__device__ __forceinline__ double range_error_plain(const double* R, const double* t, const double* off, const double* q1, double meas) {
    const double px = R[0] * off[0] + R[1] * off[1] + R[2] * off[2];
    const double py = R[3] * off[0] + R[4] * off[1] + R[5] * off[2];
    const double pz = R[6] * off[0] + R[7] * off[1] + R[8] * off[2];
    const double dx = (px + t[0]) - q1[0], dy = (py + t[1]) - q1[1], dz = (pz + t[2]) - q1[2];
    return meas - sqrt_ieee_unscaled(dx * dx + dy * dy + dz * dz);
}
// X * fromVectorMQT(dl e_D): (R Rinc, R tinc + t) — the textbook evaluation, every product with the increment's zeros and ones spelled out
// (kept as the definition; tools/oplus_probe.hip checks oplus_axis_plain against it bit for bit)
template <int D>
__device__ __forceinline__ void oplus_axis_plain_reference(const double* R, const double* t, double dl, double* Ro, double* to) {
    double v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    v[D] = dl;
    double Ri[9];
    const double w = 1.0 - (v[3] * v[3] + v[4] * v[4] + v[5] * v[5]);
    if (w < 0) { Ri[0] = 1; Ri[1] = 0; Ri[2] = 0; Ri[3] = 0; Ri[4] = 1; Ri[5] = 0; Ri[6] = 0; Ri[7] = 0; Ri[8] = 1; }
    else {
        const double qw = sqrt(w), qx = v[3], qy = v[4], qz = v[5];
        const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
        const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
        Ri[0] = 1 - (tyy + tzz); Ri[1] = txy - twz; Ri[2] = txz + twy;
        Ri[3] = txy + twz; Ri[4] = 1 - (txx + tzz); Ri[5] = tyz - twx;
        Ri[6] = txz - twy; Ri[7] = tyz + twx; Ri[8] = 1 - (txx + tyy);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) Ro[i * 3 + j] = R[i * 3 + 0] * Ri[0 * 3 + j] + R[i * 3 + 1] * Ri[1 * 3 + j] + R[i * 3 + 2] * Ri[2 * 3 + j];
        to[i] = (R[i * 3 + 0] * v[0] + R[i * 3 + 1] * v[1] + R[i * 3 + 2] * v[2]) + t[i];
    }
}
// The same numbers without the arithmetic on exact zeros and ones.  The increment of g2o's central differences has ONE non-zero entry,
// dl = +-1e-9: for a translation axis Rinc is exactly I and tinc = dl e_D, so R' = R and t'_i = fl(fl(R_iD dl) + t_i); for a rotation
// axis a = D - 3 the quaternion is (fl(sqrt(fl(1 - dl^2))), dl e_a) = (1, dl e_a) exactly (dl^2 = 1e-18 is below half an ulp of 1), so
// Rinc = I + [2 dl e_a]x with exact entries 1, 0, +-2 dl (1 - 2 dl^2 rounds to 1 as well), R' differs from R in the two columns other
// than a — each entry one multiplication and one addition, in the textbook's order — and t' = t.  (Products with +-0 only ever add a
// signed zero to a sum; the one thing that can differ is the SIGN of a result that is exactly zero.)  A perturbed rotation costs 12
// operations instead of 60, a perturbed translation 6.
template <int D>
__device__ __forceinline__ void oplus_axis_plain(const double* R, const double* t, double dl, double* Ro, double* to) {
    static_assert(D >= 0 && D < 6, "axis");
    if (D < 3) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) Ro[i * 3 + j] = R[i * 3 + j];
            to[i] = R[i * 3 + D] * dl + t[i];
        }
    } else {
        const double tw = 2 * dl;   // (t_a q_w with q_w = 1)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double r0 = R[i * 3 + 0], r1 = R[i * 3 + 1], r2 = R[i * 3 + 2];
            if (D == 3) {          // Rinc = [[1, 0, 0], [0, 1, -tw], [0, tw, 1]]
                Ro[i * 3 + 0] = r0; Ro[i * 3 + 1] = r1 + r2 * tw; Ro[i * 3 + 2] = r1 * -tw + r2;
            } else if (D == 4) {   // Rinc = [[1, 0, tw], [0, 1, 0], [-tw, 0, 1]]
                Ro[i * 3 + 0] = r0 + r2 * -tw; Ro[i * 3 + 1] = r1; Ro[i * 3 + 2] = r0 * tw + r2;
            } else {               // Rinc = [[1, -tw, 0], [tw, 1, 0], [0, 0, 1]]
                Ro[i * 3 + 0] = r0 + r1 * tw; Ro[i * 3 + 1] = r0 * -tw + r1; Ro[i * 3 + 2] = r2;
            }
            to[i] = t[i];
        }
    }
}
// the lever-arm point (X * fromVectorMQT(dl e_D)) * o = R' o + t'
template <int D>
__device__ __forceinline__ void perturbed_point_plain(const double* R, const double* t, const double* off, double dl, double* P) {
    double Ro[9], to[3];
    oplus_axis_plain<D>(R, t, dl, Ro, to);
#pragma unroll
    for (int i = 0; i < 3; ++i) P[i] = (Ro[i * 3 + 0] * off[0] + Ro[i * 3 + 1] * off[1] + Ro[i * 3 + 2] * off[2]) + to[i];
}
__device__ __forceinline__ double norm_to_plain(const double* P, double ax, double ay, double az) {
    const double dx = P[0] - ax, dy = P[1] - ay, dz = P[2] - az;
    return sqrt_ieee_unscaled(dx * dx + dy * dy + dz * dz);
}
// the squared distance of norm_to_plain (same operation order), and the norm of a PERTURBED point from the central one's
// (device_math.h: sqrt_ieee_near — the same correctly rounded number; n0 = norm_to_plain of the unperturbed point, h0 ~ 0.5 / n0)
__device__ __forceinline__ double sq_to_plain(const double* P, double ax, double ay, double az) {
    const double dx = P[0] - ax, dy = P[1] - ay, dz = P[2] - az;
    return dx * dx + dy * dy + dz * dz;
}
// J = ((meas - n+) - (meas - n-)) / (2 delta), g2o's operation order
__device__ __forceinline__ double central_difference_plain(double meas, double np, double nm) {
    constexpr double delta = 1e-9;
    constexpr double scalar = 1.0 / (2 * delta);
    double bak = meas - np;
    bak -= meas - nm;
    return scalar * bak;
}
#pragma clang fp contract(fast)

}  // namespace
}  // namespace locamd
