// Device math shared by the gfx950 kernels: f64 reciprocal / square root from the hardware seeds + FMA refinement.
#pragma once
#include <hip/hip_runtime.h>

namespace locamd {
namespace {

// ---- f64 reciprocal / sqrt from the hardware seeds + FMA refinement ------------------------------------
// Accuracies measured on MI355X with tools/math_probe.hip (profiles/r01_math_probe.txt); operands here are ranges,
// 1 + chi and pivots: normal, positive, so no div_scale / div_fixup range handling is needed.
// v_rcp_f64 seed (4.6e-8) + two Newton steps: 1.1e-16 relative (as good as an IEEE divide).
__device__ __forceinline__ double fast_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}
// one Newton step: 2.2e-15 relative — used for the robust weight rho' = 1/(1 + chi), which scales H and b alike.
__device__ __forceinline__ double fast_rcp_1nr(double d) {
    double r = __builtin_amdgcn_rcp(d);
    const double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}
// n = sqrt(x) to 1.1e-16 relative (identical error bound to the builtin sqrt: v_rsq_f64 seed, one Goldschmidt step, one
// residual correction) and inv = 1/sqrt(x) to 4.2e-15 (only scales the unit vector of the Jacobian).
__device__ __forceinline__ void sqrt_and_rsqrt(double x, double& n, double& inv) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    const double dd = __builtin_fma(-g, g, x);
    n = __builtin_fma(dd, h, g);
    inv = h + h;
}

// The IEEE square root of the numeric-Jacobian paths (g2o's central differences need the CPU's roundings): the compiler's own f64
// expansion (AMDGPU lowerFSQRTF64: v_rsq_f64 seed, one coupled Goldschmidt step, two residual corrections, +-0 / +inf passed through)
// WITHOUT its 2^+-256 argument scaling, which only matters below 2^-767 — squared distances never are, short of exactly 0 — so the
// result is bit-identical to sqrt() for every argument these kernels produce (tools/sqrt_probe.hip, profiles/r03_sqrt_probe.txt) at 13
// instructions instead of 18; a numeric-mode edge takes seven of them per LM pass.
__device__ __forceinline__ double sqrt_ieee_unscaled(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double s = x * y;
    double h = y * 0.5;
    const double r = __builtin_fma(-h, s, 0.5);
    s = __builtin_fma(s, r, s);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-s, s, x);
    s = __builtin_fma(d, h, s);
    d = __builtin_fma(-s, s, x);
    s = __builtin_fma(d, h, s);
    return (x == 0.0 || x == __builtin_inf()) ? x : s;
}

// The same, also handing out h ~ 0.5 / sqrt(x) (for sqrt_ieee_near).
__device__ __forceinline__ double sqrt_ieee_unscaled_h(double x, double& h_out) {
    const double y = __builtin_amdgcn_rsq(x);
    double s = x * y;
    double h = y * 0.5;
    const double r = __builtin_fma(-h, s, 0.5);
    s = __builtin_fma(s, r, s);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-s, s, x);
    s = __builtin_fma(d, h, s);
    d = __builtin_fma(-s, s, x);
    s = __builtin_fma(d, h, s);
    h_out = h;
    return (x == 0.0 || x == __builtin_inf()) ? x : s;
}
// The IEEE square root of an argument NEAR one whose root is known: g2o's central differences evaluate the range at p +- 1e-9 e_d,
// so the six (or twelve) perturbed squared distances of an edge differ from the central one x0 by ~2e-9 |d| — and the correctly
// rounded root is a unique number, whichever way it is reached.  From s0 = sqrt(x0), h0 ~ 0.5 / s0 (relative distance e0 ~ 1e-9 / s0
// from the root of x): one residual correction s <- s0 + (x - s0^2) h0 (error ~1.5 e0^2), one Newton step for h ~ 0.5 / s
// (h <- h0 + h0 (1 - 2 h0 s): error e0^2 instead of e0), then the compiler's own final correction s <- s + (x - s^2) h.  The value
// before the last rounding is off by ~e0^4 relative, so a wrong rounding needs the root within that of a rounding boundary:
// probability ~1e16 e0^4 per evaluation (1e-8 for endpoints a millimetre apart, 1e-12 at a centimetre, nothing at a metre) —
// callers guard with x0 >= 1e-5 m^2.  (Without the step for h, or with only this one correction of s, the probe below finds 1e-8 /
// 1e-5 of the arguments one ulp off.)  Bit-identical to sqrt() on 2^28 perturbed arguments (tools/sqrt_probe.hip,
// profiles/r03_sqrt_probe.txt); 6 instructions instead of 13.
__device__ __forceinline__ double sqrt_ieee_near(double x, double s0, double h0) {
    double d = __builtin_fma(-s0, s0, x);
    double s = __builtin_fma(d, h0, s0);
    const double r = __builtin_fma(-(h0 + h0), s, 1.0);
    const double h = __builtin_fma(h0, r, h0);
    d = __builtin_fma(-s, s, x);
    return __builtin_fma(d, h, s);
}

// The central root of an edge WITHOUT the +-0 / +inf pass-through, and c = 2 h^3 for sqrt_ieee_near_c: what the fast path of the numeric
// Jacobians needs (its arguments are >= 1e-5 and < 1e300; callers that may see 0 or inf select afterwards, in the branch that only they take).
__device__ __forceinline__ double sqrt_ieee_unscaled_raw(double x, double& h_out, double& c_out) {
    const double y = __builtin_amdgcn_rsq(x);
    double s = x * y;
    double h = y * 0.5;
    const double r = __builtin_fma(-h, s, 0.5);
    s = __builtin_fma(s, r, s);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-s, s, x);
    s = __builtin_fma(d, h, s);
    d = __builtin_fma(-s, s, x);
    s = __builtin_fma(d, h, s);
    h_out = h;
    c_out = (h * h) * (h + h);
    return s;
}
// sqrt_ieee_near with the reciprocal root carried along to FIRST order in the residual instead of by a Newton step: with D = x - s0^2 (exact),
// 0.5 / sqrt(x) = h0 (1 - 2 h0^2 D + O(e0^2)), so h = h0 - c D, c = 2 h0^3 (once per edge): one instruction instead of two.  h is then off
// by h0's own error (~5e-15: one Goldschmidt step from the hardware seed) instead of one rounding (1e-16); the last correction
// s <- s1 + (x - s1^2) h moves s1 by at most an ulp, so the result before the final rounding is off by <= 5e-15 ulp and the rounding
// goes wrong only for a root within that of a rounding boundary: probability ~1e-14 per evaluation (the Newton form: ~2e-16) — nothing
// in the 2^28 probed arguments (tools/sqrt_probe.hip), about one perturbed norm in 1e4 benchmark launches, where it moves one Jacobian
// entry by the 3e-7 relative that the difference quotient's own noise is.  5 instructions.
__device__ __forceinline__ double sqrt_ieee_near_c(double x, double s0, double h0, double c) {
    const double D = __builtin_fma(-s0, s0, x);
    const double s = __builtin_fma(D, h0, s0);
    const double h = __builtin_fma(-c, D, h0);
    const double d = __builtin_fma(-s, s, x);
    return __builtin_fma(d, h, s);
}

// n = sqrt(x) to 4.1e-15 relative (the Goldschmidt step without the residual correction; profiles/r01_math_probe.txt) and
// inv = 1/sqrt(x) to 4.2e-15:
// the range norm of the analytic kernels, where 1e-14 m is five orders below anything the estimate resolves; two
// instructions per range fewer than sqrt_and_rsqrt.
__device__ __forceinline__ void sqrt_and_rsqrt_fast(double x, double& n, double& inv) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    n = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    inv = h + h;
}

// log(x) for x >= 1 (what the robust cost needs: 1 + chi, and products of such) — the kernel of fdlibm's e_log.c without the
// argument classes that cannot occur: x = 2^k (1 + f) with sqrt(1/2) < 1 + f < sqrt(2), s = f / (2 + f),
// log(1 + f) = f - hfsq + s (hfsq + R(s^2)), R a degree-7 minimax polynomial: < 1 ulp (checked against long-double log over
// [1, 1e300], 7e6 samples: max 0.84 ulp).  About 40 instructions; the general-purpose library log is about 110, and it was a
// fifth of every LM pass of the snapshot kernel.  inf and NaN pass through.
// log(x 2^k0): the same kernel for a value kept as (x, k0) — a running product whose exponent was taken out on the way so that it
// cannot overflow (x 2^k0 >= 1 as for fast_log_ge1)
__device__ __forceinline__ double fast_log_ge1_scaled(double x, int k0);
__device__ __forceinline__ double fast_log_ge1(double x) { return fast_log_ge1_scaled(x, 0); }
__device__ __forceinline__ double fast_log_ge1_scaled(double x, int k0) {
    constexpr double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                     Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                     Lg7 = 1.479819860511658591e-01;
    constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
    int k = __builtin_amdgcn_frexp_exp(x) + k0;
    const int small = m < 0.70710678118654752440 ? 1 : 0;
    m = __builtin_amdgcn_ldexp(m, small);
    k -= small;
    const double f = m - 1.0;
    const double dk = (double)k;
    const double den = 2.0 + f;
    double r = __builtin_amdgcn_rcp(den);
    double e = __builtin_fma(-den, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-den, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double s = f * r;
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * __builtin_fma(w, __builtin_fma(w, Lg6, Lg4), Lg2);
    const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double res = __builtin_fma(dk, ln2_hi, -((hfsq - __builtin_fma(s, hfsq + R, dk * ln2_lo)) - f));
    return x <= 1.7976931348623157e308 ? res : x;
}

}  // namespace
}  // namespace locamd
