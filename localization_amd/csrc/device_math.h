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

}  // namespace
}  // namespace locamd
