// Is device_math.h's sqrt_ieee_unscaled bit-identical to the compiler's IEEE f64 sqrt on everything the numeric-Jacobian paths feed it?
//   hipcc --offload-arch=gfx950 -O3 -I localization_amd/csrc tools/sqrt_probe.hip -o tools/sqrt_probe.bin && tools/sqrt_probe.bin
// and are sqrt_ieee_near / sqrt_ieee_near_c bit-identical to it on the perturbed arguments of g2o's central differences?
// Arguments: 2^26 values log-uniform over [1e-220, 1e220], 2^26 squared distances (dx^2 + dy^2 + dz^2 of ranges 1e-6 .. 1e3 m), and the specials.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include "device_math.h"
using namespace locamd;

__device__ uint64_t rng(uint64_t& s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
__global__ void probe(unsigned long long* mismatches, unsigned long long* worst_bits, int mode) {
    uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x + 1);
    unsigned long long bad = 0;
    for (int i = 0; i < 1024; ++i) {
        double x;
        if (mode == 0) { const double u = (double)(rng(s) >> 11) * (1.0 / 9007199254740992.0); x = exp((u * 440.0 - 220.0) * 2.302585092994046); }
        else { double d2 = 0; for (int k = 0; k < 3; ++k) { const double u = (double)(rng(s) >> 11) * (1.0 / 9007199254740992.0); const double d = exp((u * 9.0 - 6.0) * 2.302585092994046); d2 = d2 + d * d; } x = d2; }
        const double a = sqrt(x), b = sqrt_ieee_unscaled(x);
        if (__double_as_longlong(a) != __double_as_longlong(b)) ++bad;
    }
    atomicAdd(mismatches, bad);
}
// sqrt_ieee_near on the arguments g2o's central differences produce: p, a random points (range n0 log-uniform over [lo, 1e3] m),
// the six squared distances with one coordinate of p moved by +-1e-9, in the plain (un-contracted) evaluation order of the kernels
#pragma clang fp contract(off)
__global__ void probe_near(unsigned long long* mismatches, double lo_log10, unsigned long long seed) {
    uint64_t s = (0xD1B54A32D192ED03ull + 2 * seed) * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x + 1);
    unsigned long long bad = 0;
    for (int i = 0; i < 342; ++i) {
        double p[3], a[3], dir[3], nn = 0;
        for (int k = 0; k < 3; ++k) { dir[k] = (double)(rng(s) >> 11) * (2.0 / 9007199254740992.0) - 1.0; nn += dir[k] * dir[k]; }
        const double u = (double)(rng(s) >> 11) * (1.0 / 9007199254740992.0);
        const double range = exp((u * (3.0 - lo_log10) + lo_log10) * 2.302585092994046) / sqrt(nn);
        for (int k = 0; k < 3; ++k) { a[k] = ((double)(rng(s) >> 11) * (2.0 / 9007199254740992.0) - 1.0) * 20.0; p[k] = a[k] + dir[k] * range; }
        const double dx = p[0] - a[0], dy = p[1] - a[1], dz = p[2] - a[2];
        const double x0 = dx * dx + dy * dy + dz * dz;
        double h0, h0r, c0;
        const double s0 = sqrt_ieee_unscaled_h(x0, h0);
        const double s0r = sqrt_ieee_unscaled_raw(x0, h0r, c0);
        if (__double_as_longlong(s0) != __double_as_longlong(sqrt(x0)) || __double_as_longlong(s0r) != __double_as_longlong(s0)) ++bad;
        for (int ax = 0; ax < 3; ++ax)
            for (int sg = 0; sg < 2; ++sg) {
                double q[3] = {dx, dy, dz};
                q[ax] = (sg ? (p[ax] - 1e-9) : (p[ax] + 1e-9)) - a[ax];
                const double x = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
                const double want = sqrt(x), got = sqrt_ieee_near(x, s0, h0), got_c = sqrt_ieee_near_c(x, s0r, h0r, c0);
                if (__double_as_longlong(want) != __double_as_longlong(got) || __double_as_longlong(want) != __double_as_longlong(got_c)) ++bad;
            }
    }
    atomicAdd(mismatches, bad);
}
#pragma clang fp contract(fast)
__global__ void specials(double* out) {
    const double v[8] = {0.0, -0.0, __builtin_inf(), 1.0, 4.0, 2.0, 1e-300, 2.2250738585072014e-308};
    for (int i = 0; i < 8; ++i) { out[2 * i] = sqrt(v[i]); out[2 * i + 1] = sqrt_ieee_unscaled(v[i]); }
}
int main() {
    unsigned long long *d, h[2] = {0, 0};
    double *o, ho[16];
    hipMalloc((void**)&d, 16); hipMalloc((void**)&o, 128);
    for (int mode = 0; mode < 2; ++mode) {
        hipMemset(d, 0, 16);
        hipLaunchKernelGGL(probe, dim3(256), dim3(256), 0, 0, d, d + 1, mode);
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("%s: %llu mismatches in %llu arguments\n", mode == 0 ? "log-uniform [1e-220, 1e220]" : "squared distances (1e-6 .. 1e3 m)", h[0], 256ull * 256 * 1024);
    }
    hipMemset(d, 0, 16);
    hipLaunchKernelGGL(probe_near, dim3(256), dim3(256), 0, 0, d, -2.5, 0ull);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("sqrt_ieee_near, ranges 3.2e-3 .. 1e3 m, +-1e-9 on one coordinate: %llu mismatches in %llu perturbed arguments\n", h[0], 256ull * 256 * 342 * 6);
    hipMemset(d, 0, 16);
    hipLaunchKernelGGL(probe_near, dim3(256), dim3(256), 0, 0, d, -2.5, 12345ull);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("sqrt_ieee_near, a second 2^27 arguments: %llu mismatches\n", h[0]);
    hipLaunchKernelGGL(specials, dim3(1), dim3(1), 0, 0, o);
    hipMemcpy(ho, o, 128, hipMemcpyDeviceToHost);
    const char* names[8] = {"+0", "-0", "+inf", "1", "4", "2", "1e-300", "DBL_MIN"};
    for (int i = 0; i < 8; ++i) printf("sqrt(%s): builtin %.17g  unscaled %.17g  %s\n", names[i], ho[2 * i], ho[2 * i + 1], memcmp(&ho[2 * i], &ho[2 * i + 1], 8) == 0 ? "same bits" : "DIFFERENT");
    return 0;
}
