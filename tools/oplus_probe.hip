// Is numeric_jacobian.h's oplus_axis_plain (the single-axis increment of g2o's central differences without the arithmetic on exact
// zeros and ones) the SAME numbers as the textbook evaluation oplus_axis_plain_reference?
//   hipcc --offload-arch=gfx950 -O3 -I localization_amd/csrc tools/oplus_probe.hip -o tools/oplus_probe.bin && tools/oplus_probe.bin
// 2^22 random poses (rotations from random unit quaternions, the identity, axis-aligned quarter turns; translations up to 100 m) x six
// axes x both signs; values are compared with ==, so a result that is exactly zero may differ in its sign (reported separately).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include "numeric_jacobian.h"
using namespace locamd;

__device__ uint64_t rng(uint64_t& s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
__device__ double uni(uint64_t& s) { return (double)(rng(s) >> 11) * (1.0 / 9007199254740992.0); }

template <int D>
__device__ void check(const double* R, const double* t, unsigned long long& bad, unsigned long long& zsign) {
    for (int sg = 0; sg < 2; ++sg) {
        const double dl = sg ? -1e-9 : 1e-9;
        double Ra[9], ta[3], Rb[9], tb[3];
        oplus_axis_plain_reference<D>(R, t, dl, Ra, ta);
        oplus_axis_plain<D>(R, t, dl, Rb, tb);
        for (int k = 0; k < 12; ++k) {
            const double a = k < 9 ? Ra[k] : ta[k - 9], b = k < 9 ? Rb[k] : tb[k - 9];
            if (!(a == b)) ++bad;
            else if (__double_as_longlong(a) != __double_as_longlong(b)) ++zsign;
        }
    }
}
__global__ void probe(unsigned long long* out) {
    uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x + 1);
    unsigned long long bad = 0, zsign = 0;
    for (int i = 0; i < 64; ++i) {
        double R[9], t[3];
        const int kind = (int)(rng(s) % 8);
        double q[4];
        if (kind == 0) { q[0] = 1; q[1] = 0; q[2] = 0; q[3] = 0; }                                // the identity (what the node starts from)
        else if (kind == 1) { const double h = 0.70710678118654752440; q[0] = h; q[1] = 0; q[2] = 0; q[3] = h; }   // a quarter turn about z
        else { double n = 0; for (int k = 0; k < 4; ++k) { q[k] = 2 * uni(s) - 1; n += q[k] * q[k]; } n = 1 / sqrt(n); for (int k = 0; k < 4; ++k) q[k] *= n; }
        const double w = q[0], x = q[1], y = q[2], z = q[3];
        R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
        R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
        R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
        for (int k = 0; k < 3; ++k) t[k] = (2 * uni(s) - 1) * (kind == 2 ? 1e-3 : 100.0);
        check<0>(R, t, bad, zsign); check<1>(R, t, bad, zsign); check<2>(R, t, bad, zsign);
        check<3>(R, t, bad, zsign); check<4>(R, t, bad, zsign); check<5>(R, t, bad, zsign);
    }
    atomicAdd(out, bad);
    atomicAdd(out + 1, zsign);
}
int main() {
    unsigned long long *d, h[2] = {0, 0};
    if (hipMalloc((void**)&d, 16) != hipSuccess || hipMemset(d, 0, 16) != hipSuccess) return 1;
    hipLaunchKernelGGL(probe, dim3(256), dim3(256), 0, 0, d);
    if (hipMemcpy(h, d, 16, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    printf("oplus_axis_plain vs the textbook evaluation: %llu mismatches, %llu results equal but of the other zero sign, in %llu entries\n", h[0], h[1],
           256ull * 256 * 64 * 6 * 2 * 12);
    return h[0] ? 2 : 0;
}
