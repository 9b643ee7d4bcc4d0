// Standalone accuracy probe for the f64 seed+refinement sequences used by the snapshot kernel (not part of the library).
// build: hipcc --offload-arch=gfx950 -O3 -o math_probe math_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
__global__ void probe(const double* x, double* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = x[i];
    double r0 = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r0, 1.0);
    double r1 = __builtin_fma(r0, e, r0);
    e = __builtin_fma(-d, r1, 1.0);
    double r2 = __builtin_fma(r1, e, r1);
    double y = __builtin_amdgcn_rsq(d);
    double g = d * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    double g0 = g;
    double dd = __builtin_fma(-g, g, d);
    double g1 = __builtin_fma(dd, h, g);
    dd = __builtin_fma(-g1, g1, d);
    double g2 = __builtin_fma(dd, h, g1);
    double i0 = h + h;
    double ee = __builtin_fma(-g1, i0, 1.0);
    double i1 = __builtin_fma(i0, ee, i0);
    double* o = out + (size_t)i * 10;
    o[0] = r0; o[1] = r1; o[2] = r2; o[3] = y; o[4] = g0; o[5] = g1; o[6] = g2; o[7] = i0; o[8] = i1; o[9] = sqrt(d);
}
int main() {
    const int n = 1 << 22;
    std::vector<double> x(n);
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> u(-20.0, 20.0);
    for (auto& v : x) v = std::exp2(u(rng));
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, (size_t)n * 80);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    probe<<<n / 256, 256>>>(dx, dout, n);
    std::vector<double> o((size_t)n * 10);
    hipMemcpy(o.data(), dout, (size_t)n * 80, hipMemcpyDeviceToHost);
    const char* names[10] = {"rcp seed", "rcp 1NR", "rcp 2NR", "rsq seed", "sqrt goldschmidt", "sqrt +1corr", "sqrt +2corr", "rsqrt 2h", "rsqrt 2h+NR", "sqrt() builtin"};
    double mx[10] = {0};
    for (int i = 0; i < n; ++i) {
        long double d = x[i];
        long double ref[10] = {1 / d, 1 / d, 1 / d, 1 / sqrtl(d), sqrtl(d), sqrtl(d), sqrtl(d), 1 / sqrtl(d), 1 / sqrtl(d), sqrtl(d)};
        for (int j = 0; j < 10; ++j) {
            double rel = (double)fabsl((o[(size_t)i * 10 + j] - ref[j]) / ref[j]);
            if (rel > mx[j]) mx[j] = rel;
        }
    }
    for (int j = 0; j < 10; ++j) printf("%-18s max rel err %.3e  (%.2f ulp of 2^-53)\n", names[j], mx[j], mx[j] / 1.1102230246251565e-16);
    return 0;
}
