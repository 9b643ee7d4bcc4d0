// Standalone accuracy probe for fast_log_ge1 (localization_amd/csrc/device_math.h) on the device; not part of the library.
// build: hipcc --offload-arch=gfx950 -O3 -I localization_amd/csrc -o tools/log_probe tools/log_probe.hip ; run on the GPU box.
#include "device_math.h"
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
__global__ void probe(const double* x, double* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { out[2 * i] = locamd::fast_log_ge1(x[i]); out[2 * i + 1] = log(x[i]); }
}
int main() {
    const int n = 1 << 22;
    std::vector<double> x(n);
    std::mt19937_64 rng(3);
    for (int pass = 0; pass < 3; ++pass) {
        const double hi = pass == 0 ? 1.0 : (pass == 1 ? 40.0 : 1000.0);   // exponents: [1,2], [1,2^40], [1,2^1000]
        std::uniform_real_distribution<double> u(0.0, hi);
        for (auto& v : x) v = std::exp2(u(rng));
        double *dx, *dout;
        hipMalloc(&dx, n * 8); hipMalloc(&dout, (size_t)n * 16);
        hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
        probe<<<n / 256, 256>>>(dx, dout, n);
        std::vector<double> o((size_t)n * 2);
        hipMemcpy(o.data(), dout, (size_t)n * 16, hipMemcpyDeviceToHost);
        double mx_fast = 0, mx_lib = 0;
        for (int i = 0; i < n; ++i) {
            const long double ref = logl((long double)x[i]);
            const double ulp = std::nextafter(std::fabs((double)ref), INFINITY) - std::fabs((double)ref);
            mx_fast = std::fmax(mx_fast, (double)fabsl(o[2 * i] - ref) / ulp);
            mx_lib = std::fmax(mx_lib, (double)fabsl(o[2 * i + 1] - ref) / ulp);
        }
        printf("x in [1, 2^%g]: fast_log_ge1 max err %.3f ulp, library log %.3f ulp\n", hi, mx_fast, mx_lib);
        hipFree(dx); hipFree(dout);
    }
    double h[4] = {INFINITY, NAN, 1.0, 1.7976931348623157e308}, *dx, *dout, o[8];
    hipMalloc(&dx, 32); hipMalloc(&dout, 64);
    hipMemcpy(dx, h, 32, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dx, dout, 4);
    hipMemcpy(o, dout, 64, hipMemcpyDeviceToHost);
    printf("specials: log(inf) = %g, log(nan) = %g, log(1) = %g, log(DBL_MAX) = %.17g (library %.17g)\n", o[0], o[2], o[4], o[6], o[7]);
    return 0;
}
