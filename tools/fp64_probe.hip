// Standalone probe: sustained f64 FMA / ADD / MUL rate of the vector ALU on this GPU, with 1, 2, 4, 8 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o fp64_probe fp64_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void __launch_bounds__(256) k(double* out, int iters, double a, double b) {
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == 0) { x0 = __builtin_fma(x0, a, b); x1 = __builtin_fma(x1, a, b); x2 = __builtin_fma(x2, a, b); x3 = __builtin_fma(x3, a, b);
                           x4 = __builtin_fma(x4, a, b); x5 = __builtin_fma(x5, a, b); x6 = __builtin_fma(x6, a, b); x7 = __builtin_fma(x7, a, b); }
            if (OP == 1) { x0 += a; x1 += a; x2 += a; x3 += a; x4 += a; x5 += a; x6 += a; x7 += a; }
            if (OP == 2) { x0 *= a; x1 *= a; x2 *= a; x3 *= a; x4 *= a; x5 *= a; x6 *= a; x7 *= a; }
            if (OP == 3) { float y0 = x0, y1 = x1; y0 = __builtin_fmaf(y0, (float)a, (float)b); y1 = __builtin_fmaf(y1, (float)a, (float)b); x0 = y0; x1 = y1; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int OP>
void run(const char* name, int blocks, double flop_per_op) {
    double* out; hipMalloc(&out, (size_t)blocks * 256 * 8);
    const int iters = 4096;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(out, 16, 1.0000001, 1e-9);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, iters, 1.0000001, 1e-9);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 256 * iters * 64;  // 8 chains x 8 unroll
    printf("%-8s blocks %5d (%.0f waves/SIMD): %.3f ms  %.2f T lane-ops/s  %.2f TFLOP/s  => %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, blocks,
           blocks * 4.0 / 1024.0, ms, ops / (ms * 1e-3) / 1e12, flop_per_op * ops / (ms * 1e-3) / 1e12,
           (ms * 1e-3 * 2.4e9) / ((double)iters * 64 * (blocks * 4.0 / 1024.0)));
    hipFree(out);
}
int main() {
    for (int b : {256, 512, 1024, 2048}) run<0>("fma_f64", b, 2);
    for (int b : {256, 1024}) run<1>("add_f64", b, 1);
    for (int b : {256, 1024}) run<2>("mul_f64", b, 1);
    return 0;
}
