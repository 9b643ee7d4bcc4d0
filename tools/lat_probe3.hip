// Does a lone wave hide the second half of an f64 FMA (8 cycles) behind other instructions?  Straight-line code, one wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#define TIC(var) do { __builtin_amdgcn_sched_barrier(0); var = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define USE(v) asm volatile("" :: "v"(v))
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
__global__ void __launch_bounds__(64) probe(long long* out, double* sink, const double* in) {
    const int lane = threadIdx.x;
    double b = in[lane + 1], c = in[lane + 2];
    double f0 = in[lane], f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, m0 = f0 + 4, m1 = f0 + 5, m2 = f0 + 6, m3 = f0 + 7;
    int i0 = lane, i1 = lane + 1, i2 = lane + 2, i3 = lane + 3;
    long long t0, t1;
    // 0: 64 FMA (4 chains)
    TIC(t0);
    REP16(f0 = __builtin_fma(f0, b, c); f1 = __builtin_fma(f1, b, c); f2 = __builtin_fma(f2, b, c); f3 = __builtin_fma(f3, b, c);)
    USE(f0); USE(f1); USE(f2); USE(f3); TIC(t1);
    if (lane == 0) out[0] = t1 - t0;
    // 1: 64 MUL (4 chains)
    TIC(t0);
    REP16(m0 = m0 * b; m1 = m1 * b; m2 = m2 * b; m3 = m3 * b;)
    USE(m0); USE(m1); USE(m2); USE(m3); TIC(t1);
    if (lane == 0) out[1] = t1 - t0;
    // 2: 64 FMA + 64 MUL alternating
    TIC(t0);
    REP16(f0 = __builtin_fma(f0, b, c); m0 = m0 * b; f1 = __builtin_fma(f1, b, c); m1 = m1 * b; f2 = __builtin_fma(f2, b, c); m2 = m2 * b;
          f3 = __builtin_fma(f3, b, c); m3 = m3 * b;)
    USE(f0); USE(f1); USE(f2); USE(f3); USE(m0); USE(m1); USE(m2); USE(m3); TIC(t1);
    if (lane == 0) out[2] = t1 - t0;
    // 3: 64 FMA + 64 int alternating
    TIC(t0);
    REP16(f0 = __builtin_fma(f0, b, c); i0 = i0 * 3 + lane; f1 = __builtin_fma(f1, b, c); i1 = i1 * 3 + lane; f2 = __builtin_fma(f2, b, c);
          i2 = i2 * 3 + lane; f3 = __builtin_fma(f3, b, c); i3 = i3 * 3 + lane;)
    USE(f0); USE(f1); USE(f2); USE(f3); USE(i0); USE(i1); USE(i2); USE(i3); TIC(t1);
    if (lane == 0) out[3] = t1 - t0;
    // 4: 64 int (4 chains)
    TIC(t0);
    REP16(i0 = i0 * 3 + lane; i1 = i1 * 3 + lane; i2 = i2 * 3 + lane; i3 = i3 * 3 + lane;)
    USE(i0); USE(i1); USE(i2); USE(i3); TIC(t1);
    if (lane == 0) out[4] = t1 - t0;
    // 5: 64 FMA + 128 MUL (1 : 2)
    TIC(t0);
    REP16(f0 = __builtin_fma(f0, b, c); m0 = m0 * b; m1 = m1 * b; f1 = __builtin_fma(f1, b, c); m2 = m2 * b; m3 = m3 * b;
          f2 = __builtin_fma(f2, b, c); m0 = m0 * c; m1 = m1 * c; f3 = __builtin_fma(f3, b, c); m2 = m2 * c; m3 = m3 * c;)
    USE(f0); USE(f1); USE(f2); USE(f3); USE(m0); USE(m1); USE(m2); USE(m3); TIC(t1);
    if (lane == 0) out[5] = t1 - t0;
    sink[lane] = f0 + f1 + f2 + f3 + m0 + m1 + m2 + m3 + i0 + i1 + i2 + i3;
}
int main() {
    long long* d; double *s, *in;
    (void)hipMalloc(&d, 64); (void)hipMalloc(&s, 64 * 8); (void)hipMalloc(&in, 128 * 8);
    double h_in[128];
    for (int i = 0; i < 128; ++i) h_in[i] = 1.0 + 1e-6 * i;
    (void)hipMemcpy(in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
    const char* names[6] = {"64 FMA", "64 MUL", "64 FMA + 64 MUL alternating", "64 FMA + 64 i32 mad alternating", "64 i32 mad", "64 FMA + 128 MUL (1:2)"};
    probe<<<1, 64>>>(d, s, in); (void)hipDeviceSynchronize();
    probe<<<1, 64>>>(d, s, in); (void)hipDeviceSynchronize();
    long long h[8];
    (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    for (int i = 0; i < 6; ++i) printf("  %-36s %6lld cycles\n", names[i], h[i]);
    return 0;
}
