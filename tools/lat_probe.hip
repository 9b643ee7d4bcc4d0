// Latency probe for gfx950 (one wave): the dependent-operation costs the small-window factorisation is made of.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lat_probe tools/lat_probe.hip && /tmp/lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(64) probe(long long* out, double* sink, int n, int active) {
    __shared__ int chain[1024];
    __shared__ double vals[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) { chain[i] = (i * 37 + 11) & 1023; vals[i] = 1.0 + 1e-9 * i; }
    __syncthreads();
    if (lane >= active) return;
    long long t0, t1;
    // (0) LDS pointer chase, ds_read_b32
    int p = lane;
    t0 = clock64();
    for (int i = 0; i < n; ++i) p = chain[p];
    t1 = clock64();
    if (lane == 0) out[0] = t1 - t0;
    // (1) dependent f64 FMA chain
    double x = vals[p], a = 1.0000001, b = 1e-9;
    t0 = clock64();
    for (int i = 0; i < n; ++i) x = __builtin_fma(x, a, b);
    t1 = clock64();
    if (lane == 0) out[1] = t1 - t0;
    // (2) 8 independent FMA chains (issue rate)
    double y0 = x, y1 = x + 1, y2 = x + 2, y3 = x + 3, y4 = x + 4, y5 = x + 5, y6 = x + 6, y7 = x + 7;
    t0 = clock64();
    for (int i = 0; i < n; ++i) {
        y0 = __builtin_fma(y0, a, b); y1 = __builtin_fma(y1, a, b); y2 = __builtin_fma(y2, a, b); y3 = __builtin_fma(y3, a, b);
        y4 = __builtin_fma(y4, a, b); y5 = __builtin_fma(y5, a, b); y6 = __builtin_fma(y6, a, b); y7 = __builtin_fma(y7, a, b);
    }
    t1 = clock64();
    if (lane == 0) out[2] = t1 - t0;
    x = y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7;
    // (3) dependent rsq chain
    double z = fabs(x) + 2.0;
    t0 = clock64();
    for (int i = 0; i < n; ++i) z = __builtin_amdgcn_rsq(z) + 1.5;
    t1 = clock64();
    if (lane == 0) out[3] = t1 - t0;
    // (4) chase + dependent f64 load (index -> double -> index)
    double acc = z;
    t0 = clock64();
    for (int i = 0; i < n; ++i) { acc += vals[p]; p = chain[(p + (int)acc) & 1023]; }
    t1 = clock64();
    if (lane == 0) out[4] = t1 - t0;
    // (5) barrier
    t0 = clock64();
    for (int i = 0; i < n; ++i) __syncthreads();
    t1 = clock64();
    if (lane == 0) out[5] = t1 - t0;
    // (6) store -> barrier -> load round trip
    t0 = clock64();
    for (int i = 0; i < n; ++i) { vals[(lane + 1) & 63] = acc; __syncthreads(); acc += vals[lane]; __syncthreads(); }
    t1 = clock64();
    if (lane == 0) out[6] = t1 - t0;
    // (7) dependent f64 mul+fma pair (2 ops)
    t0 = clock64();
    for (int i = 0; i < n; ++i) { const double m = acc * a; acc = __builtin_fma(m, b, acc); }
    t1 = clock64();
    if (lane == 0) out[7] = t1 - t0;
    sink[lane] = acc + p;
}

int main() {
    long long* d; double* s;
    hipMalloc(&d, 64); hipMalloc(&s, 64 * 8);
    const int n = 2000;
    const char* names[8] = {"LDS pointer chase (ds_read_b32)", "dependent f64 FMA", "8 independent f64 FMAs (per FMA x8)", "dependent rsq_f64 + add",
                            "LDS double + index chase (2 trips)", "s_barrier (one wave)", "LDS store, barrier, load, barrier", "dependent mul + fma"};
    {   // calibration: clock64 ticks against wall time (HIP events around a long launch)
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int big = 2000000;
        probe<<<1, 64>>>(d, s, 1000, 64); hipDeviceSynchronize();
        hipEventRecord(e0); probe<<<1, 64>>>(d, s, big, 64); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        long long h[8]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
        long long tot = 0; for (int i = 0; i < 8; ++i) tot += h[i];
        printf("calibration: %lld clock64 ticks in %.3f ms -> %.1f MHz\n", tot, ms, tot / (ms * 1e3));
    }
    for (int active : {64, 2}) {
        probe<<<1, 64>>>(d, s, n, active);
        hipDeviceSynchronize();
        probe<<<1, 64>>>(d, s, n, active);
        hipDeviceSynchronize();
        long long h[8];
        hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
        printf("active lanes %d: cycles per iteration\n", active);
        for (int i = 0; i < 8; ++i) printf("  %-40s %8.1f\n", names[i], (double)h[i] / n);
    }
    return 0;
}
