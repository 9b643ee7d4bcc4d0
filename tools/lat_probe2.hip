// Issue / latency probe for gfx950, straight-line code (no loop overhead): cycles per operation for one resident wave.
//   hipcc --offload-arch=gfx950 -O3 -o tools/lat_probe2.bin tools/lat_probe2.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define TIC(var) do { __builtin_amdgcn_sched_barrier(0); var = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define USE(v) asm volatile("" :: "v"(v))
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

__global__ void __launch_bounds__(1024) probe(long long* out, double* sink, const double* in, int active) {
    __shared__ double vals[2048];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) vals[i] = in[i & 63] + i;
    __syncthreads();
    if (lane >= active) return;
    double a = in[lane], b = in[lane + 1], c = in[lane + 2];
    long long t0, t1;
    // 0: 64 dependent FMAs
    double x = a;
    TIC(t0);
    REP64(x = __builtin_fma(x, b, c);)
    USE(x); TIC(t1);
    if (threadIdx.x == 0) out[0] = t1 - t0;
    // 1: 64 FMAs in 8 independent chains
    double y0 = x, y1 = a + 1, y2 = a + 2, y3 = a + 3, y4 = a + 4, y5 = a + 5, y6 = a + 6, y7 = a + 7;
    TIC(t0);
    REP4(REP4(y0 = __builtin_fma(y0, b, c); y1 = __builtin_fma(y1, b, c); y2 = __builtin_fma(y2, b, c); y3 = __builtin_fma(y3, b, c);)
         REP4(y4 = __builtin_fma(y4, b, c); y5 = __builtin_fma(y5, b, c); y6 = __builtin_fma(y6, b, c); y7 = __builtin_fma(y7, b, c);))
    USE(y0); USE(y1); USE(y2); USE(y3); USE(y4); USE(y5); USE(y6); USE(y7); TIC(t1);
    if (threadIdx.x == 0) out[1] = t1 - t0;
    x = ((y0 + y1) + (y2 + y3)) + ((y4 + y5) + (y6 + y7));
    // 2: 64 dependent rsq
    double z = fabs(x) + 2.0;
    TIC(t0);
    REP64(z = __builtin_amdgcn_rsq(z);)
    USE(z); TIC(t1);
    if (threadIdx.x == 0) out[2] = t1 - t0;
    // 3: 64 dependent f64 multiplies
    double w = z + 1.0;
    TIC(t0);
    REP64(w = w * b;)
    USE(w); TIC(t1);
    if (threadIdx.x == 0) out[3] = t1 - t0;
    // 4: 64 dependent 32-bit integer adds (VALU)
    int k = lane + (int)w;
    TIC(t0);
    REP64(k = k * 3 + lane;)
    USE(k); TIC(t1);
    if (threadIdx.x == 0) out[4] = t1 - t0;
    // 5: 64 independent LDS reads (b64), one wait
    double acc = 0;
    {
        const double* p = vals + (k & 7);
        TIC(t0);
        double v[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) v[i] = p[i * 9];
#pragma unroll
        for (int i = 0; i < 64; ++i) acc += v[i];
        USE(acc); TIC(t1);
        if (threadIdx.x == 0) out[5] = t1 - t0;
    }
    // 6: 16 dependent LDS reads (index chain through doubles)
    {
        int q = k & 1023;
        TIC(t0);
        REP16(q = ((int)vals[q]) & 1023;)
        USE(q); TIC(t1);
        if (threadIdx.x == 0) out[6] = t1 - t0;
        acc += q;
    }
    // 7: 64 independent f64 adds into one serial chain after independent muls (mul independent, add dependent)
    {
        double s2 = acc;
        TIC(t0);
        REP64(s2 = s2 + b;)
        USE(s2); TIC(t1);
        if (threadIdx.x == 0) out[7] = t1 - t0;
        acc = s2;
    }
    sink[lane] = acc + x + w + z;
}

int main() {
    long long* d; double *s, *in;
    (void)hipMalloc(&d, 64); (void)hipMalloc(&s, 64 * 8); (void)hipMalloc(&in, 128 * 8);
    double h_in[128];
    for (int i = 0; i < 128; ++i) h_in[i] = 1.0 + 1e-3 * i;
    (void)hipMemcpy(in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
    const char* names[8] = {"64 dependent f64 FMA", "64 f64 FMA, 8 chains", "64 dependent rsq_f64", "64 dependent f64 mul", "64 dependent i32 mad",
                            "64 independent ds_read_b64 + 64 dependent adds", "16 dependent LDS reads (+cvt)", "64 dependent f64 add"};
    const int per[8] = {64, 64, 64, 64, 64, 64, 16, 64};
    for (int threads : {64, 256, 512, 1024}) {
        const int active = 64;
        probe<<<1, threads>>>(d, s, in, active); (void)hipDeviceSynchronize();
        probe<<<1, threads>>>(d, s, in, active); (void)hipDeviceSynchronize();
        long long h[8];
        (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
        printf("waves in the workgroup %d (4 SIMDs per CU): wave 0's view\n", threads / 64);
        for (int i = 0; i < 8; ++i) printf("  %-50s %6lld cycles  %6.1f per op\n", names[i], h[i], (double)h[i] / per[i]);
    }
    return 0;
}
