// FETCH_SIZE / WRITE_SIZE calibration for the access patterns of the window kernels: tiny kernels that move a
// KNOWN number of bytes, run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes); the ratio counted / known
// per pattern is what tools/make_window_traffic.py applies instead of the guide's x2 (which MI355X_MICROARCH.md gives for 16 B per
// lane coalesced streams only).   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/fetch_calib.bin
//   read16         16 B per lane, unit stride (float4): the snapshot / fusion kernels' range tiles
//   read8          8 B per lane, unit stride: one 512-byte line per wave load — the [entry][lane] workspaces of chain_lm_kernel / chain3
//   read8_s128     8 B per lane, lanes 128 B apart: every lane its own cache line (window_lm_kernel's per-instance workspaces: a lane
//                  per block row / per pose)
//   read8_s4096    8 B per lane, lanes 4 KB apart (different DRAM pages)
//   write8 / write8_s128   the same for stores
// Every kernel touches each byte of a 1 GiB region once (4x the 256 MiB Infinity Cache), one region per kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void calib_read16(const float4* p, size_t n, float* out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}
__global__ void calib_read8(const double* p, size_t n, double* out) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 123.456) out[0] = acc;
}
// n_lines cache lines of `stride` bytes: lane l of a wave reads the first double of line (base + l)
__global__ void calib_read8_strided(const double* p, size_t n_lines, size_t stride_doubles, double* out) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_lines; i += (size_t)gridDim.x * blockDim.x) acc += p[i * stride_doubles];
    if (acc == 123.456) out[0] = acc;
}
__global__ void calib_write8(double* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (double)i;
}
__global__ void calib_write8_strided(double* p, size_t n_lines, size_t stride_doubles) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_lines; i += (size_t)gridDim.x * blockDim.x) p[i * stride_doubles] = (double)i;
}

int main() {
    const size_t GiB = 1ull << 30;
    char* buf;
    double* out;
    CHECK(hipMalloc((void**)&buf, 6 * GiB));
    CHECK(hipMalloc((void**)&out, 64));
    CHECK(hipMemset(buf, 0, 6 * GiB));
    CHECK(hipDeviceSynchronize());
    const dim3 grid(256 * 8), block(256);
    hipLaunchKernelGGL(calib_read16, grid, block, 0, 0, (const float4*)(buf + 0 * GiB), GiB / 16, (float*)out);
    hipLaunchKernelGGL(calib_read8, grid, block, 0, 0, (const double*)(buf + 1 * GiB), GiB / 8, out);
    hipLaunchKernelGGL(calib_read8_strided, grid, block, 0, 0, (const double*)(buf + 2 * GiB), GiB / 128, (size_t)16, out);
    hipLaunchKernelGGL(calib_read8_strided, grid, block, 0, 0, (const double*)(buf + 3 * GiB), GiB / 4096, (size_t)512, out);
    hipLaunchKernelGGL(calib_write8, grid, block, 0, 0, (double*)(buf + 4 * GiB), GiB / 8);
    hipLaunchKernelGGL(calib_write8_strided, grid, block, 0, 0, (double*)(buf + 5 * GiB), GiB / 128, (size_t)16);
    CHECK(hipDeviceSynchronize());
    printf("{\"read16_bytes\": %zu, \"read8_bytes\": %zu, \"read8_s128_requested_bytes\": %zu, \"read8_s128_lines\": %zu, \"read8_s4096_requested_bytes\": %zu, "
           "\"read8_s4096_lines\": %zu, \"write8_bytes\": %zu, \"write8_s128_requested_bytes\": %zu, \"write8_s128_lines\": %zu}\n",
           GiB, GiB, GiB / 128 * 8, GiB / 128, GiB / 4096 * 8, GiB / 4096, GiB, GiB / 128 * 8, GiB / 128);
    return 0;
}
