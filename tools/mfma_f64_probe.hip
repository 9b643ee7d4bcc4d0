// Operand / result layout of v_mfma_f64_16x16x4_f64 on gfx950, found by experiment (D = A B + C, A 16x4, B 4x16).
//   hipcc --offload-arch=gfx950 -O2 -o tools/mfma_f64_probe.bin tools/mfma_f64_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double v4f64 __attribute__((ext_vector_type(4)));

__global__ void probe(const double* A, const double* B, double* D) {   // A[i][k] = A[4 i + k], B[k][j] = B[16 k + j]
    const int l = threadIdx.x;
    const double a = A[4 * (l % 16) + l / 16];        // candidate: lane supplies A[i = l % 16][k = l / 16]
    const double b = B[16 * (l / 16) + l % 16];       // candidate: lane supplies B[k = l / 16][j = l % 16]
    v4f64 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[4 * l + v] = c[v];
}

int main() {
    double hA[64], hB[64], hD[256], ref[16][16];
    for (int i = 0; i < 64; ++i) { hA[i] = 1.0 + 0.37 * i + 0.01 * i * i; hB[i] = 2.0 - 0.11 * i + 0.003 * i * i; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += hA[4 * i + k] * hB[16 * k + j]; ref[i][j] = s; }
    double *dA, *dB, *dD;
    (void)hipMalloc(&dA, sizeof(hA)); (void)hipMalloc(&dB, sizeof(hB)); (void)hipMalloc(&dD, sizeof(hD));
    (void)hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD);
    (void)hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    // candidates for D: lane l, element v
    const char* names[4] = {"D[4 (l/16) + v][l % 16]", "D[l % 16][4 (l/16) + v]", "D[(l/16) + 4 v][l % 16]", "D[l % 16][(l/16) + 4 v]"};
    for (int cand = 0; cand < 4; ++cand) {
        double err = 0;
        for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
            int i, j;
            if (cand == 0) { i = 4 * (l / 16) + v; j = l % 16; }
            else if (cand == 1) { i = l % 16; j = 4 * (l / 16) + v; }
            else if (cand == 2) { i = (l / 16) + 4 * v; j = l % 16; }
            else { i = l % 16; j = (l / 16) + 4 * v; }
            err = fmax(err, fabs(hD[4 * l + v] - ref[i][j]));
        }
        printf("%-28s max |diff| = %.3e\n", names[cand], err);
    }
    return 0;
}
