// Internal interface between the C ABI (capi.cpp) and the gfx950 snapshot kernel (snapshot_kernel.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace locamd {

struct SnapshotArgs {
    const float* dist;    // [K][M4][B][4]
    const float* err;     // [K][M4][B][4]
    double* pos;          // [3][B]   state (in/out)
    double* out_pos;      // [K][3][B]
    double* out_chi2;     // [K][B]
    uint8_t* out_trials;  // [K][B] or nullptr
    const double* anchors;  // [M_PAD][3] device, padded rows = 0
    long long B;
    int K;
    int M4;
    int iterations;
    double gate;
    int gate_from_epoch;  // epochs k < gate_from_epoch run un-gated (reference warm-up, localization.cpp:309)
};

// Returns hipSuccess or the launch error; hipErrorInvalidValue if (m_pad, lpi, jac) has no instantiation.
hipError_t launch_snapshot(const SnapshotArgs& a, int m_pad, int lpi, int jac, int block_threads, hipStream_t stream);
// true if a kernel exists for this combination
bool snapshot_supported(int m_pad, int lpi);
// [K][M][B] float (anchor-major SoA, the layout a host caller naturally holds) -> float4 tiles [K][M4][B][4], slots
// m >= M filled with `pad`.  HBM-bound transpose, one float4 store per thread.
hipError_t launch_pack_kmb(const float* src_kmb, float* dst_tiles, long long B, int M, int M4, int K, float pad, hipStream_t stream);

}  // namespace locamd
