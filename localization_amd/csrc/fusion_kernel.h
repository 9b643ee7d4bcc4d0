// Internal interface of the 6-DoF fusion snapshot kernel (fusion_kernel.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace locamd {

struct FusionArgs {
    const float* dist;     // [K][2][B][4]
    const float* err;      // [K][2][B][4]
    const double* imu;     // [K][B][8]   q xyzw, orientation covariance c0 c4 c8, pad
    double* pose;          // [7][B]      state: t xyz, q xyzw (in/out)
    double* out_pose;      // [K][7][B]
    double* out_chi2;      // [K][B]
    uint8_t* out_trials;   // [K][B] or nullptr
    const double* anchors; // [8][3] device (padded rows 0)
    const double* offset;  // [3] device: antenna lever arm on the tag
    long long B;
    int K;
    int iterations;
    double gate;
    int gate_from_epoch;
    int jacobian;          // 0: analytic range Jacobians, 1: g2o's central differences (delta = 1e-9)
};

hipError_t launch_fusion(const FusionArgs& a, int block_threads, hipStream_t stream);

}  // namespace locamd
