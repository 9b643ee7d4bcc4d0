// C ABI of the batched fusion snapshot solver (BASELINE config 3). Host side only.
#include "../../include/localization_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <new>
#include <vector>

#include "fusion_kernel.h"
#include "snapshot_kernel.h"  // launch_pack_kmb

extern int locamd_fail(int code, const char* what);
extern int locamd_fail_hip(hipError_t e, const char* where);
#define LOC_HIP(expr)                                              \
    do {                                                           \
        hipError_t _e = (expr);                                    \
        if (_e != hipSuccess) return locamd_fail_hip(_e, #expr);   \
    } while (0)

struct loc_fusion {
    int device = 0;
    long long B = 0;
    int M = 0;
    loc_fusion_params prm{};
    double *d_anchors = nullptr, *d_offset = nullptr, *d_pose = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    std::vector<hipEvent_t> ev;   // loc_fusion_timing_*: one pair per launch
    int ev_used = 0;
    bool timing = false;
    long long epochs_done = 0;
    // staging for the host path
    float *d_dist = nullptr, *d_err = nullptr;
    double *d_imu = nullptr, *d_out_pose = nullptr, *d_out_chi2 = nullptr;
    uint8_t* d_out_trials = nullptr;
    int staged = 0;
    // pipelined host path (loc_fusion_solve_host_kmb)
    float *d_raw_dist = nullptr, *d_raw_err = nullptr;
    int raw_epochs = 0;
    hipStream_t in_stream = nullptr, out_stream = nullptr;
    std::vector<hipEvent_t> pipe_ev;
};

static int fusion_ensure_staging(loc_fusion* f, int32_t epochs);

static void fusion_free_staging(loc_fusion* f) {
    void* p[] = {f->d_dist, f->d_err, f->d_imu, f->d_out_pose, f->d_out_chi2, f->d_out_trials};
    for (void* x : p) if (x) (void)hipFree(x);
    f->d_dist = f->d_err = nullptr; f->d_imu = f->d_out_pose = f->d_out_chi2 = nullptr; f->d_out_trials = nullptr; f->staged = 0;
}

extern "C" {

void loc_fusion_default_params(loc_fusion_params* p) {
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->maximum_iteration = 20; p->distance_outlier = 1.0; p->gate_warmup_epochs = 1;
    p->jacobian = LOC_JAC_NUMERIC_G2O;   // the reference's configuration (types_edge_se3range.h:45-74: no linearizeOplus)
}

int loc_fusion_destroy(loc_fusion* f) {
    if (!f) return LOC_OK;
    (void)hipSetDevice(f->device);
    fusion_free_staging(f);
    for (hipEvent_t ev : f->pipe_ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : f->ev) (void)hipEventDestroy(ev);
    if (f->d_raw_dist) (void)hipFree(f->d_raw_dist);
    if (f->d_raw_err) (void)hipFree(f->d_raw_err);
    if (f->in_stream) (void)hipStreamDestroy(f->in_stream);
    if (f->out_stream) (void)hipStreamDestroy(f->out_stream);
    if (f->d_anchors) (void)hipFree(f->d_anchors);
    if (f->d_offset) (void)hipFree(f->d_offset);
    if (f->d_pose) (void)hipFree(f->d_pose);
    if (f->ev0) (void)hipEventDestroy(f->ev0);
    if (f->ev1) (void)hipEventDestroy(f->ev1);
    if (f->stream) (void)hipStreamDestroy(f->stream);
    delete f;
    return LOC_OK;
}

int loc_fusion_create(loc_fusion** out, int32_t device, int64_t batch, int32_t n_anchors, const double* anchors,
                      const loc_fusion_params* params) {
    if (!out) return locamd_fail(LOC_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (batch <= 0 || n_anchors <= 0 || !anchors) return locamd_fail(LOC_ERR_INVALID, "batch/anchors");
    if (n_anchors > 8) return locamd_fail(LOC_ERR_UNSUPPORTED, "more than 8 anchors per tag in the fusion kernel");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return locamd_fail(LOC_ERR_NO_DEVICE, "no HIP device visible: localization_amd has no CPU fallback");
    if (device < 0 || device >= ndev) return locamd_fail(LOC_ERR_INVALID, "device index out of range");
    loc_fusion_params prm;
    if (params) prm = *params; else loc_fusion_default_params(&prm);
    if (prm.block_threads == 0) prm.block_threads = 256;
    if (prm.block_threads % 64 || prm.block_threads > 256 || prm.gate_warmup_epochs < 0 ||
        (prm.jacobian != LOC_JAC_ANALYTIC && prm.jacobian != LOC_JAC_NUMERIC_G2O)) return locamd_fail(LOC_ERR_INVALID, "fusion params");
    LOC_HIP(hipSetDevice(device));
    loc_fusion* f = new (std::nothrow) loc_fusion();
    if (!f) return locamd_fail(LOC_ERR_INVALID, "out of host memory");
    f->device = device; f->B = batch; f->M = n_anchors; f->prm = prm;
    double anch[24];
    std::memset(anch, 0, sizeof(anch));
    std::memcpy(anch, anchors, sizeof(double) * 3 * (size_t)n_anchors);
    std::vector<double> pose0((size_t)7 * batch, 0.0);
    for (long long b = 0; b < batch; ++b) pose0[(size_t)6 * batch + b] = 1.0;  // identity rotation
    hipError_t e;
    if ((e = hipMalloc((void**)&f->d_anchors, sizeof(anch))) != hipSuccess ||
        (e = hipMalloc((void**)&f->d_offset, 3 * sizeof(double))) != hipSuccess ||
        (e = hipMalloc((void**)&f->d_pose, sizeof(double) * 7 * (size_t)batch)) != hipSuccess ||
        (e = hipMemcpy(f->d_anchors, anch, sizeof(anch), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(f->d_offset, prm.antenna_offset, 3 * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(f->d_pose, pose0.data(), sizeof(double) * 7 * (size_t)batch, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&f->ev0)) != hipSuccess || (e = hipEventCreate(&f->ev1)) != hipSuccess) {
        loc_fusion_destroy(f);
        return locamd_fail_hip(e, "loc_fusion_create");
    }
    *out = f;
    return LOC_OK;
}

int loc_fusion_set_poses(loc_fusion* f, const double* pose) {
    if (!f || !pose) return locamd_fail(LOC_ERR_INVALID, "null");
    LOC_HIP(hipSetDevice(f->device));
    LOC_HIP(hipMemcpy(f->d_pose, pose, sizeof(double) * 7 * (size_t)f->B, hipMemcpyHostToDevice));
    f->epochs_done = 0;
    return LOC_OK;
}
int loc_fusion_get_poses(loc_fusion* f, double* pose) {
    if (!f || !pose) return locamd_fail(LOC_ERR_INVALID, "null");
    LOC_HIP(hipSetDevice(f->device));
    LOC_HIP(hipMemcpy(pose, f->d_pose, sizeof(double) * 7 * (size_t)f->B, hipMemcpyDeviceToHost));
    return LOC_OK;
}

int loc_fusion_solve_device(loc_fusion* f, int32_t epochs, const float* dist, const float* err, const double* imu,
                            double* out_pose, double* out_chi2, uint8_t* out_trials, void* hip_stream) {
    if (!f) return locamd_fail(LOC_ERR_INVALID, "null handle");
    if (epochs <= 0 || !dist || !err || !imu || !out_pose || !out_chi2) return locamd_fail(LOC_ERR_INVALID, "solve arguments");
    if (((uintptr_t)dist | (uintptr_t)err | (uintptr_t)imu) & 15u) return locamd_fail(LOC_ERR_INVALID, "inputs must be 16-byte aligned");
    LOC_HIP(hipSetDevice(f->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : f->stream;
    locamd::FusionArgs a;
    a.dist = dist; a.err = err; a.imu = imu; a.pose = f->d_pose; a.out_pose = out_pose; a.out_chi2 = out_chi2; a.out_trials = out_trials;
    a.anchors = f->d_anchors; a.offset = f->d_offset; a.B = f->B; a.K = epochs; a.iterations = f->prm.maximum_iteration;
    a.gate = f->prm.distance_outlier;
    a.jacobian = f->prm.jacobian;
    const long long left = (long long)f->prm.gate_warmup_epochs - f->epochs_done;
    a.gate_from_epoch = left > 0 ? (int)(left > epochs ? epochs : left) : 0;
    const bool pair = f->timing && (size_t)(f->ev_used + 2) <= f->ev.size();
    LOC_HIP(hipEventRecord(f->ev0, st));
    if (pair) LOC_HIP(hipEventRecord(f->ev[f->ev_used], st));
    hipError_t e = locamd::launch_fusion(a, f->prm.block_threads, st);
    if (e != hipSuccess) return locamd_fail_hip(e, "launch_fusion");
    if (pair) { LOC_HIP(hipEventRecord(f->ev[f->ev_used + 1], st)); f->ev_used += 2; }
    LOC_HIP(hipEventRecord(f->ev1, st));
    f->timed = true;
    f->epochs_done += epochs;
    return LOC_OK;
}

int loc_fusion_solve_host(loc_fusion* f, int32_t epochs, const float* dist_h, const float* err_h, const double* imu_h,
                          double* out_pose_h, double* out_chi2_h, uint8_t* out_trials_h) {
    if (!f) return locamd_fail(LOC_ERR_INVALID, "null handle");
    if (epochs <= 0 || !dist_h || !err_h || !imu_h || !out_pose_h || !out_chi2_h) return locamd_fail(LOC_ERR_INVALID, "solve arguments");
    LOC_HIP(hipSetDevice(f->device));
    const size_t B = (size_t)f->B, K = (size_t)epochs;
    const size_t nf = K * 2 * B * 4;
    if (int rc = fusion_ensure_staging(f, epochs)) return rc;
    LOC_HIP(hipMemcpyAsync(f->d_dist, dist_h, nf * sizeof(float), hipMemcpyHostToDevice, f->stream));
    LOC_HIP(hipMemcpyAsync(f->d_err, err_h, nf * sizeof(float), hipMemcpyHostToDevice, f->stream));
    LOC_HIP(hipMemcpyAsync(f->d_imu, imu_h, K * B * 8 * sizeof(double), hipMemcpyHostToDevice, f->stream));
    int rc = loc_fusion_solve_device(f, epochs, f->d_dist, f->d_err, f->d_imu, f->d_out_pose, f->d_out_chi2, f->d_out_trials, f->stream);
    if (rc != LOC_OK) return rc;
    LOC_HIP(hipMemcpyAsync(out_pose_h, f->d_out_pose, K * 7 * B * sizeof(double), hipMemcpyDeviceToHost, f->stream));
    LOC_HIP(hipMemcpyAsync(out_chi2_h, f->d_out_chi2, K * B * sizeof(double), hipMemcpyDeviceToHost, f->stream));
    if (out_trials_h) LOC_HIP(hipMemcpyAsync(out_trials_h, f->d_out_trials, K * B, hipMemcpyDeviceToHost, f->stream));
    LOC_HIP(hipStreamSynchronize(f->stream));
    return LOC_OK;
}

int loc_fusion_solve_host_kmb(loc_fusion* f, int32_t epochs, const float* dist_kmb, const float* err_kmb, const double* imu_h,
                              double* out_pose_h, double* out_chi2_h, uint8_t* out_trials_h) {
    if (!f) return locamd_fail(LOC_ERR_INVALID, "null handle");
    if (epochs <= 0 || !dist_kmb || !err_kmb || !imu_h || !out_pose_h || !out_chi2_h) return locamd_fail(LOC_ERR_INVALID, "solve arguments");
    LOC_HIP(hipSetDevice(f->device));
    const size_t B = (size_t)f->B, M = (size_t)f->M;
    if (int rc = fusion_ensure_staging(f, epochs)) return rc;
    if (epochs > f->raw_epochs) {
        if (f->d_raw_dist) (void)hipFree(f->d_raw_dist);
        if (f->d_raw_err) (void)hipFree(f->d_raw_err);
        f->d_raw_dist = f->d_raw_err = nullptr; f->raw_epochs = 0;
        LOC_HIP(hipMalloc((void**)&f->d_raw_dist, sizeof(float) * M * B * (size_t)epochs));
        LOC_HIP(hipMalloc((void**)&f->d_raw_err, sizeof(float) * M * B * (size_t)epochs));
        f->raw_epochs = epochs;
    }
    if (!f->in_stream) LOC_HIP(hipStreamCreateWithFlags(&f->in_stream, hipStreamNonBlocking));
    if (!f->out_stream) LOC_HIP(hipStreamCreateWithFlags(&f->out_stream, hipStreamNonBlocking));
    // same chunked three-stream pipeline as loc_snapshot_solve_host_kmb (capi.cpp); pageable buffers go as one chunk
    auto pinned = [](const void* p) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
        return at.type == hipMemoryTypeHost;
    };
    const bool overlap = pinned(dist_kmb) && pinned(err_kmb) && pinned(imu_h) && pinned(out_pose_h) && pinned(out_chi2_h);
    const int ce = overlap ? (int)std::max<size_t>(1, (8u << 20) / (B * 8 * sizeof(double))) : epochs;
    const int nchunks = (epochs + ce - 1) / ce;
    while ((int)f->pipe_ev.size() < 2 * nchunks) {
        hipEvent_t ev;
        LOC_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        f->pipe_ev.push_back(ev);
    }
    for (int c = 0; c < nchunks; ++c) {
        const int k0 = c * ce, kc = std::min(ce, epochs - k0);
        const size_t roff = (size_t)k0 * M * B, rn = (size_t)kc * M * B, toff = (size_t)k0 * 2 * B * 4, ioff = (size_t)k0 * B * 8;
        LOC_HIP(hipMemcpyAsync(f->d_raw_dist + roff, dist_kmb + roff, rn * sizeof(float), hipMemcpyHostToDevice, f->in_stream));
        LOC_HIP(hipMemcpyAsync(f->d_raw_err + roff, err_kmb + roff, rn * sizeof(float), hipMemcpyHostToDevice, f->in_stream));
        LOC_HIP(hipMemcpyAsync(f->d_imu + ioff, imu_h + ioff, (size_t)kc * B * 8 * sizeof(double), hipMemcpyHostToDevice, f->in_stream));
        LOC_HIP(hipEventRecord(f->pipe_ev[2 * c], f->in_stream));
        LOC_HIP(hipStreamWaitEvent(f->stream, f->pipe_ev[2 * c], 0));
        hipError_t e = locamd::launch_pack_kmb(f->d_raw_dist + roff, f->d_dist + toff, f->B, f->M, 2, kc, 0.f, f->stream);
        if (e == hipSuccess) e = locamd::launch_pack_kmb(f->d_raw_err + roff, f->d_err + toff, f->B, f->M, 2, kc, 0.f, f->stream);
        if (e != hipSuccess) return locamd_fail_hip(e, "launch_pack_kmb");
        int rc = loc_fusion_solve_device(f, kc, f->d_dist + toff, f->d_err + toff, f->d_imu + ioff, f->d_out_pose + (size_t)k0 * 7 * B,
                                         f->d_out_chi2 + (size_t)k0 * B, f->d_out_trials + (size_t)k0 * B, f->stream);
        if (rc != LOC_OK) return rc;
        LOC_HIP(hipEventRecord(f->pipe_ev[2 * c + 1], f->stream));
        LOC_HIP(hipStreamWaitEvent(f->out_stream, f->pipe_ev[2 * c + 1], 0));
        LOC_HIP(hipMemcpyAsync(out_pose_h + (size_t)k0 * 7 * B, f->d_out_pose + (size_t)k0 * 7 * B, sizeof(double) * 7 * B * (size_t)kc, hipMemcpyDeviceToHost, f->out_stream));
        LOC_HIP(hipMemcpyAsync(out_chi2_h + (size_t)k0 * B, f->d_out_chi2 + (size_t)k0 * B, sizeof(double) * B * (size_t)kc, hipMemcpyDeviceToHost, f->out_stream));
        if (out_trials_h) LOC_HIP(hipMemcpyAsync(out_trials_h + (size_t)k0 * B, f->d_out_trials + (size_t)k0 * B, B * (size_t)kc, hipMemcpyDeviceToHost, f->out_stream));
    }
    LOC_HIP(hipStreamSynchronize(f->out_stream));
    LOC_HIP(hipStreamSynchronize(f->stream));
    return LOC_OK;
}

int loc_fusion_last_kernel_ms(loc_fusion* f, double* ms) {
    if (!f || !ms) return locamd_fail(LOC_ERR_INVALID, "null");
    if (!f->timed) { *ms = 0; return LOC_OK; }
    LOC_HIP(hipSetDevice(f->device));
    LOC_HIP(hipEventSynchronize(f->ev1));
    float t = 0;
    LOC_HIP(hipEventElapsedTime(&t, f->ev0, f->ev1));
    *ms = t;
    return LOC_OK;
}

int loc_fusion_timing_begin(loc_fusion* f, int32_t max_launches) {
    if (!f || max_launches <= 0) return locamd_fail(LOC_ERR_INVALID, "timing_begin");
    LOC_HIP(hipSetDevice(f->device));
    while ((int)f->ev.size() < 2 * max_launches) {
        hipEvent_t ev;
        LOC_HIP(hipEventCreate(&ev));
        f->ev.push_back(ev);
    }
    f->ev_used = 0;
    f->timing = true;
    return LOC_OK;
}
int loc_fusion_timing_end(loc_fusion* f, int32_t* n_launches, double* total_ms, double* avg_ms) {
    if (!f) return locamd_fail(LOC_ERR_INVALID, "timing_end");
    LOC_HIP(hipSetDevice(f->device));
    f->timing = false;
    double tot = 0;
    const int n = f->ev_used / 2;
    for (int i = 0; i < n; ++i) {
        LOC_HIP(hipEventSynchronize(f->ev[2 * i + 1]));
        float ms = 0;
        LOC_HIP(hipEventElapsedTime(&ms, f->ev[2 * i], f->ev[2 * i + 1]));
        tot += ms;
    }
    if (n_launches) *n_launches = n;
    if (total_ms) *total_ms = tot;
    if (avg_ms) *avg_ms = n ? tot / n : 0.0;
    f->ev_used = 0;
    return LOC_OK;
}

}  // extern "C"

static int fusion_ensure_staging(loc_fusion* f, int32_t epochs) {
    if (epochs <= f->staged) return LOC_OK;
    const size_t B = (size_t)f->B, K = (size_t)epochs, nf = K * 2 * B * 4;
    fusion_free_staging(f);
    LOC_HIP(hipMalloc((void**)&f->d_dist, nf * sizeof(float)));
    LOC_HIP(hipMalloc((void**)&f->d_err, nf * sizeof(float)));
    LOC_HIP(hipMalloc((void**)&f->d_imu, K * B * 8 * sizeof(double)));
    LOC_HIP(hipMalloc((void**)&f->d_out_pose, K * 7 * B * sizeof(double)));
    LOC_HIP(hipMalloc((void**)&f->d_out_chi2, K * B * sizeof(double)));
    LOC_HIP(hipMalloc((void**)&f->d_out_trials, K * B));
    f->staged = epochs;
    return LOC_OK;
}
