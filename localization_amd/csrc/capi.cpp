// C ABI of localization_amd (see include/localization_amd.h).  Host side only; device code lives in *.hip.
#include "../../include/localization_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "snapshot_kernel.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* what) {
    g_last_error = what ? what : "";
    return code;
}
int fail_hip(hipError_t e, const char* where) {
    g_last_error = std::string(where) + ": " + hipGetErrorString(e);
    return LOC_ERR_HIP;
}
#define LOC_HIP(expr)                                         \
    do {                                                      \
        hipError_t _e = (expr);                               \
        if (_e != hipSuccess) return fail_hip(_e, #expr);     \
    } while (0)

}  // namespace

// shared with the other capi_*.cpp files
int locamd_fail(int code, const char* what) { return fail(code, what); }
int locamd_fail_hip(hipError_t e, const char* where) { return fail_hip(e, where); }

struct loc_snapshot {
    int device = 0;
    long long B = 0;
    int M = 0, M4 = 0, M_PAD = 0;
    int lpi = 1;
    loc_snapshot_params prm{};
    double* d_anchors = nullptr;  // [M_PAD][3]
    double* d_pos = nullptr;      // [3][B]
    hipStream_t own_stream = nullptr;
    // staging buffers for the host convenience path
    float *d_dist = nullptr, *d_err = nullptr;
    double *d_out_pos = nullptr, *d_out_chi2 = nullptr;
    uint8_t* d_out_trials = nullptr;
    int staged_epochs = 0;
    long long epochs_done = 0;
    // pipelined host path (loc_snapshot_solve_host_kmb): raw [K][M][B] staging, copy streams, per-chunk events
    float *d_raw_dist = nullptr, *d_raw_err = nullptr;
    int raw_epochs = 0;
    hipStream_t in_stream = nullptr, out_stream = nullptr;
    std::vector<hipEvent_t> pipe_ev;
    // timing
    std::vector<hipEvent_t> ev;
    int ev_used = 0;
    bool timing = false;
};

extern "C" {

const char* loc_last_error(void) { return g_last_error.c_str(); }
int32_t loc_abi_version(void) { return LOC_ABI_VERSION; }

int32_t loc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- multi-GPU shard descriptor (host only): contiguous ceil-divided slices, SURVEY.md §8(e) --------------------------------
int loc_shard_bounds(int64_t total, int32_t rank, int32_t world, int64_t* lo, int64_t* hi) {
    if (total < 0 || world <= 0 || rank < 0 || rank >= world || !lo || !hi) return fail(LOC_ERR_INVALID, "loc_shard_bounds arguments");
    const int64_t per = (total + world - 1) / world;
    const int64_t a = rank * per < total ? rank * per : total;
    *lo = a;
    *hi = a + per < total ? a + per : total;
    return LOC_OK;
}

int loc_shard_plan(int64_t total, int32_t world, int32_t devices_per_node, loc_shard* out) {
    if (total < 0 || world <= 0 || !out) return fail(LOC_ERR_INVALID, "loc_shard_plan arguments");
    if (devices_per_node <= 0) {
        devices_per_node = loc_device_count();
        if (devices_per_node <= 0) return fail(LOC_ERR_NO_DEVICE, "no HIP device visible: pass devices_per_node explicitly");
    }
    for (int32_t r = 0; r < world; ++r) {
        out[r].rank = r; out[r].world = world; out[r].device = r % devices_per_node; out[r].reserved = 0;
        const int rc = loc_shard_bounds(total, r, world, &out[r].lo, &out[r].hi);
        if (rc != LOC_OK) return rc;
    }
    return LOC_OK;
}

void loc_snapshot_default_params(loc_snapshot_params* p) {
    if (!p) return;
    p->maximum_iteration = 20;   // reference default, localization.cpp:65
    p->distance_outlier = 1.0;   // reference default, localization.cpp:78
    p->gate_warmup_epochs = 1;
    p->jacobian = LOC_JAC_NUMERIC_G2O;   // the reference's configuration (types_edge_se3range.h:45-74: no linearizeOplus)
    p->lanes_per_instance = 0;
    p->block_threads = 0;
}

int loc_snapshot_create(loc_snapshot** out, int32_t device, int64_t batch, int32_t n_anchors,
                        const double* anchors_xyz_host, const loc_snapshot_params* params) {
    if (!out) return fail(LOC_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (batch <= 0 || n_anchors <= 0 || !anchors_xyz_host) return fail(LOC_ERR_INVALID, "batch/anchors");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(LOC_ERR_NO_DEVICE, "no HIP device visible: localization_amd has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(LOC_ERR_INVALID, "device index out of range");
    loc_snapshot_params prm;
    if (params) prm = *params; else loc_snapshot_default_params(&prm);
    if (prm.jacobian != LOC_JAC_ANALYTIC && prm.jacobian != LOC_JAC_NUMERIC_G2O) return fail(LOC_ERR_INVALID, "jacobian mode");
    const int M4 = (n_anchors + 3) / 4;
    const int M_PAD = 4 * M4;
    if (M_PAD > 16) return fail(LOC_ERR_UNSUPPORTED, "more than 16 anchors per tag");
    int lpi = prm.lanes_per_instance;
    if (lpi == 0) lpi = (M_PAD == 16) ? 2 : 1;  // measured on MI355X: one lane per tag is fastest up to 12 anchors
    if (!locamd::snapshot_supported(M_PAD, lpi)) return fail(LOC_ERR_UNSUPPORTED, "lanes_per_instance for this anchor count");
    if (prm.block_threads == 0) prm.block_threads = 256;
    if (prm.gate_warmup_epochs < 0) return fail(LOC_ERR_INVALID, "gate_warmup_epochs");
    if (prm.block_threads % 64 || prm.block_threads > 256 || prm.block_threads < 64) return fail(LOC_ERR_INVALID, "block_threads");

    LOC_HIP(hipSetDevice(device));
    loc_snapshot* s = new (std::nothrow) loc_snapshot();
    if (!s) return fail(LOC_ERR_INVALID, "out of host memory");
    s->device = device; s->B = batch; s->M = n_anchors; s->M4 = M4; s->M_PAD = M_PAD; s->lpi = lpi; s->prm = prm;
    std::vector<double> anch((size_t)M_PAD * 3, 0.0);
    std::memcpy(anch.data(), anchors_xyz_host, sizeof(double) * 3 * (size_t)n_anchors);
    hipError_t e;
    if ((e = hipMalloc((void**)&s->d_anchors, anch.size() * sizeof(double))) != hipSuccess ||
        (e = hipMalloc((void**)&s->d_pos, sizeof(double) * 3 * (size_t)batch)) != hipSuccess ||
        (e = hipMemcpy(s->d_anchors, anch.data(), anch.size() * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemset(s->d_pos, 0, sizeof(double) * 3 * (size_t)batch)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking)) != hipSuccess) {
        loc_snapshot_destroy(s);
        return fail_hip(e, "loc_snapshot_create");
    }
    *out = s;
    return LOC_OK;
}

static void free_staging(loc_snapshot* s) {
    if (s->d_dist) (void)hipFree(s->d_dist);
    if (s->d_err) (void)hipFree(s->d_err);
    if (s->d_out_pos) (void)hipFree(s->d_out_pos);
    if (s->d_out_chi2) (void)hipFree(s->d_out_chi2);
    if (s->d_out_trials) (void)hipFree(s->d_out_trials);
    s->d_dist = s->d_err = nullptr; s->d_out_pos = s->d_out_chi2 = nullptr; s->d_out_trials = nullptr;
    s->staged_epochs = 0;
}
static int ensure_staging(loc_snapshot* s, int32_t epochs) {
    if (epochs <= s->staged_epochs) return LOC_OK;
    const size_t nf = loc_snapshot_range_floats(s, epochs), B = (size_t)s->B;
    free_staging(s);
    LOC_HIP(hipMalloc((void**)&s->d_dist, nf * sizeof(float)));
    LOC_HIP(hipMalloc((void**)&s->d_err, nf * sizeof(float)));
    LOC_HIP(hipMalloc((void**)&s->d_out_pos, sizeof(double) * 3 * B * (size_t)epochs));
    LOC_HIP(hipMalloc((void**)&s->d_out_chi2, sizeof(double) * B * (size_t)epochs));
    LOC_HIP(hipMalloc((void**)&s->d_out_trials, B * (size_t)epochs));
    s->staged_epochs = epochs;
    return LOC_OK;
}

int loc_snapshot_destroy(loc_snapshot* s) {
    if (!s) return LOC_OK;
    (void)hipSetDevice(s->device);
    free_staging(s);
    for (hipEvent_t ev : s->ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : s->pipe_ev) (void)hipEventDestroy(ev);
    if (s->d_raw_dist) (void)hipFree(s->d_raw_dist);
    if (s->d_raw_err) (void)hipFree(s->d_raw_err);
    if (s->in_stream) (void)hipStreamDestroy(s->in_stream);
    if (s->out_stream) (void)hipStreamDestroy(s->out_stream);
    if (s->d_anchors) (void)hipFree(s->d_anchors);
    if (s->d_pos) (void)hipFree(s->d_pos);
    if (s->own_stream) (void)hipStreamDestroy(s->own_stream);
    delete s;
    return LOC_OK;
}

int64_t loc_snapshot_batch(const loc_snapshot* s) { return s ? s->B : 0; }
int32_t loc_snapshot_anchor_groups(const loc_snapshot* s) { return s ? s->M4 : 0; }
int32_t loc_snapshot_lanes_per_instance(const loc_snapshot* s) { return s ? s->lpi : 0; }
size_t loc_snapshot_range_floats(const loc_snapshot* s, int32_t epochs) {
    if (!s || epochs <= 0) return 0;
    return (size_t)epochs * (size_t)s->M4 * (size_t)s->B * 4u;
}

int loc_snapshot_set_positions(loc_snapshot* s, const double* pos) {
    if (!s || !pos) return fail(LOC_ERR_INVALID, "null");
    LOC_HIP(hipSetDevice(s->device));
    LOC_HIP(hipMemcpy(s->d_pos, pos, sizeof(double) * 3 * (size_t)s->B, hipMemcpyHostToDevice));
    s->epochs_done = 0;
    return LOC_OK;
}
int loc_snapshot_get_positions(loc_snapshot* s, double* pos) {
    if (!s || !pos) return fail(LOC_ERR_INVALID, "null");
    LOC_HIP(hipSetDevice(s->device));
    LOC_HIP(hipMemcpy(pos, s->d_pos, sizeof(double) * 3 * (size_t)s->B, hipMemcpyDeviceToHost));
    return LOC_OK;
}
void* loc_snapshot_positions_device(loc_snapshot* s) { return s ? (void*)s->d_pos : nullptr; }
int64_t loc_snapshot_epochs_done(const loc_snapshot* s) { return s ? s->epochs_done : 0; }
int loc_snapshot_set_epochs_done(loc_snapshot* s, int64_t epochs) {
    if (!s || epochs < 0) return fail(LOC_ERR_INVALID, "epochs");
    s->epochs_done = epochs;
    return LOC_OK;
}

int loc_snapshot_pack_ranges_host(const loc_snapshot* s, int32_t epochs, const float* src, float* dst, float pad_value) {
    if (!s || !src || !dst || epochs <= 0) return fail(LOC_ERR_INVALID, "pack_ranges");
    const size_t B = (size_t)s->B;
    for (int k = 0; k < epochs; ++k)
        for (int g = 0; g < s->M4; ++g)
            for (int j = 0; j < 4; ++j) {
                const int m = 4 * g + j;
                float* d = dst + (((size_t)k * s->M4 + g) * B) * 4 + j;
                if (m < s->M) {
                    const float* sp = src + ((size_t)k * s->M + m) * B;
                    for (size_t b = 0; b < B; ++b) d[4 * b] = sp[b];
                } else {
                    for (size_t b = 0; b < B; ++b) d[4 * b] = pad_value;
                }
            }
    return LOC_OK;
}

int loc_snapshot_solve_device(loc_snapshot* s, int32_t epochs, const float* dist_dev, const float* err_dev,
                              double* out_pos_dev, double* out_chi2_dev, uint8_t* out_trials_dev, void* hip_stream) {
    if (!s) return fail(LOC_ERR_INVALID, "null handle");
    if (epochs <= 0 || !dist_dev || !err_dev || !out_pos_dev || !out_chi2_dev) return fail(LOC_ERR_INVALID, "solve arguments");
    if (((uintptr_t)dist_dev | (uintptr_t)err_dev) & 15u) return fail(LOC_ERR_INVALID, "range tiles must be 16-byte aligned");
    LOC_HIP(hipSetDevice(s->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : s->own_stream;
    locamd::SnapshotArgs a;
    a.dist = dist_dev; a.err = err_dev; a.pos = s->d_pos; a.out_pos = out_pos_dev; a.out_chi2 = out_chi2_dev;
    a.out_trials = out_trials_dev; a.anchors = s->d_anchors; a.B = s->B; a.K = epochs; a.M4 = s->M4;
    a.iterations = s->prm.maximum_iteration; a.gate = s->prm.distance_outlier;
    {
        long long left = (long long)s->prm.gate_warmup_epochs - s->epochs_done;
        a.gate_from_epoch = left > 0 ? (int)(left > epochs ? epochs : left) : 0;
    }
    const bool timed = s->timing && (size_t)(s->ev_used + 2) <= s->ev.size();
    if (timed) LOC_HIP(hipEventRecord(s->ev[s->ev_used], st));
    hipError_t e = locamd::launch_snapshot(a, s->M_PAD, s->lpi, s->prm.jacobian, s->prm.block_threads, st);
    if (e != hipSuccess) return fail_hip(e, "launch_snapshot");
    if (timed) { LOC_HIP(hipEventRecord(s->ev[s->ev_used + 1], st)); s->ev_used += 2; }
    s->epochs_done += epochs;
    return LOC_OK;
}

int loc_snapshot_solve_host(loc_snapshot* s, int32_t epochs, const float* dist_h, const float* err_h,
                            double* out_pos_h, double* out_chi2_h, uint8_t* out_trials_h) {
    if (!s) return fail(LOC_ERR_INVALID, "null handle");
    if (epochs <= 0 || !dist_h || !err_h || !out_pos_h || !out_chi2_h) return fail(LOC_ERR_INVALID, "solve arguments");
    LOC_HIP(hipSetDevice(s->device));
    const size_t nf = loc_snapshot_range_floats(s, epochs);
    const size_t B = (size_t)s->B;
    if (int rc = ensure_staging(s, epochs)) return rc;
    LOC_HIP(hipMemcpyAsync(s->d_dist, dist_h, nf * sizeof(float), hipMemcpyHostToDevice, s->own_stream));
    LOC_HIP(hipMemcpyAsync(s->d_err, err_h, nf * sizeof(float), hipMemcpyHostToDevice, s->own_stream));
    int rc = loc_snapshot_solve_device(s, epochs, s->d_dist, s->d_err, s->d_out_pos, s->d_out_chi2, s->d_out_trials, s->own_stream);
    if (rc != LOC_OK) return rc;
    LOC_HIP(hipMemcpyAsync(out_pos_h, s->d_out_pos, sizeof(double) * 3 * B * (size_t)epochs, hipMemcpyDeviceToHost, s->own_stream));
    LOC_HIP(hipMemcpyAsync(out_chi2_h, s->d_out_chi2, sizeof(double) * B * (size_t)epochs, hipMemcpyDeviceToHost, s->own_stream));
    if (out_trials_h) LOC_HIP(hipMemcpyAsync(out_trials_h, s->d_out_trials, B * (size_t)epochs, hipMemcpyDeviceToHost, s->own_stream));
    LOC_HIP(hipStreamSynchronize(s->own_stream));
    return LOC_OK;
}

int loc_snapshot_solve_host_kmb(loc_snapshot* s, int32_t epochs, const float* dist_kmb, const float* err_kmb,
                                double* out_pos_h, double* out_chi2_h, uint8_t* out_trials_h) {
    if (!s) return fail(LOC_ERR_INVALID, "null handle");
    if (epochs <= 0 || !dist_kmb || !err_kmb || !out_pos_h || !out_chi2_h) return fail(LOC_ERR_INVALID, "solve arguments");
    LOC_HIP(hipSetDevice(s->device));
    const size_t B = (size_t)s->B, M = (size_t)s->M, M4 = (size_t)s->M4;
    if (int rc = ensure_staging(s, epochs)) return rc;
    if (epochs > s->raw_epochs) {
        if (s->d_raw_dist) (void)hipFree(s->d_raw_dist);
        if (s->d_raw_err) (void)hipFree(s->d_raw_err);
        s->d_raw_dist = s->d_raw_err = nullptr; s->raw_epochs = 0;
        LOC_HIP(hipMalloc((void**)&s->d_raw_dist, sizeof(float) * M * B * (size_t)epochs));
        LOC_HIP(hipMalloc((void**)&s->d_raw_err, sizeof(float) * M * B * (size_t)epochs));
        s->raw_epochs = epochs;
    }
    if (!s->in_stream) LOC_HIP(hipStreamCreateWithFlags(&s->in_stream, hipStreamNonBlocking));
    if (!s->out_stream) LOC_HIP(hipStreamCreateWithFlags(&s->out_stream, hipStreamNonBlocking));
    // chunks of ~8 MB per input array: small enough that copy-in, solve and copy-out of neighbouring chunks overlap,
    // large enough that a chunk's launch fills the GPU (epochs of one chunk stay sequential per tag inside the kernel)
    // (pageable buffers cannot overlap anyway — the runtime stages them synchronously — so they go as one chunk)
    auto pinned = [](const void* p) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
        return at.type == hipMemoryTypeHost;
    };
    const bool overlap = pinned(dist_kmb) && pinned(err_kmb) && pinned(out_pos_h) && pinned(out_chi2_h);
    const int ce = overlap ? (int)std::max<size_t>(1, (8u << 20) / (M * B * sizeof(float))) : epochs;
    const int nchunks = (epochs + ce - 1) / ce;
    while ((int)s->pipe_ev.size() < 2 * nchunks) {
        hipEvent_t ev;
        LOC_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        s->pipe_ev.push_back(ev);
    }
    for (int c = 0; c < nchunks; ++c) {
        const int k0 = c * ce, kc = std::min(ce, epochs - k0);
        const size_t roff = (size_t)k0 * M * B, rn = (size_t)kc * M * B, toff = (size_t)k0 * M4 * B * 4;
        LOC_HIP(hipMemcpyAsync(s->d_raw_dist + roff, dist_kmb + roff, rn * sizeof(float), hipMemcpyHostToDevice, s->in_stream));
        LOC_HIP(hipMemcpyAsync(s->d_raw_err + roff, err_kmb + roff, rn * sizeof(float), hipMemcpyHostToDevice, s->in_stream));
        LOC_HIP(hipEventRecord(s->pipe_ev[2 * c], s->in_stream));
        LOC_HIP(hipStreamWaitEvent(s->own_stream, s->pipe_ev[2 * c], 0));
        hipError_t e = locamd::launch_pack_kmb(s->d_raw_dist + roff, s->d_dist + toff, s->B, s->M, s->M4, kc, 0.f, s->own_stream);
        if (e == hipSuccess) e = locamd::launch_pack_kmb(s->d_raw_err + roff, s->d_err + toff, s->B, s->M, s->M4, kc, 0.f, s->own_stream);
        if (e != hipSuccess) return fail_hip(e, "launch_pack_kmb");
        int rc = loc_snapshot_solve_device(s, kc, s->d_dist + toff, s->d_err + toff, s->d_out_pos + (size_t)k0 * 3 * B,
                                           s->d_out_chi2 + (size_t)k0 * B, s->d_out_trials + (size_t)k0 * B, s->own_stream);
        if (rc != LOC_OK) return rc;
        LOC_HIP(hipEventRecord(s->pipe_ev[2 * c + 1], s->own_stream));
        LOC_HIP(hipStreamWaitEvent(s->out_stream, s->pipe_ev[2 * c + 1], 0));
        LOC_HIP(hipMemcpyAsync(out_pos_h + (size_t)k0 * 3 * B, s->d_out_pos + (size_t)k0 * 3 * B, sizeof(double) * 3 * B * (size_t)kc, hipMemcpyDeviceToHost, s->out_stream));
        LOC_HIP(hipMemcpyAsync(out_chi2_h + (size_t)k0 * B, s->d_out_chi2 + (size_t)k0 * B, sizeof(double) * B * (size_t)kc, hipMemcpyDeviceToHost, s->out_stream));
        if (out_trials_h) LOC_HIP(hipMemcpyAsync(out_trials_h + (size_t)k0 * B, s->d_out_trials + (size_t)k0 * B, B * (size_t)kc, hipMemcpyDeviceToHost, s->out_stream));
    }
    LOC_HIP(hipStreamSynchronize(s->out_stream));
    LOC_HIP(hipStreamSynchronize(s->own_stream));
    return LOC_OK;
}

int loc_host_alloc(void** out, size_t bytes) {
    if (!out || bytes == 0) return fail(LOC_ERR_INVALID, "loc_host_alloc");
    *out = nullptr;
    LOC_HIP(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return LOC_OK;
}
int loc_host_free(void* p) {
    if (!p) return LOC_OK;
    LOC_HIP(hipHostFree(p));
    return LOC_OK;
}

int loc_snapshot_timing_begin(loc_snapshot* s, int32_t max_launches) {
    if (!s || max_launches <= 0) return fail(LOC_ERR_INVALID, "timing_begin");
    LOC_HIP(hipSetDevice(s->device));
    while ((int)s->ev.size() < 2 * max_launches) {
        hipEvent_t ev;
        LOC_HIP(hipEventCreate(&ev));
        s->ev.push_back(ev);
    }
    s->ev_used = 0;
    s->timing = true;
    return LOC_OK;
}
int loc_snapshot_timing_end(loc_snapshot* s, int32_t* n_launches, double* total_ms, double* avg_ms) {
    if (!s) return fail(LOC_ERR_INVALID, "timing_end");
    LOC_HIP(hipSetDevice(s->device));
    s->timing = false;
    double tot = 0;
    const int n = s->ev_used / 2;
    for (int i = 0; i < n; ++i) {
        LOC_HIP(hipEventSynchronize(s->ev[2 * i + 1]));
        float ms = 0;
        LOC_HIP(hipEventElapsedTime(&ms, s->ev[2 * i], s->ev[2 * i + 1]));
        tot += ms;
    }
    if (n_launches) *n_launches = n;
    if (total_ms) *total_ms = tot;
    if (avg_ms) *avg_ms = n ? tot / n : 0.0;
    s->ev_used = 0;
    return LOC_OK;
}

}  // extern "C"
