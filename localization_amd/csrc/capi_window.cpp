// C ABI of the batched sliding-window solver (see include/localization_amd.h). Host side only.
#include "../../include/localization_amd.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <algorithm>
#include <chrono>
#include <new>
#include <atomic>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "window_kernel.h"

extern int locamd_fail(int code, const char* what);
extern int locamd_fail_hip(hipError_t e, const char* where);
#define LOC_HIP(expr)                                              \
    do {                                                           \
        hipError_t _e = (expr);                                    \
        if (_e != hipSuccess) return locamd_fail_hip(_e, #expr);   \
    } while (0)

struct loc_window {
    int device = 0;
    long long B = 0;
    locamd::WindowCaps caps{};
    int n_anchors = 0, anchors_cap = 0;
    int iterations = 10;
    int jacobian = LOC_JAC_NUMERIC_G2O, natural_order = 0;   // default = the reference's configuration
    double* d_anchors = nullptr;
    std::vector<double> h_anchors;   // handles of <= 4 windows (a node's own): the table as last set; the device copy follows on demand
    bool anchors_dirty = false;
    int32_t *d_counts = nullptr, *d_ridx = nullptr, *d_pidx = nullptr, *d_sidx = nullptr;
    double *d_poses = nullptr, *d_rval = nullptr, *d_pval = nullptr, *d_sval = nullptr, *d_result = nullptr;
    double* d_workspace = nullptr;  // HBM copy of the (H, L) matrices when they do not fit LDS
    double* d_poses_in = nullptr;   // resident mode: the uploaded initial estimates (every resident solve starts from them)
    double* d_chain_ws = nullptr;   // chain windows (one lane per window, window_kernel.hip: chain_lm_kernel): its workspace
    double* d_chain3_ws = nullptr;  // translation-only chain windows (chain3_kernel.hip)
    // Host-built tables of the kernels that need them (tree_wave / tree_lm: the elimination schedule; arrow3: row order + packed edge
    // records).  Two sets: [0] for loc_window_solve_host calls, [1] owned by the resident batch from its upload on — a host-path solve in
    // between must not disturb what the next loc_window_solve_resident walks.
    struct WinAux {
        int32_t* d_tsched = nullptr;
        size_t tsched_cap = 0;
        std::vector<int32_t> h_tsched;
        locamd::TreeSched tsched{};
        int32_t *d_ahdr = nullptr, *d_arslot = nullptr;
        double *d_arec = nullptr, *d_aprec = nullptr;
        size_t arec_cap = 0, aprec_cap = 0;   // doubles allocated
        int arrow_nb_max = 0, arrow_jmax = 0, arrow_jpmax = 0;
        int arrow_jch[16] = {0}, arrow_jpch[16] = {0};   // records per chunk of 64 rows (the most any row of the chunk has, over the batch)
        std::vector<int32_t> h_ahdr, h_arslot;
        std::vector<double> h_arec, h_aprec;
    } aux[2];
    double* d_tree_ws = nullptr;    // workspaces (used only while a launch runs)
    double* d_arrow_ws = nullptr;
    int arrow_ws_nb = 0;            // border size d_arrow_ws was allocated for
    double* d_roff1 = nullptr;      // optional lever arms of endpoint 1 (loc_window_set_endpoint1_offsets), [B][nr_max][3]
    bool has_off1 = false;
    int resident_topology = 0;      // LOC_WINDOW_KERNEL_* the uploaded batch qualifies for by its structure (the batch-size threshold is applied per solve)
    long long chain_min = -1;       // smallest batch that takes a lane-per-window kernel (-1: the default / LOCAMD_CHAIN_MIN_BATCH)
    long long n_resident = 0;
    int resident_min_anchors = 0;   // anchors the resident batch references (loc_window_set_anchors may not shrink below it)
    bool resident_solved = false;   // a resident solve has run since the upload (loc_window_download has something to fetch)
    int last_kind = -1;             // LOC_WINDOW_KERNEL_* of the last launch
    hipEvent_t resident_done = nullptr;  // recorded after every resident launch on the stream it ran on: whatever touches the shared
    bool resident_inflight = false;      // device state waits for THIS (the caller's stream is not kept: it may be destroyed any time)
    // kernel-selection switches: the environment is read ONCE, here at creation (loc_window_set_option changes them afterwards)
    struct Opts {
        long long env_chain_min = 12288;   // LOCAMD_CHAIN_MIN_BATCH, or the default
        bool env_chain_min_set = false;
        int arrow3 = -1;                   // LOCAMD_ARROW3: -1 default (windows of more than 64 poses), 0 never, 1 whenever the batch qualifies
        int tree = -1;                     // LOCAMD_TREE: -1 default, 0 never, 2 the lane-per-window variant
        bool wave3 = true, wave6 = true, chain3 = true, zero_copy = true, topology_cache = true;
        bool kernel_events = true;         // "kernel_events" 0: no HIP events around the launch of a zero-copy solve (loc_window_last_kernel_ms then reports launch-to-completion on the host clock)
    } opt;
    // structural verdict of the last host-path batch, keyed on a hash of (n, counts, index tables): a caller that replays one graph
    // with new measurements (the node's window between two slides, a Monte-Carlo batch) skips the chain / forest / arrowhead tests
    struct TopoCache {
        bool valid = false;
        unsigned long long key = 0;
        int64_t n = 0;
        bool chain = false, single_pairs = false, se3_pairs = false, tree_ok = false, tree_tried = false;
    } topo_cache;
    double t_validate_ms = 0, t_topology_ms = 0, t_run_ms = 0;   // loc_window_last_host_timing
    bool t_cached = false;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double last_ms = 0.0;
    std::vector<hipEvent_t> ev;     // loc_window_timing_*: one pair per resident launch
    int ev_used = 0;
    bool timing = false;
    // small calls (a node's single window): all inputs travel as one page-locked block, all outputs as another
    char *h_stage = nullptr, *d_stage = nullptr;
};
static constexpr size_t kStageBytes = 4u << 20;

// The host passes over a batch (validation, structure hash, chain / translation-only scans) are O(instances x edges) and run in front of a
// kernel of a millisecond or two: batches of >= 4 096 instances are split over up to eight threads (f(lo, hi) on disjoint instance ranges).
template <class F>
static void parallel_chunks(int64_t n, F&& f) {
    const unsigned hw = std::thread::hardware_concurrency();
    const int nt = n >= 4096 ? (int)std::min<unsigned>(8u, hw ? hw : 1u) : 1;
    if (nt <= 1) { f((int64_t)0, n, 0); return; }
    const int64_t per = (n + nt - 1) / nt;
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back([&f, t, per, n] { f(std::min(n, t * per), std::min(n, (t + 1) * per), t); });
    f((int64_t)0, std::min(n, per), 0);
    for (auto& x : th) x.join();
}

extern "C" {

static locamd::WindowCaps to_caps(const loc_window_caps* caps) {
    int bw = caps->bw_max < 0 ? caps->nv_max - 1 : caps->bw_max;
    if (bw > caps->nv_max - 1) bw = caps->nv_max - 1;
    if (bw < 0) bw = 0;
    return locamd::WindowCaps{caps->nv_max, caps->nr_max, caps->np_max, caps->ns_max, bw};
}

size_t loc_window_lds_bytes(const loc_window_caps* caps) {
    if (!caps) return 0;
    if (caps->nv_max <= 0) return 0;
    locamd::WindowCaps c = to_caps(caps);
    const size_t in_lds = locamd::window_lds_bytes(c, false);
    return (in_lds <= 160 * 1024 - 512 && c.nv_max <= 64) ? in_lds : locamd::window_lds_bytes(c, true) + 36 * sizeof(double);  // large windows: index tables + exchange block; the rest in the HBM workspace
}

int loc_window_destroy(loc_window* w) {
    if (!w) return LOC_OK;
    (void)hipSetDevice(w->device);
    void* ptrs[] = {w->d_anchors, w->d_counts, w->d_ridx, w->d_pidx, w->d_sidx, w->d_poses, w->d_rval, w->d_pval, w->d_sval, w->d_result, w->d_workspace, w->d_poses_in,
                    w->d_chain_ws, w->d_chain3_ws, w->d_roff1, w->d_tree_ws, w->d_arrow_ws,
                    w->aux[0].d_tsched, w->aux[0].d_ahdr, w->aux[0].d_arslot, w->aux[0].d_arec, w->aux[0].d_aprec,
                    w->aux[1].d_tsched, w->aux[1].d_ahdr, w->aux[1].d_arslot, w->aux[1].d_arec, w->aux[1].d_aprec};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (hipEvent_t e : w->ev) (void)hipEventDestroy(e);
    if (w->h_stage) (void)hipHostFree(w->h_stage);
    if (w->d_stage) (void)hipFree(w->d_stage);
    if (w->ev0) (void)hipEventDestroy(w->ev0);
    if (w->ev1) (void)hipEventDestroy(w->ev1);
    if (w->resident_done) (void)hipEventDestroy(w->resident_done);
    if (w->stream) (void)hipStreamDestroy(w->stream);
    delete w;
    return LOC_OK;
}

int loc_window_create(loc_window** out, int32_t device, int64_t batch, const loc_window_caps* caps,
                      int32_t n_anchors, const double* anchors, int32_t maximum_iteration) {
    if (!out) return locamd_fail(LOC_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (batch <= 0 || !caps || n_anchors < 0 || (n_anchors > 0 && !anchors)) return locamd_fail(LOC_ERR_INVALID, "window arguments");
    if (caps->nv_max <= 0 || caps->nv_max > 4096 || caps->nr_max < 0 || caps->np_max < 0 || caps->ns_max < 0)
        return locamd_fail(LOC_ERR_UNSUPPORTED, "window capacities (1 <= nv_max <= 4096)");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return locamd_fail(LOC_ERR_NO_DEVICE, "no HIP device visible: localization_amd has no CPU fallback");
    if (device < 0 || device >= ndev) return locamd_fail(LOC_ERR_INVALID, "device index out of range");
    LOC_HIP(hipSetDevice(device));
    loc_window* w = new (std::nothrow) loc_window();
    if (!w) return locamd_fail(LOC_ERR_INVALID, "out of host memory");
    w->device = device; w->B = batch; w->n_anchors = n_anchors; w->anchors_cap = n_anchors > 0 ? n_anchors : 1; w->iterations = maximum_iteration;
    w->caps = to_caps(caps);
    // windows of more than 64 poses always keep their arrays in the HBM workspace (their structure tables alone fill the LDS)
    const bool global_a = locamd::window_lds_bytes(w->caps, false) > 160 * 1024 - 512 || w->caps.nv_max > 64;
    const size_t B = (size_t)batch;
    const size_t na = (size_t)(n_anchors > 0 ? n_anchors : 1);
    hipError_t e;
    auto alloc = [&](void** p, size_t bytes) { return hipMalloc(p, bytes ? bytes : 8); };
    if ((e = alloc((void**)&w->d_anchors, na * 3 * sizeof(double))) != hipSuccess ||
        (e = alloc((void**)&w->d_counts, B * 4 * sizeof(int32_t))) != hipSuccess ||
        (e = alloc((void**)&w->d_poses, B * caps->nv_max * 12 * sizeof(double))) != hipSuccess ||
        (e = alloc((void**)&w->d_ridx, B * caps->nr_max * 2 * sizeof(int32_t))) != hipSuccess ||
        (e = alloc((void**)&w->d_rval, B * caps->nr_max * 5 * sizeof(double))) != hipSuccess ||
        (e = alloc((void**)&w->d_pidx, B * caps->np_max * sizeof(int32_t))) != hipSuccess ||
        (e = alloc((void**)&w->d_pval, B * caps->np_max * 18 * sizeof(double))) != hipSuccess ||
        (e = alloc((void**)&w->d_sidx, B * caps->ns_max * 4 * sizeof(int32_t))) != hipSuccess ||
        (e = alloc((void**)&w->d_sval, B * caps->ns_max * 48 * sizeof(double))) != hipSuccess ||
        (e = alloc((void**)&w->d_result, B * 8 * sizeof(double))) != hipSuccess ||
        (global_a && (e = hipMalloc((void**)&w->d_workspace, B * locamd::window_workspace_doubles(w->caps) * sizeof(double))) != hipSuccess) ||
        (e = hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&w->ev0)) != hipSuccess || (e = hipEventCreate(&w->ev1)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&w->resident_done, hipEventDisableTiming)) != hipSuccess ||
        (n_anchors > 0 && (e = hipMemcpy(w->d_anchors, anchors, (size_t)n_anchors * 3 * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess)) {
        loc_window_destroy(w);
        return locamd_fail_hip(e, "loc_window_create");
    }
    if (batch <= 4 && n_anchors > 0) w->h_anchors.assign(anchors, anchors + (size_t)n_anchors * 3);
    {   // the A/B switches of the environment, read once
        loc_window::Opts& o = w->opt;
        if (const char* v = getenv("LOCAMD_CHAIN_MIN_BATCH")) { o.env_chain_min = atoll(v); o.env_chain_min_set = true; }
        if (const char* v = getenv("LOCAMD_ARROW3")) o.arrow3 = v[0] == '1' ? 1 : 0;
        if (const char* v = getenv("LOCAMD_TREE")) o.tree = v[0] == '0' ? 0 : (v[0] == 'l' ? 2 : -1);
        if (const char* v = getenv("LOCAMD_WAVE3")) o.wave3 = v[0] != '0';
        if (const char* v = getenv("LOCAMD_WAVE6")) o.wave6 = v[0] != '0';
        if (const char* v = getenv("LOCAMD_CHAIN3")) o.chain3 = v[0] != '0';
        if (getenv("LOCAMD_NO_ZERO_COPY")) o.zero_copy = false;
        if (const char* v = getenv("LOCAMD_KERNEL_EVENTS")) o.kernel_events = atoi(v) != 0;
    }
    *out = w;
    return LOC_OK;
}

// whatever is about to touch the device state a resident launch reads or writes waits for that launch first
static int wait_resident(loc_window* w) {
    if (!w->resident_inflight) return LOC_OK;
    LOC_HIP(hipEventSynchronize(w->resident_done));
    w->resident_inflight = false;
    return LOC_OK;
}

// the device copy of the anchor table
static int upload_anchors(loc_window* w, int32_t n_anchors, const double* anchors) {
    LOC_HIP(hipSetDevice(w->device));
    if (int rc = wait_resident(w)) return rc;   // a resident launch may still be reading the table
    if (n_anchors > w->anchors_cap) {
        double* p = nullptr;
        LOC_HIP(hipMalloc((void**)&p, (size_t)n_anchors * 3 * sizeof(double)));
        if (w->d_anchors) (void)hipFree(w->d_anchors);
        w->d_anchors = p;
        w->anchors_cap = n_anchors;
    }
    if (n_anchors > 0) LOC_HIP(hipMemcpy(w->d_anchors, anchors, (size_t)n_anchors * 3 * sizeof(double), hipMemcpyHostToDevice));
    w->n_anchors = n_anchors;
    w->anchors_dirty = false;
    return LOC_OK;
}
static int flush_anchors(loc_window* w) {
    if (!w->anchors_dirty) return LOC_OK;
    return upload_anchors(w, w->n_anchors, w->h_anchors.data());
}

int loc_window_set_anchors(loc_window* w, int32_t n_anchors, const double* anchors) {
    if (!w || n_anchors < 0 || (n_anchors > 0 && !anchors)) return locamd_fail(LOC_ERR_INVALID, "set_anchors");
    if (w->n_resident > 0 && n_anchors < w->resident_min_anchors)
        return locamd_fail(LOC_ERR_INVALID, "set_anchors: the resident batch references more anchors than the new table holds (upload again first)");
    if (w->B <= 4) {
        // a node's handle: its table changes with every message (the window slides), and its solve usually reads the table from the
        // page-locked staging block (loc_window_solve_host) — the device copy is refreshed only when a launch needs it
        w->h_anchors.assign(anchors, anchors + (size_t)n_anchors * 3);
        w->n_anchors = n_anchors;
        w->anchors_dirty = true;
        return LOC_OK;
    }
    return upload_anchors(w, n_anchors, anchors);
}

// host-side shape check: a bad index would fault the GPU
static int validate_instances(const loc_window* w, int64_t n, const int32_t* counts, const double* poses, const int32_t* r_idx,
                              const double* r_val, const int32_t* p_idx, const double* p_val, const int32_t* s_idx,
                              const double* s_val) {
    if (!w || !counts || !poses) return locamd_fail(LOC_ERR_INVALID, "window solve arguments");
    if (n <= 0 || n > w->B) return locamd_fail(LOC_ERR_INVALID, "n_instances");
    const locamd::WindowCaps& c = w->caps;
    if ((c.nr_max && (!r_idx || !r_val)) || (c.np_max && (!p_idx || !p_val)) || (c.ns_max && (!s_idx || !s_val)))
        return locamd_fail(LOC_ERR_INVALID, "missing edge arrays");
    // (the error message is set on the calling thread: the workers only report WHICH check failed first in their range)
    static const char* const kWhat[] = {nullptr, "counts exceed capacities", "range edge vertex index", "range edge couples poses further apart than bw_max",
                                        "prior edge vertex index", "SE3 edge vertex index", "SE3 edge couples poses further apart than bw_max"};
    std::atomic<int> first_bad{0};
    parallel_chunks(n, [&](int64_t lo, int64_t hi, int) {
        auto check = [&](int64_t i) -> int {
            const int32_t* cn = counts + i * 4;
            if (cn[0] < 0 || cn[0] > c.nv_max || cn[1] < 0 || cn[1] > c.nr_max || cn[2] < 0 || cn[2] > c.np_max || cn[3] < 0 || cn[3] > c.ns_max) return 1;
            for (int e = 0; e < cn[1]; ++e) {
                const int32_t* ix = r_idx + ((size_t)i * c.nr_max + e) * 2;
                if (ix[0] < 0 || ix[0] >= cn[0] || ix[1] >= cn[0] || ix[1] < -w->n_anchors || ix[0] == ix[1]) return 2;
                if (ix[1] >= 0 && (ix[0] - ix[1] > c.bw_max || ix[1] - ix[0] > c.bw_max)) return 3;
            }
            for (int e = 0; e < cn[2]; ++e) {
                const int32_t v = p_idx[(size_t)i * c.np_max + e];
                if (v < 0 || v >= cn[0]) return 4;
            }
            for (int e = 0; e < cn[3]; ++e) {
                const int32_t* ix = s_idx + ((size_t)i * c.ns_max + e) * 4;
                if (ix[0] < 0 || ix[0] >= cn[0] || ix[1] < 0 || ix[1] >= cn[0] || ix[0] == ix[1]) return 5;
                if (ix[0] - ix[1] > c.bw_max || ix[1] - ix[0] > c.bw_max) return 6;
            }
            return 0;
        };
        for (int64_t i = lo; i < hi && first_bad.load(std::memory_order_relaxed) == 0; ++i) {
            const int bad = check(i);
            if (bad) { int zero = 0; first_bad.compare_exchange_strong(zero, bad); return; }
        }
    });
    if (const int bad = first_bad.load()) return locamd_fail(LOC_ERR_INVALID, kWhat[bad]);
    return LOC_OK;
}

// Large batches of CHAIN windows (every pose-to-pose edge — range or SE3 — joins consecutive poses; edges ordered by their
// later pose and priors by pose — the order Localization::addRangeEdge / addImuEdge create them in) run one lane per window
// (chain_lm_kernel; chain3_lm_kernel when the batch is translation-only).  Below the threshold a wave per window is faster (the
// lane-per-window kernels take about as long for 1 000 windows as for 65 536); LOCAMD_CHAIN_MIN_BATCH in the environment moves it
// (0 = never).
static long long chain_min_batch(const loc_window* w) { return w->opt.env_chain_min; }
// translation-only (the exact 3-DoF reduction, chain3_kernel.hip / arrow3_kernel.hip): no EdgeSE3, every lever arm zero, every
// rotation the identity, priors with an identity measurement rotation and no rotation information
static bool translation_only(const loc_window* w, int64_t n, const int32_t* counts, const double* poses, const double* r_val, const double* p_val) {
    const locamd::WindowCaps& c = w->caps;
    static const double I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (w->n_anchors > 500000) return false;   // (the packed endpoint word of chain3 holds 2^19 anchors)
    std::atomic<bool> all{true};
    parallel_chunks(n, [&](int64_t lo, int64_t hi, int) {
        auto one = [&](int64_t i) {
            const int32_t* cn = counts + i * 4;
            if (cn[3] != 0 || cn[0] > 1048575) return false;
            for (int e = 0; e < cn[1]; ++e) {
                const double* v = r_val + ((size_t)i * c.nr_max + e) * 5;
                if (v[2] != 0.0 || v[3] != 0.0 || v[4] != 0.0) return false;
            }
            for (int p = 0; p < cn[0]; ++p)
                if (std::memcmp(poses + ((size_t)i * c.nv_max + p) * 12, I9, sizeof(I9)) != 0) return false;
            for (int e = 0; e < cn[2]; ++e) {
                const double* v = p_val + ((size_t)i * c.np_max + e) * 18;
                if (std::memcmp(v, I9, sizeof(I9)) != 0 || v[15] != 0.0 || v[16] != 0.0 || v[17] != 0.0) return false;
            }
            return true;
        };
        for (int64_t i = lo; i < hi && all.load(std::memory_order_relaxed); ++i)
            if (!one(i)) { all.store(false); return; }
    });
    return all.load();
}

// CHAIN + BORDER ("arrowhead": BASELINE config 4, anchor self-calibration — a tag trajectory whose poses range to a few nodes that
// are unknowns themselves, localization.cpp:94-98).  The border of an instance = its last nb0 pose slots, nb0 = the smallest number
// such that every pose-to-pose edge between NON-consecutive slots has an endpoint there; the other poses form the chain (one edge
// per consecutive pair at most).  The chain is cut into up to four segments at separator poses, which join the border (one level of
// nested dissection: arrow3_lm_kernel sweeps the segments with one wave each).  Rows = chain rows segment by segment, then border
// rows (separators first, then the original border in slot order).  A chain row owns its edges to anchors, to border poses and
// to the previous chain row; a border row those to lower-index border poses and to anchors.  Every row's edges (creation order)
// and priors are packed as records [chunk of 64 rows][slot][lane].
static bool build_arrow_aux(loc_window* w, int which, int64_t n, const int32_t* counts, const int32_t* r_idx, const double* r_val, const int32_t* p_idx,
                            const double* p_val) {
    loc_window::WinAux& A = w->aux[which];
    const locamd::WindowCaps& c = w->caps;
    const int NW = 4;
    const int nchunk = (c.nv_max + 63) / 64;
    A.h_ahdr.assign((size_t)n * 8, 0);
    A.h_arslot.assign((size_t)n * c.nv_max, 0);
    std::vector<int32_t> cls, nedge, nprior, pairs;
    std::vector<int> seps;
    // pass 1: structure, record counts
    int nb_max = 0, jmax = 1, jpmax = 1;
    int jch[16], jpch[16];
    for (int k = 0; k < 16; ++k) { jch[k] = 1; jpch[k] = 1; }
    for (int64_t i = 0; i < n; ++i) {
        const int32_t* cn = counts + i * 4;
        const int nv = cn[0], nr = cn[1], np = cn[2];
        const int32_t* ri = r_idx + (size_t)i * c.nr_max * 2;
        int nb0 = 0;
        for (int e = 0; e < nr; ++e) {
            const int v0 = ri[2 * e], v1 = ri[2 * e + 1];
            if (v1 < 0) {
                if (-1 - v1 >= locamd::kArrowMaxAnchors) return false;   // (the kernel keeps the anchor table in LDS)
                continue;
            }
            const int hi = v0 > v1 ? v0 : v1, lo = v0 > v1 ? v1 : v0;
            if (hi - lo != 1 && nv - hi > nb0) nb0 = nv - hi;
        }
        if (nb0 < 1 || nb0 > 12 || nv - nb0 < 2) return false;
        const int n0 = nv - nb0;
        int nseg = n0 / 24;
        if (nseg > NW) nseg = NW;
        if (nseg < 1) nseg = 1;
        const int nb = nb0 + nseg - 1, nc = n0 - (nseg - 1);
        cls.assign((size_t)nv, 0);
        seps.clear();
        for (int k = 1; k < nseg; ++k) seps.push_back((int)((long long)k * n0 / nseg));
        int32_t* hdr = A.h_ahdr.data() + (size_t)i * 8;
        int32_t* rslot = A.h_arslot.data() + (size_t)i * c.nv_max;
        hdr[0] = nb; hdr[1] = nseg; hdr[2] = nc; hdr[3] = 0;
        int q = 0, si = 0;
        for (int v = 0; v < n0; ++v) {
            if (si < (int)seps.size() && v == seps[si]) { cls[v] = -1 - si; rslot[nc + si] = v; ++si; hdr[3 + si] = q; continue; }
            cls[v] = q; rslot[q] = v; ++q;
        }
        for (int s2 = nseg; s2 <= NW; ++s2) hdr[3 + s2] = nc;   // (segments nseg .. NW-1 are empty)
        for (int v = n0; v < nv; ++v) { const int b = (nseg - 1) + (v - n0); cls[v] = -1 - b; rslot[nc + b] = v; }
        // owners
        nedge.assign((size_t)nv, 0); nprior.assign((size_t)nv, 0); pairs.assign((size_t)nc + 1, 0);
        for (int e = 0; e < nr; ++e) {
            const int v0 = ri[2 * e], v1 = ri[2 * e + 1];
            int row;
            if (v1 < 0) row = cls[v0] >= 0 ? cls[v0] : nc + (-1 - cls[v0]);
            else {
                const int c0 = cls[v0], c1 = cls[v1];
                if (c0 >= 0 && c1 >= 0) {
                    if (c0 - c1 != 1 && c1 - c0 != 1) return false;   // (cannot happen: non-consecutive edges end in the border)
                    row = c0 > c1 ? c0 : c1;
                    if (++pairs[row] > 1) return false;                // one edge per consecutive chain pair
                } else if (c0 >= 0) row = c0;
                else if (c1 >= 0) row = c1;
                else row = nc + ((-1 - c0) > (-1 - c1) ? (-1 - c0) : (-1 - c1));
            }
            if (++nedge[row] > jmax) jmax = nedge[row];
            if (row / 64 < 16 && nedge[row] > jch[row / 64]) jch[row / 64] = nedge[row];
        }
        const int32_t* pi = p_idx + (size_t)i * c.np_max;
        for (int e = 0; e < np; ++e) {
            const int cv = cls[pi[e]], row = cv >= 0 ? cv : nc + (-1 - cv);
            if (++nprior[row] > jpmax) jpmax = nprior[row];
            if (row / 64 < 16 && nprior[row] > jpch[row / 64]) jpch[row / 64] = nprior[row];
        }
        if (nb > nb_max) nb_max = nb;
    }
    if (jmax > 64 || jpmax > 16 || nb_max > 15) return false;
    if (locamd::window_arrow3_lds_bytes(c, nb_max) > 160 * 1024 - 512) return false;
    // pass 2: the records
    const size_t rec_per = (size_t)nchunk * jmax * 64 * 3, prec_per = (size_t)nchunk * jpmax * 64 * 7;
    A.h_arec.assign((size_t)n * rec_per, -1.0);
    A.h_aprec.assign((size_t)n * prec_per, 0.0);
    for (int64_t i = 0; i < n; ++i) {
        const int32_t* cn = counts + i * 4;
        const int nv = cn[0], nr = cn[1], np = cn[2];
        const int32_t* ri = r_idx + (size_t)i * c.nr_max * 2;
        const double* rv = r_val + (size_t)i * c.nr_max * 5;
        const int32_t* hdr = A.h_ahdr.data() + (size_t)i * 8;
        const int32_t* rslot = A.h_arslot.data() + (size_t)i * c.nv_max;
        const int nb = hdr[0], nc = hdr[2];
        cls.assign((size_t)nv, 0);
        for (int r = 0; r < nc + nb; ++r) cls[rslot[r]] = r < nc ? r : -1 - (r - nc);
        nedge.assign((size_t)nv, 0); nprior.assign((size_t)nv, 0);
        double* rec = A.h_arec.data() + (size_t)i * rec_per;
        for (int e = 0; e < nr; ++e) {
            const int v0 = ri[2 * e], v1 = ri[2 * e + 1];
            int row, kind, idx, own0;
            if (v1 < 0) { row = cls[v0] >= 0 ? cls[v0] : nc + (-1 - cls[v0]); kind = 0; idx = -1 - v1; own0 = 1; }
            else {
                const int c0 = cls[v0], c1 = cls[v1];
                if (c0 >= 0 && c1 >= 0) { row = c0 > c1 ? c0 : c1; kind = 1; idx = 0; own0 = c0 > c1; }
                else if (c0 >= 0) { row = c0; kind = 2; idx = -1 - c1; own0 = 1; }
                else if (c1 >= 0) { row = c1; kind = 2; idx = -1 - c0; own0 = 0; }
                else {
                    const int b0 = -1 - c0, b1 = -1 - c1;
                    if (b0 == b1) return false;
                    row = nc + (b0 > b1 ? b0 : b1); kind = 2; idx = b0 > b1 ? b1 : b0; own0 = b0 > b1;
                }
            }
            double* q = rec + (((size_t)(row / 64) * jmax + nedge[row]++) * 64 + row % 64) * 3;
            q[0] = (double)((idx << 3) | (kind << 1) | own0); q[1] = rv[5 * e]; q[2] = rv[5 * e + 1];
        }
        const int32_t* pi = p_idx + (size_t)i * c.np_max;
        const double* pv = p_val + (size_t)i * c.np_max * 18;
        double* prec = A.h_aprec.data() + (size_t)i * prec_per;
        for (int e = 0; e < np; ++e) {
            const int cv = cls[pi[e]], row = cv >= 0 ? cv : nc + (-1 - cv);
            double* q = prec + (((size_t)(row / 64) * jpmax + nprior[row]++) * 64 + row % 64) * 7;
            q[0] = 1.0;
            for (int k = 0; k < 3; ++k) { q[1 + k] = pv[18 * e + 9 + k]; q[4 + k] = pv[18 * e + 12 + k]; }
        }
    }
    A.arrow_nb_max = nb_max; A.arrow_jmax = jmax; A.arrow_jpmax = jpmax;
    for (int k = 0; k < 16; ++k) { A.arrow_jch[k] = jch[k]; A.arrow_jpch[k] = jpch[k]; }
    return true;
}
static hipError_t upload_arrow_aux(loc_window* w, int which, int64_t n, hipStream_t st) {
    loc_window::WinAux& A = w->aux[which];
    const locamd::WindowCaps& c = w->caps;
    const size_t B = (size_t)w->B, N = (size_t)n;
    hipError_t e;
    if (!A.d_ahdr) {
        if ((e = hipMalloc((void**)&A.d_ahdr, B * 8 * sizeof(int32_t))) != hipSuccess ||
            (e = hipMalloc((void**)&A.d_arslot, B * c.nv_max * sizeof(int32_t))) != hipSuccess) return e;
    }
    if (A.arec_cap < A.h_arec.size()) {
        if (A.d_arec) (void)hipFree(A.d_arec);
        A.d_arec = nullptr; A.arec_cap = 0;
        if ((e = hipMalloc((void**)&A.d_arec, A.h_arec.size() / N * B * sizeof(double))) != hipSuccess) return e;
        A.arec_cap = A.h_arec.size() / N * B;
    }
    if (A.aprec_cap < A.h_aprec.size()) {
        if (A.d_aprec) (void)hipFree(A.d_aprec);
        A.d_aprec = nullptr; A.aprec_cap = 0;
        if ((e = hipMalloc((void**)&A.d_aprec, A.h_aprec.size() / N * B * sizeof(double))) != hipSuccess) return e;
        A.aprec_cap = A.h_aprec.size() / N * B;
    }
    if (!w->d_arrow_ws || w->arrow_ws_nb < A.arrow_nb_max) {
        if (w->d_arrow_ws) (void)hipFree(w->d_arrow_ws);
        w->d_arrow_ws = nullptr;
        if ((e = hipMalloc((void**)&w->d_arrow_ws, B * locamd::window_arrow3_workspace_doubles(c, A.arrow_nb_max) * sizeof(double))) != hipSuccess) return e;
        w->arrow_ws_nb = A.arrow_nb_max;
    }
    if ((e = hipMemcpyAsync(A.d_ahdr, A.h_ahdr.data(), N * 8 * sizeof(int32_t), hipMemcpyHostToDevice, st)) != hipSuccess ||
        (e = hipMemcpyAsync(A.d_arslot, A.h_arslot.data(), N * c.nv_max * sizeof(int32_t), hipMemcpyHostToDevice, st)) != hipSuccess ||
        (e = hipMemcpyAsync(A.d_arec, A.h_arec.data(), A.h_arec.size() * sizeof(double), hipMemcpyHostToDevice, st)) != hipSuccess ||
        (e = hipMemcpyAsync(A.d_aprec, A.h_aprec.data(), A.h_aprec.size() * sizeof(double), hipMemcpyHostToDevice, st)) != hipSuccess) return e;
    return hipStreamSynchronize(st);   // (the host vectors may be rebuilt by the next call)
}

// FOREST windows of ONE shared topology (BASELINE config 5: the key-frame star of addPoseEdge, localization.cpp:254-290, replayed
// with different measurements in every instance): every instance has the same counts and index tables, and the pose-to-pose
// edges form a forest.  Builds the elimination schedule tree_lm_kernel walks: nodes in post-order (children before their parent,
// a node's children heavy subtree first so that the leaves of one parent are consecutive), per node its parent and its edges.
// Layout of the int table: node[nv] par[nv] r_off[nv+1] r_list[nr] p_off[nv+1] p_list[np] s_off[nv+1] s_list[ns] r_idx[2 nr] s_idx[4 ns].
static bool build_tree_sched(loc_window* w, int which, int64_t n, const int32_t* counts, const int32_t* r_idx, const int32_t* p_idx, const int32_t* s_idx) {
    loc_window::WinAux& A = w->aux[which];
    const locamd::WindowCaps& c = w->caps;
    const int nv = counts[0], nr = counts[1], np = counts[2], ns = counts[3];
    if (nv < 2 || nv > 64 || w->has_off1) return false;
    for (int64_t i = 1; i < n; ++i) {   // one topology
        if (std::memcmp(counts + i * 4, counts, 4 * sizeof(int32_t)) != 0) return false;
        if (nr && std::memcmp(r_idx + (size_t)i * c.nr_max * 2, r_idx, (size_t)nr * 2 * sizeof(int32_t)) != 0) return false;
        if (np && std::memcmp(p_idx + (size_t)i * c.np_max, p_idx, (size_t)np * sizeof(int32_t)) != 0) return false;
        if (ns && std::memcmp(s_idx + (size_t)i * c.ns_max * 4, s_idx, (size_t)ns * 4 * sizeof(int32_t)) != 0) return false;
    }
    // adjacency (pairs joined by at least one edge); a forest has no cycle: union-find on the distinct pairs
    std::vector<int> uf((size_t)nv);
    for (int v = 0; v < nv; ++v) uf[(size_t)v] = v;
    auto find = [&](int v) { while (uf[(size_t)v] != v) { uf[(size_t)v] = uf[(size_t)uf[(size_t)v]]; v = uf[(size_t)v]; } return v; };
    std::vector<std::vector<int>> adj((size_t)nv);
    auto join = [&](int a, int b) -> bool {
        for (int x : adj[(size_t)a]) if (x == b) return true;   // a second edge on the same pair
        const int ra = find(a), rb = find(b);
        if (ra == rb) return false;                             // a cycle
        uf[(size_t)ra] = rb;
        adj[(size_t)a].push_back(b); adj[(size_t)b].push_back(a);
        return true;
    };
    for (int e = 0; e < nr; ++e) if (r_idx[2 * e + 1] >= 0 && !join(r_idx[2 * e], r_idx[2 * e + 1])) return false;
    for (int e = 0; e < ns; ++e) if (!join(s_idx[4 * e], s_idx[4 * e + 1])) return false;
    // root of every component = its CENTRE (the middle of a longest path: two breadth-first searches), so that the elimination by
    // height takes half as many steps as from an end (config 5's chain of eight keys: 6 levels instead of 9); parents towards the
    // root; subtree sizes; post-order, heavy child first
    std::vector<int> parent((size_t)nv, -2), size((size_t)nv, 1), order, stack, depth((size_t)nv, 0);
    int nroots = 0, maxdepth = 0;
    std::vector<int> bfs;
    std::vector<int> seen((size_t)nv, 0), dist((size_t)nv, 0), from((size_t)nv, -1), centre_of;
    auto far_from = [&](int start) {   // the farthest node from `start` inside its component (dist / from filled)
        std::vector<int> q2{start};
        std::vector<int> mark((size_t)nv, 0);
        mark[(size_t)start] = 1; dist[(size_t)start] = 0; from[(size_t)start] = -1;
        int last = start;
        for (size_t h = 0; h < q2.size(); ++h) {
            const int v = q2[h];
            last = v;
            for (int x : adj[(size_t)v]) if (!mark[(size_t)x]) { mark[(size_t)x] = 1; dist[(size_t)x] = dist[(size_t)v] + 1; from[(size_t)x] = v; q2.push_back(x); }
        }
        for (int v : q2) seen[(size_t)v] = 1;
        return last;
    };
    for (int v0 = 0; v0 < nv; ++v0) {
        if (seen[(size_t)v0]) continue;
        const int a1 = far_from(v0);
        const int b1 = far_from(a1);        // a1 .. b1: a longest path of this tree
        int c1 = b1;
        for (int step = dist[(size_t)b1] / 2; step > 0; --step) c1 = from[(size_t)c1];
        centre_of.push_back(c1);
    }
    for (int root : centre_of) {
        if (parent[(size_t)root] != -2) continue;
        parent[(size_t)root] = -1; ++nroots;
        const size_t b0 = bfs.size();
        bfs.push_back(root);
        for (size_t h = b0; h < bfs.size(); ++h) {
            const int v = bfs[h];
            for (int x : adj[(size_t)v]) if (parent[(size_t)x] == -2) { parent[(size_t)x] = v; depth[(size_t)x] = depth[(size_t)v] + 1; if (depth[(size_t)x] > maxdepth) maxdepth = depth[(size_t)x]; bfs.push_back(x); }
        }
        for (size_t h = bfs.size(); h-- > b0 + 1;) size[(size_t)parent[(size_t)bfs[h]]] += size[(size_t)bfs[h]];
    }
    std::vector<std::vector<int>> kids((size_t)nv);
    for (int v = 0; v < nv; ++v) if (parent[(size_t)v] >= 0) kids[(size_t)parent[(size_t)v]].push_back(v);
    for (auto& k : kids) std::stable_sort(k.begin(), k.end(), [&](int a2, int b2) { return size[(size_t)a2] > size[(size_t)b2]; });
    // iterative post-order
    for (int root = 0; root < nv; ++root) {
        if (parent[(size_t)root] != -1) continue;
        std::vector<std::pair<int, size_t>> st;
        st.push_back({root, 0});
        while (!st.empty()) {
            auto& top = st.back();
            if (top.second < kids[(size_t)top.first].size()) { const int ch = kids[(size_t)top.first][top.second++]; st.push_back({ch, 0}); }
            else { order.push_back(top.first); st.pop_back(); }
        }
    }
    if ((int)order.size() != nv) return false;
    std::vector<int> pos((size_t)nv);
    for (int k = 0; k < nv; ++k) pos[(size_t)order[(size_t)k]] = k;
    // edges by node: a unary edge belongs to its pose; an edge between a node and its parent to the node (the child)
    std::vector<std::vector<int>> re((size_t)nv), pe((size_t)nv), se((size_t)nv);
    for (int e = 0; e < nr; ++e) {
        const int v0 = r_idx[2 * e], v1 = r_idx[2 * e + 1];
        if (v1 < 0) re[(size_t)pos[(size_t)v0]].push_back(e);
        else re[(size_t)pos[(size_t)(parent[(size_t)v0] == v1 ? v0 : v1)]].push_back(e);
    }
    for (int e = 0; e < np; ++e) pe[(size_t)pos[(size_t)p_idx[e]]].push_back(e);
    for (int e = 0; e < ns; ++e) {
        const int vi = s_idx[4 * e], vj = s_idx[4 * e + 1];
        se[(size_t)pos[(size_t)(parent[(size_t)vi] == vj ? vi : vj)]].push_back(e);
    }
    std::vector<int32_t>& t = A.h_tsched;
    t.clear();
    for (int k = 0; k < nv; ++k) t.push_back(order[(size_t)k]);
    for (int k = 0; k < nv; ++k) { const int p = parent[(size_t)order[(size_t)k]]; t.push_back(p < 0 ? -1 : pos[(size_t)p]); }
    auto lists = [&](const std::vector<std::vector<int>>& L) {
        int acc = 0;
        for (int k = 0; k < nv; ++k) { t.push_back(acc); acc += (int)L[(size_t)k].size(); }
        t.push_back(acc);
        for (int k = 0; k < nv; ++k) for (int e : L[(size_t)k]) t.push_back(e);
    };
    lists(re); lists(pe); lists(se);
    for (int i = 0; i < 2 * nr; ++i) t.push_back(r_idx[i]);
    for (int i = 0; i < 4 * ns; ++i) t.push_back(s_idx[i]);
    // by pose slot (tree_wave_kernel): parent, height, children, edges
    std::vector<int> height((size_t)nv, 0);
    int hmax = 0;
    for (int k = 0; k < nv; ++k) {   // (post-order: children before their parent)
        const int v = order[(size_t)k], p = parent[(size_t)v];
        if (p >= 0 && height[(size_t)p] < height[(size_t)v] + 1) height[(size_t)p] = height[(size_t)v] + 1;
        if (height[(size_t)v] > hmax) hmax = height[(size_t)v];
    }
    for (int v = 0; v < nv; ++v) t.push_back(parent[(size_t)v]);
    for (int v = 0; v < nv; ++v) t.push_back(height[(size_t)v]);
    auto by_slot = [&](const std::vector<std::vector<int>>& L, bool positions) {   // L indexed by slot, or by schedule position
        int acc = 0;
        for (int v = 0; v < nv; ++v) { t.push_back(acc); acc += (int)L[(size_t)(positions ? pos[(size_t)v] : v)].size(); }
        t.push_back(acc);
        for (int v = 0; v < nv; ++v) for (int e : L[(size_t)(positions ? pos[(size_t)v] : v)]) t.push_back(e);
    };
    // (tree_wave_kernel wants a node's LEAF children first: their sums are taken in one parallel pass right after the leaves' level)
    std::vector<std::vector<int>> kids_lf((size_t)nv);
    std::vector<int> nleafkids((size_t)nv, 0);
    for (int v = 0; v < nv; ++v) {
        for (int ch : kids[(size_t)v]) if (height[(size_t)ch] == 0) { kids_lf[(size_t)v].push_back(ch); ++nleafkids[(size_t)v]; }
        for (int ch : kids[(size_t)v]) if (height[(size_t)ch] != 0) kids_lf[(size_t)v].push_back(ch);
    }
    by_slot(kids_lf, false); by_slot(re, true); by_slot(pe, true); by_slot(se, true);
    for (int v = 0; v < nv; ++v) t.push_back(nleafkids[(size_t)v]);
    // the inner nodes, parents first (tree_wave_kernel sums over children with lane = entry, one inner node after the other)
    int nu = 0;
    for (int h = hmax; h >= 1; --h)
        for (int v = 0; v < nv; ++v) if (height[(size_t)v] == h) { t.push_back(v); ++nu; }
    A.tsched.nu = nu;
    {   // a pose's position in the children list just emitted (by_slot(kids_lf)): parents in slot order, a parent's leaf children first
        std::vector<int> kpos((size_t)nv, -1);
        int at = 0;
        for (int v = 0; v < nv; ++v) for (int ch : kids_lf[(size_t)v]) kpos[(size_t)ch] = at++;
        for (int v = 0; v < nv; ++v) if (kpos[(size_t)v] < 0) kpos[(size_t)v] = at++;   // roots
        for (int v = 0; v < nv; ++v) t.push_back(kpos[(size_t)v]);
    }
    A.tsched.nlev = hmax + 1;
    A.tsched.max_se3_per_node = 0;
    A.tsched.max_r_per_node = 0;
    for (int k = 0; k < nv; ++k) if ((int)se[(size_t)k].size() > A.tsched.max_se3_per_node) A.tsched.max_se3_per_node = (int)se[(size_t)k].size();
    for (int k = 0; k < nv; ++k) if ((int)re[(size_t)k].size() > A.tsched.max_r_per_node) A.tsched.max_r_per_node = (int)re[(size_t)k].size();
    A.tsched.nv = nv; A.tsched.nr = nr; A.tsched.np = np; A.tsched.ns = ns; A.tsched.depth = maxdepth + 1; A.tsched.nroots = nroots;
    return true;
}
static hipError_t upload_tree_sched(loc_window* w, int which, hipStream_t st) {
    loc_window::WinAux& A = w->aux[which];
    hipError_t e;
    if (A.tsched_cap < A.h_tsched.size()) {
        if (A.d_tsched) (void)hipFree(A.d_tsched);
        A.d_tsched = nullptr; A.tsched_cap = 0;
        if ((e = hipMalloc((void**)&A.d_tsched, A.h_tsched.size() * sizeof(int32_t))) != hipSuccess) return e;
        A.tsched_cap = A.h_tsched.size();
    }
    if (!w->d_tree_ws && (e = hipMalloc((void**)&w->d_tree_ws, locamd::window_tree_workspace_doubles(w->caps, w->B) * sizeof(double))) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(A.d_tsched, A.h_tsched.data(), A.h_tsched.size() * sizeof(int32_t), hipMemcpyHostToDevice, st)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    locamd::TreeSched& ts = A.tsched;
    const int nv = ts.nv, nr = ts.nr, np = ts.np, ns = ts.ns;
    const int32_t* p = A.d_tsched;
    ts.node = p; p += nv; ts.par = p; p += nv;
    ts.r_off = p; p += nv + 1; ts.r_list = p; p += nr;
    ts.p_off = p; p += nv + 1; ts.p_list = p; p += np;
    ts.s_off = p; p += nv + 1; ts.s_list = p; p += ns;
    ts.r_idx = p; p += 2 * nr; ts.s_idx = p; p += 4 * ns;
    ts.w_par = p; p += nv; ts.w_height = p; p += nv;
    ts.w_koff = p; p += nv + 1; ts.w_klist = p; p += nv - ts.nroots;
    ts.w_roff = p; p += nv + 1; ts.w_rlist = p; p += nr;
    ts.w_poff = p; p += nv + 1; ts.w_plist = p; p += np;
    ts.w_soff = p; p += nv + 1; ts.w_slist = p; p += ns;
    ts.w_kleaf = p; p += nv;
    ts.w_ulist = p; p += ts.nu;
    ts.w_kpos = p;
    return hipSuccess;
}

static long long tree_min_batch(const loc_window* w) {
    const long long mn = w->chain_min >= 0 ? w->chain_min : chain_min_batch(w);
    if (mn <= 0) return 1ll << 62;     // (threshold 0 = "never a batch kernel")
    return mn < 256 ? mn : 256;
}

// 64-bit hash of a batch's STRUCTURE: n, the counts and the used entries of the index tables (never the measurements)
static unsigned long long hash_structure(const loc_window* w, int64_t n, const int32_t* counts, const int32_t* r_idx, const int32_t* p_idx, const int32_t* s_idx) {
    const locamd::WindowCaps& c = w->caps;
    unsigned long long part[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // one hash per thread's instance range, combined in range order
    parallel_chunks(n, [&](int64_t lo, int64_t hi, int t) {
        unsigned long long h = 0x9e3779b97f4a7c15ull + (unsigned long long)t;
        auto mix = [&h](const int32_t* p, size_t cnt) {
            size_t i = 0;
            for (; i + 2 <= cnt; i += 2) {
                unsigned long long v;
                std::memcpy(&v, p + i, 8);
                h = (h ^ v) * 0xff51afd7ed558ccdull;
                h ^= h >> 32;
            }
            if (i < cnt) { h = (h ^ (unsigned long long)(uint32_t)p[i]) * 0xc4ceb9fe1a85ec53ull; h ^= h >> 29; }
        };
        for (int64_t i = lo; i < hi; ++i) {
            const int32_t* cn = counts + i * 4;
            mix(cn, 4);
            if (cn[1]) mix(r_idx + (size_t)i * c.nr_max * 2, (size_t)cn[1] * 2);
            if (cn[2]) mix(p_idx + (size_t)i * c.np_max, (size_t)cn[2]);
            if (cn[3]) mix(s_idx + (size_t)i * c.ns_max * 4, (size_t)cn[3] * 4);
        }
        part[t & 7] = h;
    });
    unsigned long long h = 0x9e3779b97f4a7c15ull ^ (unsigned long long)n ^ (w->has_off1 ? 0x51ull << 56 : 0);
    for (int t = 0; t < 8; ++t) { h = (h ^ part[t]) * 0xff51afd7ed558ccdull; h ^= h >> 32; }
    return h;
}

// what the batch qualifies for BY ITS STRUCTURE: LOC_WINDOW_KERNEL_GENERAL, _CHAIN (block-tridiagonal, 6-DoF), _CHAIN3, _ARROW3 or _TREE
// (for _ARROW3 the edge lists are left in w->h_a*)
// which: the set of host-built tables ARROW3 / TREE fill (0: a loc_window_solve_host call, 1: the resident batch).
// Host-path calls (which == 0) keep the structural verdict of the previous batch in w->topo_cache: the same counts and index tables
// (one 64-bit hash; a collision — 2^-64 per call — would hand a batch to a kernel built for another structure) skip the tests below.
// What depends on the VALUES (translation_only: identity rotations, zero lever arms; arrow3's packed edge records) is looked at every time.
static int batch_topology(loc_window* w, int which, int64_t n, const int32_t* counts, const double* poses, const int32_t* r_idx, const double* r_val,
                          const int32_t* p_idx, const double* p_val, const int32_t* s_idx) {
    const locamd::WindowCaps& c = w->caps;
    loc_window::TopoCache& tc = w->topo_cache;
    const bool use_cache = which == 0 && w->opt.topology_cache;
    unsigned long long key = 0;
    bool hit = false;
    if (use_cache) {
        key = hash_structure(w, n, counts, r_idx, p_idx, s_idx);
        hit = tc.valid && tc.key == key && tc.n == n;
    }
    if (which == 0) w->t_cached = hit;
    bool chain = true;
    bool single_pairs = true;   // no EdgeSE3 anywhere and at most one range edge per pair of consecutive poses (wave6_lm_kernel's rank-1 couplings)
    bool se3_pairs = false;     // EdgeSE3 factors, at most one per pair of consecutive poses, and at most one range edge per pair (wave6_lm_kernel<JAC, true>)
    if (hit) { chain = tc.chain; single_pairs = tc.single_pairs; se3_pairs = tc.se3_pairs; }
    if (!hit) {
        std::atomic<bool> a_chain{true}, a_single{true}, a_single_r{true}, a_single_s{true}, a_any_s{false};
        parallel_chunks(n, [&](int64_t lo, int64_t hi, int) {
            bool chain_l = true, single_l = true, single_r = true, single_s = true, any_s = false;
            for (int64_t i = lo; i < hi && chain_l && a_chain.load(std::memory_order_relaxed); ++i) {
                const int32_t* cn = counts + i * 4;
                if (cn[3] != 0) { single_l = false; any_s = true; }
                int last = 0;
                for (int e = 0; e < cn[3]; ++e) {   // EdgeSE3 factors: between consecutive poses, ordered by their later pose (addTwistEdge)
                    const int32_t* ix = s_idx + ((size_t)i * c.ns_max + e) * 4;
                    const int key2 = ix[1] > ix[0] ? ix[1] : ix[0];
                    if (key2 < last || (ix[0] - ix[1] != 1 && ix[1] - ix[0] != 1)) { chain_l = false; break; }
                    if (key2 == last) single_s = false;   // (a second EdgeSE3 on the same pair; poses are numbered from 0, so `last` = 0 is no pair)
                    last = key2;
                }
                last = 0;
                int last_pair = -1;
                for (int e = 0; e < cn[1] && chain_l; ++e) {
                    const int32_t* ix = r_idx + ((size_t)i * c.nr_max + e) * 2;
                    const int key2 = ix[1] > ix[0] ? ix[1] : ix[0];
                    if (key2 < last) chain_l = false;
                    if (ix[1] >= 0) { if (key2 == last_pair) { single_l = false; single_r = false; } last_pair = key2; }
                    last = key2;
                    if (ix[1] >= 0 && ix[0] - ix[1] != 1 && ix[1] - ix[0] != 1) chain_l = false;
                }
                last = 0;
                for (int e = 0; e < cn[2] && chain_l; ++e) {
                    const int32_t v = p_idx[(size_t)i * c.np_max + e];
                    if (v < last) chain_l = false;
                    last = v;
                }
            }
            if (!chain_l) a_chain.store(false);
            if (!single_l) a_single.store(false);
            if (!single_r) a_single_r.store(false);
            if (!single_s) a_single_s.store(false);
            if (any_s) a_any_s.store(true);
        });
        chain = a_chain.load();
        single_pairs = a_single.load();   // (only meaningful for chain batches, where every range was scanned)
        se3_pairs = a_any_s.load() && a_single_r.load() && a_single_s.load();
    }
    if (use_cache && !hit) { tc.valid = true; tc.key = key; tc.n = n; tc.chain = chain; tc.single_pairs = single_pairs; tc.se3_pairs = se3_pairs; tc.tree_tried = false; tc.tree_ok = false; }
    if (chain) {
        if (translation_only(w, n, counts, poses, r_val, p_val)) return LOC_WINDOW_KERNEL_CHAIN3;
        if (single_pairs && c.nv_max <= 64 && locamd::window_wave6_lds_bytes(c) <= locamd::kWave6MaxLds) return LOC_WINDOW_KERNEL_WAVE6;
        // cfg/uwb_twist.yaml's window: a twist EdgeSE3 per consecutive pair next to the ranges — the wave-per-window kernel with full coupling
        // blocks.  (The same window was tried on tree_wave_kernel first — a chain is a forest, rooted at its centre it has 8 levels: 0.58 … 0.67 ms
        // per solve against the general kernel's 0.62 ms, tools/dev/probe_tree_chain.py: no speculative trials, one or two busy lanes per level.)
        if (se3_pairs && c.nv_max <= 63 && c.ns_max <= 64 && locamd::window_wave6_lds_bytes(c, true) <= locamd::kWave6MaxLds) return LOC_WINDOW_KERNEL_WAVE6S;   // (nv + 1 lanes: the middle pose twice)
        return LOC_WINDOW_KERNEL_CHAIN;
    }
    {
        // (option "arrow3": 0 = never, 1 = whenever the batch qualifies; default: windows of more than 64 poses — below that the
        //  wave-per-window kernel keeps everything in LDS and is the better choice)
        const bool want = w->opt.arrow3 >= 0 ? w->opt.arrow3 == 1 : c.nv_max > 64;
        if (want && translation_only(w, n, counts, poses, r_val, p_val) && build_arrow_aux(w, which, n, counts, r_idx, r_val, p_idx, p_val)) return LOC_WINDOW_KERNEL_ARROW3;
    }
    {
        // (option "tree" = 0: never.  One wave per window, so any batch gains; the host-side comparison of the index tables is only worth
        //  it from a few hundred windows on — or from the chain threshold when that was lowered, as the tests do)
        if (w->opt.tree != 0 && n >= tree_min_batch(w)) {
            if (hit && tc.tree_tried) {
                if (tc.tree_ok) return LOC_WINDOW_KERNEL_TREE;   // (aux[0]'s schedule is still the one built for this structure)
            } else {
                const bool ok = build_tree_sched(w, which, n, counts, r_idx, p_idx, s_idx);
                if (use_cache) { tc.tree_tried = true; tc.tree_ok = ok; }
                if (ok) return LOC_WINDOW_KERNEL_TREE;
            }
        }
    }
    return LOC_WINDOW_KERNEL_GENERAL;
}
// the kernel a batch of n windows with that structure takes NOW (threshold, ordering override, the handle's options)
static int pick_kernel(const loc_window* w, int64_t n, int topology) {
    const long long mn = w->chain_min >= 0 ? w->chain_min : chain_min_batch(w);
    const bool default_rule = w->chain_min < 0 && !w->opt.env_chain_min_set;
    if (w->has_off1) return LOC_WINDOW_KERNEL_GENERAL;   // (lever arms on endpoint 1: only the general kernel evaluates them)
    if (mn <= 0 || w->natural_order) return LOC_WINDOW_KERNEL_GENERAL;   // threshold 0 = "never anything but the general kernel" (every structure)
    if (topology == LOC_WINDOW_KERNEL_WAVE6) {
        // a 6-DoF chain batch that also qualifies for wave6_lm_kernel (one wave per window, rank-1 couplings).  Measured on twelve-pose
        // cfg/uwb_imu.yaml windows: 1.15e7 windows/s at 4 096, 16 384 and 65 536 windows against chain_lm_kernel's 1.1e6 / 4.3e6 / 6.9e6 —
        // so by default it takes every batch; an explicit threshold hands batches from that size on to the lane-per-window kernel.
        // option "wave6" = 0: as before (the general kernel below the threshold, chain_lm_kernel from it on), for A/B runs.
        const bool off = !w->opt.wave6;
        if (n >= mn && (off || !default_rule)) return LOC_WINDOW_KERNEL_CHAIN;
        return off ? LOC_WINDOW_KERNEL_GENERAL : LOC_WINDOW_KERNEL_WAVE6;
    }
    if (topology == LOC_WINDOW_KERNEL_WAVE6S) {   // the same rule for chains with EdgeSE3 factors
        const bool off = !w->opt.wave6;
        if (n >= mn && (off || !default_rule)) return LOC_WINDOW_KERNEL_CHAIN;
        return off ? LOC_WINDOW_KERNEL_GENERAL : LOC_WINDOW_KERNEL_WAVE6S;
    }
    if (topology == LOC_WINDOW_KERNEL_ARROW3) return LOC_WINDOW_KERNEL_ARROW3;   // (one workgroup per window: any batch size)
    if (topology == LOC_WINDOW_KERNEL_TREE) return n < tree_min_batch(w) ? LOC_WINDOW_KERNEL_GENERAL : LOC_WINDOW_KERNEL_TREE;
    if (topology == LOC_WINDOW_KERNEL_CHAIN3 && w->caps.nv_max <= 64 && locamd::window_wave3_lds_bytes(w->caps) <= 64 * 1024) {
        // translation-only chains of <= 64 poses (the node's single window first of all): one wave per window with 3x3 blocks, rank-1
        // couplings and speculative LM trials.  Measured on ten-pose windows: 0.056 ms for one window, 3.6e7 windows/s (numeric) /
        // 4.0e7 (analytic) from ~8 000 windows on — level with chain3_lm_kernel at 65 536 windows, ahead of it everywhere else — so by
        // default it takes every batch; an explicit threshold (loc_window_set_chain_threshold / LOCAMD_CHAIN_MIN_BATCH) hands batches
        // from that size on to the lane-per-window kernel.  Options "wave3" / "chain3" = 0: no such kernel (A/B runs, tests).
        if ((default_rule || n < mn) && w->opt.wave3 && w->opt.chain3) return LOC_WINDOW_KERNEL_WAVE3;
    }
    if (topology == LOC_WINDOW_KERNEL_CHAIN3 && default_rule && n >= 4096 && n < mn) {
        // the translation-only kernel is worth it from ~4 096 windows on (it takes ~1 ms for any batch up to 16 384, the wave-per-window
        // kernel 4.3e6 windows/s): e.g. one GPU's 8 192-window share of a 65 536-window job split over eight
        if (w->opt.chain3) return LOC_WINDOW_KERNEL_CHAIN3;
    }
    if (topology == LOC_WINDOW_KERNEL_GENERAL || n < mn) return LOC_WINDOW_KERNEL_GENERAL;
    if (topology == LOC_WINDOW_KERNEL_CHAIN3 && !w->opt.chain3) return LOC_WINDOW_KERNEL_CHAIN;   // the 6-DoF kernel on a translation-only batch (A/B runs, tests)
    return topology;
}
static hipError_t launch_any(loc_window* w, int which, const locamd::WindowArgs& a, hipStream_t st, int kind) {
    loc_window::WinAux& A = w->aux[which];
    w->last_kind = kind;
    if (kind == LOC_WINDOW_KERNEL_ARROW3) {
        locamd::ArrowAux x;
        x.hdr = A.d_ahdr; x.rslot = A.d_arslot; x.rec = A.d_arec; x.prec = A.d_aprec;
        x.ws = w->d_arrow_ws; x.nb_max = A.arrow_nb_max; x.jmax = A.arrow_jmax; x.jpmax = A.arrow_jpmax; x.nchunk = (w->caps.nv_max + 63) / 64;
        for (int k = 0; k < 16; ++k) { x.jch[k] = A.arrow_jch[k]; x.jpch[k] = A.arrow_jpch[k]; }
        return locamd::launch_window_arrow3(a, x, st);
    }
    if (kind == LOC_WINDOW_KERNEL_TREE) {   // (option "tree" = 2: the one-lane-per-window variant, for A/B runs)
        if (w->opt.tree == 2 || A.tsched.max_se3_per_node > 1) {   // (tree_wave_kernel: one EdgeSE3 per node)
            w->last_kind = LOC_WINDOW_KERNEL_TREE_LANE;
            return locamd::launch_window_tree(a, A.tsched, w->d_tree_ws, st);
        }
        return locamd::launch_window_tree_wave(a, A.tsched, st);
    }
    if (kind == LOC_WINDOW_KERNEL_CHAIN3) {
        if (!w->d_chain3_ws) {
            hipError_t e = hipMalloc((void**)&w->d_chain3_ws, locamd::window_chain3_workspace_doubles(w->caps, w->B) * sizeof(double));
            if (e != hipSuccess) return e;
        }
        return locamd::launch_window_chain3(a, w->d_chain3_ws, st);
    }
    if (kind == LOC_WINDOW_KERNEL_WAVE3) return locamd::launch_window_wave3(a, st);
    if (kind == LOC_WINDOW_KERNEL_WAVE6) return locamd::launch_window_wave6(a, false, st);
    if (kind == LOC_WINDOW_KERNEL_WAVE6S) return locamd::launch_window_wave6(a, true, st);
    if (kind == LOC_WINDOW_KERNEL_GENERAL) return locamd::launch_window(a, st);
    if (!w->d_chain_ws) {
        hipError_t e = hipMalloc((void**)&w->d_chain_ws, locamd::window_chain_workspace_doubles(w->caps, w->B) * sizeof(double));
        if (e != hipSuccess) return e;
    }
    return locamd::launch_window_chain(a, w->d_chain_ws, st);
}

int loc_window_last_kernel_kind(const loc_window* w, int32_t* kind) {
    if (!w || !kind) return locamd_fail(LOC_ERR_INVALID, "null");
    *kind = w->last_kind;
    return LOC_OK;
}

int loc_window_set_chain_threshold(loc_window* w, int64_t min_batch) {
    if (!w) return locamd_fail(LOC_ERR_INVALID, "null");
    w->chain_min = min_batch;
    return LOC_OK;
}

int loc_window_set_option(loc_window* w, const char* name, int64_t value) {
    if (!w || !name) return locamd_fail(LOC_ERR_INVALID, "set_option");
    const std::string k(name);
    loc_window::Opts& o = w->opt;
    auto flag = [&](bool& f) { if (value != 0 && value != 1) return locamd_fail(LOC_ERR_INVALID, "set_option: 0 or 1"); f = value == 1; return (int)LOC_OK; };
    if (k == "chain_min_batch") { w->chain_min = value; return LOC_OK; }
    if (k == "arrow3") { if (value < -1 || value > 1) return locamd_fail(LOC_ERR_INVALID, "set_option arrow3: -1, 0 or 1"); o.arrow3 = (int)value; w->topo_cache.valid = false; return LOC_OK; }
    if (k == "tree") { if (value != -1 && value != 0 && value != 2) return locamd_fail(LOC_ERR_INVALID, "set_option tree: -1, 0 or 2"); o.tree = (int)value; w->topo_cache.valid = false; return LOC_OK; }
    if (k == "wave3") return flag(o.wave3);
    if (k == "wave6") return flag(o.wave6);
    if (k == "chain3") return flag(o.chain3);
    if (k == "zero_copy") return flag(o.zero_copy);
    if (k == "kernel_events") return flag(o.kernel_events);
    if (k == "topology_cache") { w->topo_cache.valid = false; return flag(o.topology_cache); }
    return locamd_fail(LOC_ERR_INVALID, "set_option: unknown option name");
}

int loc_window_last_host_timing(const loc_window* w, double* out) {
    if (!w || !out) return locamd_fail(LOC_ERR_INVALID, "null");
    out[0] = w->t_validate_ms; out[1] = w->t_topology_ms; out[2] = w->t_run_ms; out[3] = w->t_cached ? 1.0 : 0.0;
    return LOC_OK;
}

int loc_window_set_endpoint1_offsets(loc_window* w, int64_t n, const double* off1) {
    if (!w || (off1 && (n <= 0 || n > w->B))) return locamd_fail(LOC_ERR_INVALID, "set_endpoint1_offsets");
    if (!off1) { w->has_off1 = false; return LOC_OK; }   // (has_off1 is part of the structure hash: no cache entry survives a change of it)
    if (w->caps.nr_max <= 0) return locamd_fail(LOC_ERR_INVALID, "set_endpoint1_offsets: no range edges in this solver");
    LOC_HIP(hipSetDevice(w->device));
    if (int rc = wait_resident(w)) return rc;
    LOC_HIP(hipStreamSynchronize(w->stream));
    const size_t row = (size_t)w->caps.nr_max * 3 * sizeof(double);
    if (!w->d_roff1) LOC_HIP(hipMalloc((void**)&w->d_roff1, (size_t)w->B * row));
    LOC_HIP(hipMemcpy(w->d_roff1, off1, (size_t)n * row, hipMemcpyHostToDevice));
    // rows [n, B): no lever arm (a call with fewer instances than an earlier one must not leave that one's behind)
    if (n < w->B) LOC_HIP(hipMemset((char*)w->d_roff1 + (size_t)n * row, 0, (size_t)(w->B - n) * row));
    w->has_off1 = true;
    return LOC_OK;
}

int loc_window_set_jacobian(loc_window* w, int32_t jacobian) {
    if (!w || (jacobian != LOC_JAC_ANALYTIC && jacobian != LOC_JAC_NUMERIC_G2O)) return locamd_fail(LOC_ERR_INVALID, "jacobian mode");
    w->jacobian = jacobian;
    return LOC_OK;
}
int loc_window_set_ordering(loc_window* w, int32_t natural) {
    if (!w) return locamd_fail(LOC_ERR_INVALID, "null");
    w->natural_order = natural != 0;
    return LOC_OK;
}

int loc_window_solve_host(loc_window* w, int64_t n, const int32_t* counts, double* poses, const int32_t* r_idx,
                          const double* r_val, const int32_t* p_idx, const double* p_val, const int32_t* s_idx,
                          const double* s_val, double* result) {
    if (!result) return locamd_fail(LOC_ERR_INVALID, "window solve arguments");
    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); };
    const auto t_begin = clk::now();
    {
        const int rc = validate_instances(w, n, counts, poses, r_idx, r_val, p_idx, p_val, s_idx, s_val);
        if (rc != LOC_OK) return rc;
    }
    w->t_validate_ms = ms_since(t_begin);
    const locamd::WindowCaps& c = w->caps;
    LOC_HIP(hipSetDevice(w->device));
    // a resident launch (possibly on a caller's stream) may still be using the workspaces and tables this call shares with it
    if (int rc = wait_resident(w)) return rc;
    const size_t N = (size_t)n;
    hipStream_t st = w->stream;
    {
        // Single-block path: [poses | result | counts | r_val | p_val | s_val | r_idx | p_idx | s_idx], 16-byte aligned
        // pieces, sized for this call's n.  One H2D copy, one launch, one D2H copy of [poses | result]: a node's solve is
        // then ~0.2 ms of host/PCIe overhead around the kernel instead of a dozen small pageable copies.
        auto al = [](size_t b) { return (b + 15) & ~(size_t)15; };
        const size_t sz[9] = {al(N * c.nv_max * 12 * sizeof(double)), al(N * 8 * sizeof(double)), al(N * 4 * sizeof(int32_t)),
                              al(N * c.nr_max * 5 * sizeof(double)), al(N * c.np_max * 18 * sizeof(double)), al(N * c.ns_max * 48 * sizeof(double)),
                              al(N * c.nr_max * 2 * sizeof(int32_t)), al(N * c.np_max * sizeof(int32_t)), al(N * c.ns_max * 4 * sizeof(int32_t))};
        size_t off[10]; off[0] = 0;
        for (int i = 0; i < 9; ++i) off[i + 1] = off[i] + sz[i];
        if (off[9] <= kStageBytes) {
            if (!w->h_stage) LOC_HIP(hipHostMalloc((void**)&w->h_stage, kStageBytes, hipHostMallocDefault));
            if (!w->d_stage) LOC_HIP(hipMalloc((void**)&w->d_stage, kStageBytes));
            char* h = w->h_stage; char* d = w->d_stage;
            std::memcpy(h + off[0], poses, N * c.nv_max * 12 * sizeof(double));
            std::memcpy(h + off[2], counts, N * 4 * sizeof(int32_t));
            if (c.nr_max) { std::memcpy(h + off[3], r_val, N * c.nr_max * 5 * sizeof(double)); std::memcpy(h + off[6], r_idx, N * c.nr_max * 2 * sizeof(int32_t)); }
            if (c.np_max) { std::memcpy(h + off[4], p_val, N * c.np_max * 18 * sizeof(double)); std::memcpy(h + off[7], p_idx, N * c.np_max * sizeof(int32_t)); }
            if (c.ns_max) { std::memcpy(h + off[5], s_val, N * c.ns_max * 48 * sizeof(double)); std::memcpy(h + off[8], s_idx, N * c.ns_max * 4 * sizeof(int32_t)); }
            const auto t_topo = clk::now();
            const int kind = pick_kernel(w, n, batch_topology(w, 0, n, counts, poses, r_idx, r_val, p_idx, p_val, s_idx));
            w->t_topology_ms = ms_since(t_topo);
            const auto t_run = clk::now();
            // A handful of small windows on wave3_lm_kernel (the node's own solve): the kernel reads its few KB of input once and
            // writes 1 KB of results — it does so straight from / to the page-locked staging block (host-coherent memory, mapped
            // into the device's address space), which saves the two DMA operations around a ~75 us kernel.
            const size_t anchor_bytes = (size_t)w->n_anchors * 3 * sizeof(double);
            // (tree_wave_kernel likewise reads its inputs once, in its prologue — unless a pose has priors or more than two range edges, which it
            //  fetches from memory on every sweep: such windows are copied to the device first)
            const bool tree_once = kind == LOC_WINDOW_KERNEL_TREE && w->aux[0].tsched.np == 0 && w->aux[0].tsched.max_r_per_node <= 2 &&
                                   w->aux[0].tsched.max_se3_per_node <= 1 && w->opt.tree != 2;
            const bool zero_copy = (kind == LOC_WINDOW_KERNEL_WAVE3 || kind == LOC_WINDOW_KERNEL_WAVE6 || kind == LOC_WINDOW_KERNEL_WAVE6S || tree_once) && n <= 4 && w->B <= 4 && w->h_anchors.size() == (size_t)w->n_anchors * 3 &&
                                   off[9] + anchor_bytes <= kStageBytes && w->opt.zero_copy;
            if (zero_copy) {
                d = h;
                if (anchor_bytes) std::memcpy(h + off[9], w->h_anchors.data(), anchor_bytes);
            } else {
                const int rc = flush_anchors(w);
                if (rc != LOC_OK) return rc;
                LOC_HIP(hipMemcpyAsync(d, h, off[9], hipMemcpyHostToDevice, st));
            }
            locamd::WindowArgs a;
            a.poses = (double*)(d + off[0]); a.poses_in = a.poses; a.jacobian = w->jacobian; a.natural_order = w->natural_order; a.result = (double*)(d + off[1]); a.counts = (const int32_t*)(d + off[2]);
            a.r_val = (const double*)(d + off[3]); a.p_val = (const double*)(d + off[4]); a.s_val = (const double*)(d + off[5]);
            a.r_off1 = w->has_off1 ? w->d_roff1 : nullptr;
            a.r_idx = (const int32_t*)(d + off[6]); a.p_idx = (const int32_t*)(d + off[7]); a.s_idx = (const int32_t*)(d + off[8]);
            a.anchors = zero_copy ? (const double*)(h + off[9]) : w->d_anchors; a.workspace = w->d_workspace;
            a.n_anchors = w->n_anchors; a.B = (int)n; a.iterations = w->iterations; a.caps = c;
            // (the two event records around a ~50 us kernel are not free; option "kernel_events" = 0 drops them for the zero-copy solve)
            const bool events = w->opt.kernel_events || !zero_copy;
            if (events) LOC_HIP(hipEventRecord(w->ev0, st));
            if (kind == LOC_WINDOW_KERNEL_ARROW3) LOC_HIP(upload_arrow_aux(w, 0, n, st));
            if (kind == LOC_WINDOW_KERNEL_TREE) LOC_HIP(upload_tree_sched(w, 0, st));
            const auto t_launch = clk::now();
            hipError_t e = launch_any(w, 0, a, st, kind);
            if (e != hipSuccess) return locamd_fail_hip(e, "launch_window");
            if (events) LOC_HIP(hipEventRecord(w->ev1, st));
            if (!zero_copy) LOC_HIP(hipMemcpyAsync(h, d, off[2], hipMemcpyDeviceToHost, st));  // [poses | result]
            LOC_HIP(hipStreamSynchronize(st));
            float ms = (float)ms_since(t_launch);
            std::memcpy(poses, h + off[0], N * c.nv_max * 12 * sizeof(double));
            std::memcpy(result, h + off[1], N * 8 * sizeof(double));
            if (events) LOC_HIP(hipEventElapsedTime(&ms, w->ev0, w->ev1));
            w->last_ms = ms;
            w->t_run_ms = ms_since(t_run);
            return LOC_OK;
        }
    }
    {
        const int rc = flush_anchors(w);
        if (rc != LOC_OK) return rc;
    }
    // The large path stages its batch in the device arrays a resident batch lives in: that batch is gone from here on
    // (loc_window_solve_resident / loc_window_download return LOC_ERR_INVALID until the next loc_window_upload).
    w->n_resident = 0; w->resident_solved = false; w->resident_topology = LOC_WINDOW_KERNEL_GENERAL; w->resident_min_anchors = 0;
    const auto t_topo = clk::now();
    const int kind = pick_kernel(w, n, batch_topology(w, 0, n, counts, poses, r_idx, r_val, p_idx, p_val, s_idx));
    w->t_topology_ms = ms_since(t_topo);
    const auto t_run = clk::now();
    LOC_HIP(hipMemcpyAsync(w->d_counts, counts, N * 4 * sizeof(int32_t), hipMemcpyHostToDevice, st));
    LOC_HIP(hipMemcpyAsync(w->d_poses, poses, N * c.nv_max * 12 * sizeof(double), hipMemcpyHostToDevice, st));
    if (c.nr_max) {
        LOC_HIP(hipMemcpyAsync(w->d_ridx, r_idx, N * c.nr_max * 2 * sizeof(int32_t), hipMemcpyHostToDevice, st));
        LOC_HIP(hipMemcpyAsync(w->d_rval, r_val, N * c.nr_max * 5 * sizeof(double), hipMemcpyHostToDevice, st));
    }
    if (c.np_max) {
        LOC_HIP(hipMemcpyAsync(w->d_pidx, p_idx, N * c.np_max * sizeof(int32_t), hipMemcpyHostToDevice, st));
        LOC_HIP(hipMemcpyAsync(w->d_pval, p_val, N * c.np_max * 18 * sizeof(double), hipMemcpyHostToDevice, st));
    }
    if (c.ns_max) {
        LOC_HIP(hipMemcpyAsync(w->d_sidx, s_idx, N * c.ns_max * 4 * sizeof(int32_t), hipMemcpyHostToDevice, st));
        LOC_HIP(hipMemcpyAsync(w->d_sval, s_val, N * c.ns_max * 48 * sizeof(double), hipMemcpyHostToDevice, st));
    }
    locamd::WindowArgs a;
    a.counts = w->d_counts; a.poses = w->d_poses; a.poses_in = a.poses; a.jacobian = w->jacobian; a.natural_order = w->natural_order; a.r_idx = w->d_ridx; a.r_val = w->d_rval; a.p_idx = w->d_pidx;
    a.r_off1 = w->has_off1 ? w->d_roff1 : nullptr;
    a.p_val = w->d_pval; a.s_idx = w->d_sidx; a.s_val = w->d_sval; a.anchors = w->d_anchors; a.result = w->d_result;
    a.workspace = w->d_workspace;
    a.n_anchors = w->n_anchors; a.B = (int)n; a.iterations = w->iterations; a.caps = c;
    if (kind == LOC_WINDOW_KERNEL_ARROW3) LOC_HIP(upload_arrow_aux(w, 0, n, st));
    if (kind == LOC_WINDOW_KERNEL_TREE) LOC_HIP(upload_tree_sched(w, 0, st));
    LOC_HIP(hipEventRecord(w->ev0, st));
    hipError_t e = launch_any(w, 0, a, st, kind);
    if (e != hipSuccess) return locamd_fail_hip(e, "launch_window");
    LOC_HIP(hipEventRecord(w->ev1, st));
    LOC_HIP(hipMemcpyAsync(poses, w->d_poses, N * c.nv_max * 12 * sizeof(double), hipMemcpyDeviceToHost, st));
    LOC_HIP(hipMemcpyAsync(result, w->d_result, N * 8 * sizeof(double), hipMemcpyDeviceToHost, st));
    LOC_HIP(hipStreamSynchronize(st));
    float ms = 0;
    LOC_HIP(hipEventElapsedTime(&ms, w->ev0, w->ev1));
    w->last_ms = ms;
    w->t_run_ms = ms_since(t_run);
    return LOC_OK;
}

// ---- device-resident operation --------------------------------------------------------------------------------------
int loc_window_upload(loc_window* w, int64_t n, const int32_t* counts, const double* poses, const int32_t* r_idx,
                      const double* r_val, const int32_t* p_idx, const double* p_val, const int32_t* s_idx,
                      const double* s_val) {
    {
        const int rc = validate_instances(w, n, counts, poses, r_idx, r_val, p_idx, p_val, s_idx, s_val);
        if (rc != LOC_OK) return rc;
    }
    const locamd::WindowCaps& c = w->caps;
    LOC_HIP(hipSetDevice(w->device));
    const size_t N = (size_t)n;
    // a resident launch of the previous batch may still be running on the handle's (or the caller's) stream: it reads what the
    // copies below overwrite
    LOC_HIP(hipStreamSynchronize(w->stream));
    if (int rc = wait_resident(w)) return rc;
    // (a failing step below must not leave a half-described resident batch behind: nothing is resident until everything is)
    w->n_resident = 0; w->resident_solved = false; w->resident_topology = LOC_WINDOW_KERNEL_GENERAL; w->resident_min_anchors = 0;
    if (!w->d_poses_in) LOC_HIP(hipMalloc((void**)&w->d_poses_in, (size_t)w->B * c.nv_max * 12 * sizeof(double)));
    LOC_HIP(hipMemcpy(w->d_counts, counts, N * 4 * sizeof(int32_t), hipMemcpyHostToDevice));
    LOC_HIP(hipMemcpy(w->d_poses_in, poses, N * c.nv_max * 12 * sizeof(double), hipMemcpyHostToDevice));
    if (c.nr_max) {
        LOC_HIP(hipMemcpy(w->d_ridx, r_idx, N * c.nr_max * 2 * sizeof(int32_t), hipMemcpyHostToDevice));
        LOC_HIP(hipMemcpy(w->d_rval, r_val, N * c.nr_max * 5 * sizeof(double), hipMemcpyHostToDevice));
    }
    if (c.np_max) {
        LOC_HIP(hipMemcpy(w->d_pidx, p_idx, N * c.np_max * sizeof(int32_t), hipMemcpyHostToDevice));
        LOC_HIP(hipMemcpy(w->d_pval, p_val, N * c.np_max * 18 * sizeof(double), hipMemcpyHostToDevice));
    }
    if (c.ns_max) {
        LOC_HIP(hipMemcpy(w->d_sidx, s_idx, N * c.ns_max * 4 * sizeof(int32_t), hipMemcpyHostToDevice));
        LOC_HIP(hipMemcpy(w->d_sval, s_val, N * c.ns_max * 48 * sizeof(double), hipMemcpyHostToDevice));
    }
    const int topology = batch_topology(w, 1, n, counts, poses, r_idx, r_val, p_idx, p_val, s_idx);
    if (topology == LOC_WINDOW_KERNEL_ARROW3) LOC_HIP(upload_arrow_aux(w, 1, n, w->stream));
    if (topology == LOC_WINDOW_KERNEL_TREE) LOC_HIP(upload_tree_sched(w, 1, w->stream));
    int max_anchor = 0;   // anchors referenced: v1 = -1 - anchor
    for (int64_t i = 0; i < n; ++i)
        for (int e = 0; e < counts[i * 4 + 1]; ++e) {
            const int32_t v1 = r_idx[((size_t)i * c.nr_max + e) * 2 + 1];
            if (v1 < 0 && -v1 > max_anchor) max_anchor = -v1;
        }
    w->resident_min_anchors = max_anchor;
    w->resident_topology = topology;
    w->resident_solved = false;
    w->n_resident = n;
    return LOC_OK;
}

int loc_window_solve_resident(loc_window* w, void* hip_stream) {
    if (!w || w->n_resident <= 0) return locamd_fail(LOC_ERR_INVALID, "nothing uploaded");
    LOC_HIP(hipSetDevice(w->device));
    {
        const int rc = flush_anchors(w);
        if (rc != LOC_OK) return rc;
    }
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : w->stream;
    locamd::WindowArgs a;
    a.counts = w->d_counts; a.poses_in = w->d_poses_in; a.poses = w->d_poses; a.r_idx = w->d_ridx; a.r_val = w->d_rval; a.p_idx = w->d_pidx;
    a.r_off1 = w->has_off1 ? w->d_roff1 : nullptr;
    a.p_val = w->d_pval; a.s_idx = w->d_sidx; a.s_val = w->d_sval; a.anchors = w->d_anchors; a.result = w->d_result;
    a.workspace = w->d_workspace;
    a.n_anchors = w->n_anchors; a.B = (int)w->n_resident; a.iterations = w->iterations; a.jacobian = w->jacobian;
    a.natural_order = w->natural_order; a.caps = w->caps;
    const bool timed = w->timing && (size_t)(w->ev_used + 2) <= w->ev.size();
    if (timed) LOC_HIP(hipEventRecord(w->ev[w->ev_used], st));
    // (the batch-size threshold and the ordering override are looked at per solve: loc_window_set_chain_threshold /
    //  loc_window_set_ordering after the upload take effect)
    hipError_t e = launch_any(w, 1, a, st, pick_kernel(w, w->n_resident, w->resident_topology));
    if (e != hipSuccess) return locamd_fail_hip(e, "launch_window");
    if (timed) { LOC_HIP(hipEventRecord(w->ev[w->ev_used + 1], st)); w->ev_used += 2; }
    LOC_HIP(hipEventRecord(w->resident_done, st));
    w->resident_inflight = true;
    w->resident_solved = true;
    return LOC_OK;
}

int loc_window_download(loc_window* w, double* poses, double* result) {
    if (!w || w->n_resident <= 0) return locamd_fail(LOC_ERR_INVALID, "nothing uploaded");
    if (!w->resident_solved) return locamd_fail(LOC_ERR_INVALID, "loc_window_download: no resident solve has run since the upload");
    LOC_HIP(hipSetDevice(w->device));
    if (int rc = wait_resident(w)) return rc;
    const size_t N = (size_t)w->n_resident;
    if (poses) LOC_HIP(hipMemcpy(poses, w->d_poses, N * w->caps.nv_max * 12 * sizeof(double), hipMemcpyDeviceToHost));
    if (result) LOC_HIP(hipMemcpy(result, w->d_result, N * 8 * sizeof(double), hipMemcpyDeviceToHost));
    return LOC_OK;
}

void* loc_window_poses_device(loc_window* w) { return w ? (void*)w->d_poses : nullptr; }
void* loc_window_result_device(loc_window* w) { return w ? (void*)w->d_result : nullptr; }

int loc_window_timing_begin(loc_window* w, int32_t max_launches) {
    if (!w || max_launches <= 0) return locamd_fail(LOC_ERR_INVALID, "timing_begin");
    LOC_HIP(hipSetDevice(w->device));
    while ((int)w->ev.size() < 2 * max_launches) {
        hipEvent_t ev;
        LOC_HIP(hipEventCreate(&ev));
        w->ev.push_back(ev);
    }
    w->ev_used = 0;
    w->timing = true;
    return LOC_OK;
}
int loc_window_timing_end(loc_window* w, int32_t* n_launches, double* total_ms, double* avg_ms) {
    if (!w) return locamd_fail(LOC_ERR_INVALID, "timing_end");
    LOC_HIP(hipSetDevice(w->device));
    w->timing = false;
    double tot = 0;
    const int n = w->ev_used / 2;
    for (int i = 0; i < n; ++i) {
        LOC_HIP(hipEventSynchronize(w->ev[2 * i + 1]));
        float ms = 0;
        LOC_HIP(hipEventElapsedTime(&ms, w->ev[2 * i], w->ev[2 * i + 1]));
        tot += ms;
    }
    if (n_launches) *n_launches = n;
    if (total_ms) *total_ms = tot;
    if (avg_ms) *avg_ms = n ? tot / n : 0.0;
    w->ev_used = 0;
    return LOC_OK;
}

int loc_window_last_kernel_ms(loc_window* w, double* ms) {
    if (!w || !ms) return locamd_fail(LOC_ERR_INVALID, "null");
    *ms = w->last_ms;
    return LOC_OK;
}

}  // extern "C"
