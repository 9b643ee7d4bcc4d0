/*
 * localization_amd — C ABI of the MI355X-native range-localization solver.
 *
 * The reference (sair-lab/localization) has no plugin/FFI seam: class Localization owns a
 * g2o::SparseOptimizer by value (reference src/localization/localization.h:168) and calls it directly.
 * The boundary cut here is exactly the set of g2o calls Localization/Robot make; every
 * entry point below names the reference call site(s) it replaces.  Plain pointers and sizes only; no
 * exceptions cross the ABI; every function returns a loc_status (0 = OK, <0 = error).
 *
 * There is NO CPU fallback: creating a solver without a usable HIP device fails with LOC_ERR_NO_DEVICE.
 *
 * Threading: like the reference (one ros::spin() thread, localization_node.cpp:98) a handle is not
 * thread-safe; distinct handles are independent.  All device work of a handle is issued on the HIP stream
 * the caller passes (or the handle's own stream when NULL).
 */
#ifndef LOCALIZATION_AMD_H
#define LOCALIZATION_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LOC_ABI_VERSION 4 /* 2: jacobian mode for every solver, resident window solves, loc_node_add_rl_range;
                           * 3: numeric (g2o) Jacobians are the default everywhere, loc_shard_*, loc_window_last_kernel_kind;
                           * 4: loc_window_set_option / _last_host_timing, loc_node_flush_tail / _last_kernel_kind, loc_fusion_timing_*;
                           *    a large loc_window_solve_host drops the resident batch */

typedef enum loc_status {
    LOC_OK = 0,
    LOC_ERR_INVALID = -1,      /* bad argument / shape */
    LOC_ERR_NO_DEVICE = -2,    /* no HIP device: the product path refuses to run */
    LOC_ERR_HIP = -3,          /* a HIP runtime call failed (see loc_last_error) */
    LOC_ERR_UNKNOWN_NODE = -4, /* node id not in nodesId (reference: std::map::at throws, localization.cpp:306) */
    LOC_ERR_UNSUPPORTED = -5,  /* shape outside what the kernels are built for */
    LOC_ERR_SINGULAR = -6      /* non-invertible covariance (reference: MatrixXd::inverse, localization.cpp:277,601) */
} loc_status;

/* Range-edge Jacobian. The reference inherits g2o's numeric central difference (delta = 1e-9) because
 * EdgeSE3Range does not override linearizeOplus (types_edge_se3range.h:45-74).  ANALYTIC is the exact
 * derivative of the same residual (0 where the endpoints coincide, which is what the numeric form gives). */
enum { LOC_JAC_ANALYTIC = 0, LOC_JAC_NUMERIC_G2O = 1 };

const char* loc_last_error(void);  /* thread-local message of the last failing call */
int32_t loc_abi_version(void);
int32_t loc_device_count(void);    /* 0 when no HIP device is visible */

/* ================================================================================================
 * Multi-GPU shard descriptor (SURVEY.md §8(b) "device selection + stream; multi-GPU shard descriptor", §8(e)).
 * Tags / windows / hypotheses are independent least-squares problems (one Localization object each in the reference:
 * localization_node.cpp:46), so a batch of `total` instances is split into contiguous slices of ceil(total / world)
 * instances, one per rank = one process = one GPU; anchors and parameters are replicated; there is NO collective on
 * the solve path.  Host-only helpers (no device needed): a C++ caller computes its slice with the same rule the Python
 * harness and bench.py use.
 * ============================================================================================== */
typedef struct loc_shard {
    int32_t rank, world;   /* this slice belongs to rank `rank` of `world` */
    int32_t device;        /* HIP device index on the rank's node: rank % devices_per_node */
    int32_t reserved;
    int64_t lo, hi;        /* instances [lo, hi) of the batch; hi - lo may be 0 for trailing ranks */
} loc_shard;
/* [lo, hi) of rank `rank`; LOC_ERR_INVALID unless total >= 0 and 0 <= rank < world */
int loc_shard_bounds(int64_t total, int32_t rank, int32_t world, int64_t* lo, int64_t* hi);
/* fills out[0 .. world-1]; devices_per_node <= 0: loc_device_count() (LOC_ERR_NO_DEVICE when that is 0) */
int loc_shard_plan(int64_t total, int32_t world, int32_t devices_per_node, loc_shard* out);

/* ================================================================================================
 * Batched snapshot solver — BASELINE config 2 (8-anchor UWB, B independent tags, 3-DoF position).
 *
 * One "update" = for one tag, ingest one epoch of M anchor ranges and run what the reference runs per
 * solve: create_range_edge per range (information 1/err^2, RobustKernelCauchy delta 1;
 * localization.cpp:608-627, :318), the outlier gate on the prior estimate (:306-313), then
 * Localization::solve() = initializeOptimization + optimize(maximum_iteration) with g2o's
 * Levenberg-Marquardt (:164-170), then optimizer.chi2() (:197).  With identity antenna offsets the
 * 6-DoF g2o problem reduces exactly to a 3x3 one (SURVEY.md §8(a) note), which is what the kernel solves.
 *
 * Device layouts (all device pointers unless the name says host):
 *   dist, err : float  [K][M4][B][4]   M4 = ceil(M/4); anchor m sits at [m/4][..][m%4]; padded lanes err = 0
 *   pos       : double [3][B]          state carried across epochs (in: prior/initial estimate, out: last)
 *   out_pos   : double [K][3][B]       estimate after each epoch's solve
 *   out_chi2  : double [K][B]          optimizer.chi2() after each solve (non-robust, last evaluated state)
 *   out_trials: uint8  [K][B] or NULL  number of LM trials (linear solves) spent
 * ============================================================================================== */
typedef struct loc_snapshot loc_snapshot;

typedef struct loc_snapshot_params {
    int32_t maximum_iteration;   /* optimizer/maximum_iteration, localization.cpp:65 (default 20; cfg yaml: 10) */
    double distance_outlier;     /* robot/distance_outlier, localization.cpp:78; <= 0 disables the gate */
    int32_t gate_warmup_epochs;  /* epochs after (re)initialisation that run un-gated: the reference gates only once
                                    number_measurements > trajectory_length (localization.cpp:309). default 1 */
    int32_t jacobian;            /* LOC_JAC_*; loc_snapshot_default_params: LOC_JAC_NUMERIC_G2O (the reference's configuration) */
    int32_t lanes_per_instance;  /* 0 = library default; otherwise 1,2,4,8 (must divide the padded anchor count) */
    int32_t block_threads;       /* 0 = default (256) */
} loc_snapshot_params;

void loc_snapshot_default_params(loc_snapshot_params* p);

/* anchors_xyz_host: [M][3] doubles = /uwb/nodesPos of the static nodes (localization.cpp:86-108). */
int loc_snapshot_create(loc_snapshot** out, int32_t device, int64_t batch, int32_t n_anchors,
                        const double* anchors_xyz_host, const loc_snapshot_params* params);
int loc_snapshot_destroy(loc_snapshot* s);

int64_t loc_snapshot_batch(const loc_snapshot* s);
int32_t loc_snapshot_anchor_groups(const loc_snapshot* s);         /* M4 */
int32_t loc_snapshot_lanes_per_instance(const loc_snapshot* s);
/* number of floats in a dist/err buffer for `epochs` epochs = epochs * M4 * B * 4 */
size_t loc_snapshot_range_floats(const loc_snapshot* s, int32_t epochs);

/* Robot::init estimate (robot.cpp:47) for every tag: host [3][B] doubles -> device state, and back.
 * Setting positions restarts the gate warm-up (epoch counter = 0). */
int loc_snapshot_set_positions(loc_snapshot* s, const double* pos_soa_host);
int loc_snapshot_get_positions(loc_snapshot* s, double* pos_soa_host);
void* loc_snapshot_positions_device(loc_snapshot* s); /* the [3][B] device state itself */
int64_t loc_snapshot_epochs_done(const loc_snapshot* s);
int loc_snapshot_set_epochs_done(loc_snapshot* s, int64_t epochs);

/* Pack host ranges given as [K][M][B] (anchor-major SoA) into the device tile layout [K][M4][B][4]. */
int loc_snapshot_pack_ranges_host(const loc_snapshot* s, int32_t epochs, const float* src_kmb, float* dst_tiles,
                                  float pad_value);

/* The hot path: K epochs for all B tags, inputs and outputs resident in HBM. Asynchronous on hip_stream. */
int loc_snapshot_solve_device(loc_snapshot* s, int32_t epochs, const float* dist_dev, const float* err_dev,
                              double* out_pos_dev, double* out_chi2_dev, uint8_t* out_trials_dev,
                              void* hip_stream);

/* Convenience: same, with host buffers in the tile layout (stages over PCIe, synchronous). */
int loc_snapshot_solve_host(loc_snapshot* s, int32_t epochs, const float* dist_tiles_host,
                            const float* err_tiles_host, double* out_pos_host, double* out_chi2_host,
                            uint8_t* out_trials_host);

/* The batched mirror of the reference's per-message callback for callers that hold plain host arrays
 * (SURVEY.md 8(b) `batch_push_ranges` + `batch_solve` + `batch_get_positions`): distance / distance_err as
 * [K][M][B] float32 (epoch, anchor, tag — what K rounds of addRangeEdge, localization.cpp:297-376, would have been fed),
 * outputs [K][3][B] / [K][B].  Copy-in, tile packing + solve, and copy-out run as a chunked three-stream pipeline, so the
 * call costs about max(PCIe in, solve, PCIe out) rather than their sum — when the host buffers are page-locked
 * (loc_host_alloc); with pageable memory the result is the same, the copies just do not overlap.  Synchronous. */
int loc_snapshot_solve_host_kmb(loc_snapshot* s, int32_t epochs, const float* dist_kmb_host, const float* err_kmb_host,
                                double* out_pos_host, double* out_chi2_host, uint8_t* out_trials_host);

/* Page-locked host memory for the host paths above (hipHostMalloc / hipHostFree). */
int loc_host_alloc(void** out, size_t bytes);
int loc_host_free(void* p);

/* HIP-event timing of the solve kernel on the stream it is launched on (bench.py's roofline leg).
 * loc_snapshot_timing_begin() arms per-launch event pairs; _end() synchronises and returns the number of
 * timed launches, their total and average duration in milliseconds. */
int loc_snapshot_timing_begin(loc_snapshot* s, int32_t max_launches);
int loc_snapshot_timing_end(loc_snapshot* s, int32_t* n_launches, double* total_ms, double* avg_ms);


/* ================================================================================================
 * Batched sliding-window graph solver — the reference's general case (BASELINE configs 1, 3, 5 shapes):
 * B independent instances, each the graph one Localization object holds when it calls solve()
 * (localization.cpp:164-170): moving VertexSE3 poses (robot.cpp:75-110), fixed anchors, EdgeSE3Range factors
 * with an antenna lever arm on endpoint 0 (localization.cpp:331-340, types_edge_se3range.cpp:105-114; endpoint 1's through
 * loc_window_set_endpoint1_offsets),
 * EdgeSE3Prior with diagonal information (IMU / lidar, localization.cpp:476-486, 513-525) and EdgeSE3
 * (pose / twist, localization.cpp:263-281, 588-602).  Range and SE3 edges carry RobustKernelCauchy(1) as in the
 * reference (range always; SE3 per the `robust` flag); priors do not.  Range Jacobians: g2o's central differences by default
 * (the reference's configuration, types_edge_se3range.h:45-74); analytic ones with loc_window_set_jacobian.
 *
 * Host layouts (arrays of B instances, fixed capacities per instance):
 *   counts int32 [B][4]           nv, nr, np, ns  (poses, range edges, priors, SE3 edges actually used)
 *   poses  double[B][nv_max][12]  R row-major (9), t (3); in: estimates, out: optimised estimates
 *   r_idx  int32 [B][nr_max][2]   v0 = pose slot; v1 = pose slot, or -1 - anchor_index for a fixed anchor
 *   r_val  double[B][nr_max][5]   measurement, information (1/cov), lever arm xyz of endpoint 0
 *   p_idx  int32 [B][np_max]      pose slot
 *   p_val  double[B][np_max][18]  INVERSE measurement Z^-1 as R(9), t(3); information diagonal (6)
 *   s_idx  int32 [B][ns_max][4]   vi, vj, robust (0/1), 0
 *   s_val  double[B][ns_max][48]  INVERSE measurement Z^-1 as R(9), t(3); information 6x6 row-major
 *   result double[B][8]           chi2() over all edges at the last evaluated state, robust chi2 of the accepted
 *                                 state, final lambda, outer iterations run, LM trials, terminated flag, [6] number of
 *                                 pose-to-pose edges that share their pair of poses with another edge, [7] diagnostics of
 *                                 the factorisation (windows of <= 512 poses): elimination levels * 65536 + blocks of the
 *                                 factor + (poses in the dense root supernode) / 16; the one-lane-per-window kernel
 *                                 reports nv * 65536 + 2 nv - 1
 * The normal equations are kept as sparse 6x6 blocks (storage sized from the envelope bound nv_max * bw_max: per pose,
 * columns from its leftmost neighbour to itself), so storage and work scale with nv_max * bw_max^2, not nv_max^3:
 * cfg/uwb_pose.yaml's 500-pose chain is 3000 rows of ~12 entries.
 * Windows whose per-instance arrays fit 160 KiB keep everything in LDS (loc_window_lds_bytes tells); larger ones keep
 * them in a per-instance slice of an HBM workspace the handle allocates (a few MB for 500 poses).
 * ============================================================================================== */
typedef struct loc_window loc_window;
typedef struct loc_window_caps {
    int32_t nv_max, nr_max, np_max, ns_max;
    int32_t bw_max; /* widest pose-to-pose coupling |vi - vj| of any binary edge, in pose slots; < 0 = nv_max - 1 (dense).
                     * Pass the true bound: it sizes the per-instance storage, and small windows then fit twice as many
                     * instances per CU (measured: 4096 ten-pose chains 3.3 ms with -1, 1.8 ms with 1). */
} loc_window_caps;

int loc_window_create(loc_window** out, int32_t device, int64_t batch, const loc_window_caps* caps,
                      int32_t n_anchors, const double* anchors_xyz_host, int32_t maximum_iteration);
int loc_window_destroy(loc_window* w);
/* replace the fixed-vertex table (re-uploads; grows the device buffer when needed) */
int loc_window_set_anchors(loc_window* w, int32_t n_anchors, const double* anchors_xyz_host);
size_t loc_window_lds_bytes(const loc_window_caps* caps);
/* Synchronous: stages the B instances over PCIe, runs one launch, copies poses and results back. */
int loc_window_solve_host(loc_window* w, int64_t n_instances, const int32_t* counts, double* poses,
                          const int32_t* r_idx, const double* r_val, const int32_t* p_idx, const double* p_val,
                          const int32_t* s_idx, const double* s_val, double* result);
/* kernel time of the last loc_window_solve_host launch (HIP events on its stream), milliseconds */
int loc_window_last_kernel_ms(loc_window* w, double* ms);
/* EdgeSE3Range carries a lever arm per ENDPOINT (Isometry3d offset[2], types_edge_se3range.h:73; setVertexOffset(int, ...),
 * types_edge_se3range.cpp:99-103; both used in the residual, :108-112).  r_val holds endpoint 0's — the only one the reference
 * ever sets (localization.cpp:334).  This call supplies endpoint 1's for the instances of every LATER solve / upload: off1 =
 * [n_instances][nr_max][3] (xyz per range edge; for a fixed endpoint 1 — identity rotation — the point is the anchor + o1), or
 * NULL to go back to none.  Batches with endpoint-1 lever arms are solved by the general kernel (LOC_WINDOW_KERNEL_GENERAL). */
int loc_window_set_endpoint1_offsets(loc_window* w, int64_t n_instances, const double* off1_xyz);
/* LOC_JAC_NUMERIC_G2O (default: the reference's configuration) or LOC_JAC_ANALYTIC (opt-in fast mode) for the EdgeSE3Range
 * factors of every later solve */
int loc_window_set_jacobian(loc_window* w, int32_t jacobian);
/* Windows of up to 512 poses are eliminated in a minimum-degree order the kernel computes per instance (what CHOLMOD's AMD
 * ordering does for the reference, localization.h:82-84: a key-frame star then factors without fill, a chain with a dense
 * border is dissected into ~log2(n) levels); natural != 0 keeps the caller's pose order instead, and so does a window
 * whose fill under that order would not fit the storage bw_max sized.  Larger windows (513 ... 1024 poses) always use the
 * caller's order (loc_node_* packs them leaf-first). */
int loc_window_set_ordering(loc_window* w, int32_t natural);
/* Large batches of CHAIN windows — every pose-to-pose edge (range or SE3) joins consecutive poses (the smoothness edge of
 * Robot::new_vertex, robot.cpp:75-110; addTwistEdge's EdgeSE3, localization.cpp:438-459), edges listed in the order of their later
 * pose and priors in pose order (the order addRangeEdge / addImuEdge / addTwistEdge create them in: cfg/uwb_only.yaml,
 * cfg/uwb_imu.yaml, cfg/uwb_imu_lidar.yaml, cfg/uwb_twist.yaml) — are solved one GPU lane per window by a block-tridiagonal
 * kernel (same LM, elimination in pose order).  min_batch: the smallest batch that takes that
 * path (default 12 288, or LOCAMD_CHAIN_MIN_BATCH from the environment; 0: never; < 0: back to the default). */
int loc_window_set_chain_threshold(loc_window* w, int64_t min_batch);
/* Which kernel the last solve of this handle ran (the choice depends on the batch: its size and structure).
 *   GENERAL  window_lm_kernel: one wave (65 ... 512 poses: eight waves) per window, sparse block Cholesky in a minimum-degree order
 *   CHAIN    chain_lm_kernel: one lane per window, block-tridiagonal 6x6 (batches >= the chain threshold of chain windows)
 *   CHAIN3   chain3_lm_kernel: the same for TRANSLATION-ONLY batches (windows of more than 64 poses from 4 096 windows on; windows of <= 64
 *            poses only past an explicit threshold: WAVE3 serves those at every batch size) — no EdgeSE3, every lever arm zero, every rotation the
 *            identity (what Robot::init, robot.cpp:47, and the default identity antenna offsets, localization.h:170, give:
 *            cfg/uwb_only.yaml on the example bag), priors without rotation information: the 6-DoF problem then reduces EXACTLY
 *            to 3x3 blocks (types_edge_se3range.cpp:105-114 does not see the rotation; SURVEY.md §8(a) note)
 *   ARROW3   arrow3_lm_kernel: one wave per window for TRANSLATION-ONLY windows that are a chain with a small dense border — a
 *            trajectory whose poses range to up to 12 nodes that are unknowns themselves, held in the LAST pose slots (anchor
 *            self-calibration, BASELINE config 4; "every node moves", localization.cpp:94-98): the chain cut into four segments
 *            (one wave each), block-tridiagonal sweeps + the border's Schur complement on the f64 matrix cores; taken by
 *            windows of more than 64 poses (any batch size)
 *   TREE     tree_wave_kernel: batches (>= 256 windows) in which EVERY window has the same structure (same counts and index
 *            tables: one graph replayed with different measurements) and that structure is a forest of up to 64 poses (BASELINE
 *            config 5: the key-frame star of addPoseEdge, localization.cpp:254-290, + one anchor range per pose).  The fill-free
 *            elimination schedule is computed once on the host; one wave per window with LANE = POSE and a pose's whole solver
 *            state in that lane's registers (no workspace in memory), elimination by height.  (Nodes with several EdgeSE3 to
 *            their parent: tree_lm_kernel, one lane per window on the same schedule.)
 *   WAVE3   a translation-only chain batch (as CHAIN3) of windows of <= 64 poses, any batch size — in particular the drop-in node's
 *            own solve, one window per range message: wave3_lm_kernel, one wave per window, lane = edge for the residuals /
 *            Jacobians, lane = pose for the 3x3-block normal equations, rank-1 couplings solved by a scalar recurrence handed from
 *            lane to lane, the next LM trials' lambdas solved speculatively in the idle lanes.
 *   WAVE6   the 6-DoF sibling of WAVE3 (IMU / lidar priors, an antenna lever arm on the pose): chain windows of <= 64 poses without
 *            EdgeSE3 factors and with at most one range edge per pair of consecutive poses, any batch size (an explicit chain threshold
 *            hands larger batches to CHAIN).
 *   WAVE6S  the same kernel with FULL coupling blocks (wave6_lm_kernel<JAC, SE3>): chain windows of <= 63 poses WITH an EdgeSE3 factor between
 *            consecutive poses — Localization::addTwistEdge (localization.cpp:438-459, cfg/uwb_twist.yaml) — at most one per pair, at
 *            most one range edge per pair; the block Cholesky runs from both ends of the chain towards the middle pose.  Switched by
 *            option "wave6" as well.
 * All of them run the same LM and agree to the tolerances of DESIGN.md §3; result[6] / result[7] keep their meaning (the
 * lane-per-window kernels eliminate in pose order: result[7] = nv * 65536 + 2 nv - 1). */
enum { LOC_WINDOW_KERNEL_NONE = -1, LOC_WINDOW_KERNEL_GENERAL = 0, LOC_WINDOW_KERNEL_CHAIN = 1, LOC_WINDOW_KERNEL_CHAIN3 = 2, LOC_WINDOW_KERNEL_ARROW3 = 3, LOC_WINDOW_KERNEL_TREE = 4,
       LOC_WINDOW_KERNEL_TREE_LANE = 5 /* reported only: the lane-per-window variant of TREE ran */,
       LOC_WINDOW_KERNEL_WAVE3 = 6 /* translation-only chains of <= 64 poses: one wave per window (wave3_lm_kernel) */,
       LOC_WINDOW_KERNEL_WAVE6 = 7 /* 6-DoF chains of <= 64 poses (no EdgeSE3, one range edge per consecutive pair): wave6_lm_kernel */,
       LOC_WINDOW_KERNEL_WAVE6S = 8 /* the same with EdgeSE3 factors between consecutive poses (cfg/uwb_twist.yaml), at most one per pair: wave6_lm_kernel<.., SE3> */ };
int loc_window_last_kernel_kind(const loc_window* w, int32_t* kind);
/* Kernel-selection switches of ONE handle.  They are read from the environment ONCE, when the handle is created (LOCAMD_CHAIN_MIN_BATCH,
 * LOCAMD_ARROW3, LOCAMD_TREE, LOCAMD_WAVE3, LOCAMD_WAVE6, LOCAMD_CHAIN3, LOCAMD_NO_ZERO_COPY: A/B runs and tests), never at solve
 * time; this call changes them afterwards.  name / value:
 *   "chain_min_batch"  as loc_window_set_chain_threshold (0: never anything but the general kernel; < 0: the default rule)
 *   "arrow3"           -1 default (windows of more than 64 poses), 0 never, 1 whenever the batch qualifies
 *   "tree"             -1 default, 0 never, 2 the lane-per-window variant (tree_lm_kernel)
 *   "wave3" "wave6" "chain3" "zero_copy"   1 (default) / 0
 *   "topology_cache"   1 (default) / 0: reuse the structural verdict of the previous batch when counts and index tables hash the same
 *   "kernel_events"    1 (default; LOCAMD_KERNEL_EVENTS) / 0: no HIP events around the launch of a zero-copy solve (a handful of small windows);
 *                      loc_window_last_kernel_ms then reports launch-to-completion on the host clock.  The node's own handle runs with 0.
 * LOC_ERR_INVALID for an unknown name or value. */
int loc_window_set_option(loc_window* w, const char* name, int64_t value);
/* Host-side cost of the last loc_window_solve_host call, milliseconds: [0] argument validation, [1] structure analysis (kernel
 * choice: hash of the index tables, chain / forest / arrowhead tests, host-built schedules), [2] staging + launch + copy back +
 * synchronise, [3] 1.0 when the structural verdict came from the handle's cache, else 0.0 */
int loc_window_last_host_timing(const loc_window* w, double* validate_topology_run_cached);
/* Device-resident operation: upload n instances once (same host layouts as loc_window_solve_host), then run
 * loc_window_solve_resident any number of times — each launch starts from the uploaded estimates, is asynchronous on
 * hip_stream (NULL = the handle's own stream) and leaves poses / result on the device — and fetch them with
 * loc_window_download (synchronises).
 * Mixing with loc_window_solve_host on the same handle: every loc_window_solve_host first waits for the last resident launch (a
 * handle-owned event; the caller's stream is not kept).  A SMALL host solve (its inputs fit the 4 MiB staging block) leaves the
 * resident batch intact.  A LARGE one reuses the resident batch's device arrays: it DROPS the resident batch — the next
 * loc_window_solve_resident / loc_window_download return LOC_ERR_INVALID until loc_window_upload is called again.  loc_window_timing_begin/_end bracket resident launches with HIP events on the
 * stream they run on, like loc_snapshot_timing_*. */
int loc_window_upload(loc_window* w, int64_t n_instances, const int32_t* counts, const double* poses,
                      const int32_t* r_idx, const double* r_val, const int32_t* p_idx, const double* p_val,
                      const int32_t* s_idx, const double* s_val);
int loc_window_solve_resident(loc_window* w, void* hip_stream);
int loc_window_download(loc_window* w, double* poses, double* result);
void* loc_window_poses_device(loc_window* w);   /* double [B][nv_max][12] */
void* loc_window_result_device(loc_window* w);  /* double [B][8] */
int loc_window_timing_begin(loc_window* w, int32_t max_launches);
int loc_window_timing_end(loc_window* w, int32_t* n_launches, double* total_ms, double* avg_ms);

/* ================================================================================================
 * Node front-end — `class Localization` behind the ABI (one moving tag, its ring window, its anchors).
 * Same callbacks, same parameters, same gating as the reference; every solve() is one window-kernel launch.
 *   loc_node_create        Localization::Localization          localization.cpp:32-161 (+ Robot::init, robot.cpp:31-58)
 *   loc_node_add_range     Localization::addRangeEdge          localization.cpp:297-376
 *   loc_node_add_imu       Localization::addImuEdge            localization.cpp:499-535
 *   loc_node_add_pose      Localization::addPoseEdge           localization.cpp:254-290
 *   loc_node_add_twist     Localization::addTwistEdge          localization.cpp:438-459, 560-605
 *   loc_node_add_lidar     Localization::addLidarEdge          localization.cpp:462-496
 *   loc_node_add_rl_range  Localization::addRLRangeEdge        localization.cpp:378-436 (built only with -DRELATIVE_LOCALIZATION,
 *                                                              CMakeLists.txt:137; message uwb_reloc::uwbTalkData)
 *   loc_node_solve         Localization::solve + publish       localization.cpp:164-251
 *   loc_node_get_path      Robot::vertices2path                robot.cpp:61-72
 * add_* return 1 when the call ran a solve (out is filled), 0 when it did not, < 0 on error.
 * Poses are 8 doubles: stamp, x, y, z, qx, qy, qz, qw (the TUM order the reference logs, localization.cpp:630-642).
 * Limits of this kernel version: trajectory_length (x number of nodes with topic/relative_range) <= 1024 poses.
 * ============================================================================================== */
typedef struct loc_node loc_node;

typedef struct loc_node_config {
    int32_t trajectory_length;        /* robot/trajectory_length           localization.cpp:72 (no default) */
    double maximum_velocity;          /* robot/maximum_velocity (1.0)       :75 */
    double distance_outlier;          /* robot/distance_outlier (1.0)       :78 */
    int32_t maximum_iteration;        /* optimizer/maximum_iteration (20)   :65 */
    double minimum_optimize_error;    /* optimizer/minimum_optimize_error (1000)  :68 */
    int32_t publish_range, publish_pose, publish_twist, publish_lidar, publish_imu; /* publish_flag/...  :146-158 */
    int32_t has_relative_range;       /* topic/relative_range present: every node moves  :94 */
    int32_t jacobian;                 /* LOC_JAC_NUMERIC_G2O (default) = what the reference's EdgeSE3Range inherits from g2o
                                         (types_edge_se3range.h:45-74), or LOC_JAC_ANALYTIC (opt-in fast mode) */
    int32_t publish_relative_range;   /* publish_flag/relative_range  :158 */
} loc_node_config;

typedef struct loc_node_output {
    int32_t solved;                   /* a solve ran */
    int32_t published;                /* chi2 < minimum_optimize_error (localization.cpp:199-205) */
    double chi2;                      /* optimizer.chi2() */
    double realtime[8];               /* robots[self].current_pose()            :208 */
    double optimized[8];              /* path->poses[trajectory_length / 2]     :220 */
    int32_t outer_iterations, lm_trials;
} loc_node_output;

void loc_node_default_config(loc_node_config* c);
/* ids[n-1] is the moving tag (nodesId.back(), :89); pos_xyz = /uwb/nodesPos; antenna_xyz = /uwb/antennaOffset or NULL */
int loc_node_create(loc_node** out, int32_t device, const loc_node_config* cfg, int32_t n_nodes, const int32_t* ids,
                    const double* pos_xyz, int32_t n_antenna, const double* antenna_xyz);
int loc_node_destroy(loc_node* n);
int loc_node_add_range(loc_node* n, int32_t requester_id, int32_t responder_id, double stamp, float distance,
                       float distance_err, int32_t antenna, const char* frame_id, loc_node_output* out);
int loc_node_add_imu(loc_node* n, double stamp, const double* q_xyzw, const double* orientation_cov9,
                     const char* frame_id, loc_node_output* out);
int loc_node_add_pose(loc_node* n, double stamp, const double* pose_xyz_qxyzw, const double* cov36,
                      const char* frame_id, loc_node_output* out);
int loc_node_add_twist(loc_node* n, double stamp, const double* twist_lin_ang6, const double* cov36,
                       const char* frame_id, loc_node_output* out);
int loc_node_add_lidar(loc_node* n, double stamp, double z, const char* frame_id, loc_node_output* out);
/* uwbTalkData: time_stamp, rqstrId, rspdrId, d, (rqstr_vx, rqstr_vy, rqstr_vz).  Peer range with the fixed sigma_d = 0.054 m, the
 * responder's smoothness edge, and for a moving requester an EdgeSE3 from its previous pose with measurement
 * translate(dt * v) and information diag(1/sigma_v^2 x3, 0 x3), no robust kernel; solves when publish_relative_range. */
int loc_node_add_rl_range(loc_node* n, int32_t requester_id, int32_t responder_id, double stamp, double distance,
                          const double* requester_velocity_xyz, loc_node_output* out);
int loc_node_solve(loc_node* n, loc_node_output* out);
int loc_node_get_path(loc_node* n, int32_t node_id, double* out_T_by_8, int32_t capacity_poses);
int32_t loc_node_number_measurements(const loc_node* n);
/* Where the last solve's time went, milliseconds: [0] packing the window on the host, [1] the window solve call (copy in, launch, copy
 * out, synchronise), [2] of which the kernel: launch to completion on the host clock — HIP events around the kernel when LOCAMD_KERNEL_EVENTS=1 was
 * set when the node made its solver handle (they cost ~4 us per message).  The reference prints the same figure per solve (CPPTimer,
 * localization.cpp:166,191) */
int loc_node_last_timing(const loc_node* n, double* pack_solve_kernel_ms);
/* LOC_WINDOW_KERNEL_* of the node's last solve (LOC_WINDOW_KERNEL_NONE before the first) */
int loc_node_last_kernel_kind(const loc_node* n, int32_t* kind);
/* Localization::~Localization (localization.cpp:708-717): at destruction the reference appends path->poses[T/2 .. T-1] of the moving
 * tag to its "optimized" log (every earlier row of that log is a published path[T/2]), so that the log ends with the newest half of the
 * window.  Returns those rows (oldest first, 8 doubles each, the same poses loc_node_get_path gives at [T/2 .. T-1]) and their number;
 * LOC_ERR_INVALID when capacity_poses < T - T/2.  The node stays usable. */
int loc_node_flush_tail(loc_node* n, double* out_rows_by_8, int32_t capacity_poses);
/* Fleet mode: with deferred on, add_* only mark the node "solve pending"; loc_nodes_solve_batch then solves every
 * pending node of the array in ONE launch per parameter group (returns how many were solved). */
int loc_node_set_deferred(loc_node* n, int32_t on);
int32_t loc_node_solve_pending(const loc_node* n);
int loc_nodes_solve_batch(loc_node** nodes, int32_t n_nodes, loc_node_output* outs);
/* loc_nodes_solve_batch groups the pending nodes by (device, maximum_iteration, jacobian) — one launch per group, every
 * node solved with its own parameters — and keeps one batch solver per group cached for the calling thread; this frees them. */
int loc_nodes_release_batch_cache(void);

/* ================================================================================================
 * Batched fusion snapshot solver — BASELINE config 3 (8 anchors + IMU orientation prior, 6-DoF state).
 * Per tag and epoch, in the order the reference's callbacks produce it:
 *   IMU quaternion overwrites the rotation, translation kept                    localization.cpp:505-513
 *   EdgeSE3Prior on that pose, information diag(0,0,0,1/c0,1/c4,1/c8)            localization.cpp:515-525
 *   M EdgeSE3Range factors with the antenna lever arm on the tag side            localization.cpp:331-336
 *   outlier gate on the vertex origins, warm-up as in loc_snapshot_*             localization.cpp:306-313
 *   Localization::solve() on the 6-DoF vertex, optimizer.chi2()                  localization.cpp:164-170, 197
 * Device layouts:
 *   dist, err : float  [K][2][B][4]   as loc_snapshot_* with M <= 8
 *   imu       : double [K][B][8]      q x y z w, orientation_covariance[0], [4], [8], pad
 *   pose      : double [7][B]         state t xyz, q xyzw (in: initial, out: last); outputs [K][7][B], chi2 [K][B]
 * ============================================================================================== */
typedef struct loc_fusion loc_fusion;
typedef struct loc_fusion_params {
    int32_t maximum_iteration;   /* optimizer/maximum_iteration */
    double distance_outlier;     /* robot/distance_outlier; <= 0 disables */
    int32_t gate_warmup_epochs;  /* default 1 */
    double antenna_offset[3];    /* /uwb/antennaOffset of the antenna every range uses (localization.cpp:111-123, 333) */
    int32_t block_threads;       /* 0 = 256 */
    int32_t jacobian;            /* LOC_JAC_* for the range factors; default LOC_JAC_NUMERIC_G2O */
} loc_fusion_params;

void loc_fusion_default_params(loc_fusion_params* p);
int loc_fusion_create(loc_fusion** out, int32_t device, int64_t batch, int32_t n_anchors, const double* anchors_xyz_host,
                      const loc_fusion_params* params);
int loc_fusion_destroy(loc_fusion* f);
int loc_fusion_set_poses(loc_fusion* f, const double* pose_soa_host /* [7][B] */);
int loc_fusion_get_poses(loc_fusion* f, double* pose_soa_host);
int loc_fusion_solve_device(loc_fusion* f, int32_t epochs, const float* dist_dev, const float* err_dev, const double* imu_dev,
                            double* out_pose_dev, double* out_chi2_dev, uint8_t* out_trials_dev, void* hip_stream);
int loc_fusion_solve_host(loc_fusion* f, int32_t epochs, const float* dist_tiles_host, const float* err_tiles_host,
                          const double* imu_host, double* out_pose_host, double* out_chi2_host, uint8_t* out_trials_host);
/* Host arrays in their natural layout: dist/err [K][M][B] float32, imu [K][B][8]; tiles are packed on the GPU and the call
 * is pipelined like loc_snapshot_solve_host_kmb (page-locked buffers from loc_host_alloc overlap the copies). */
int loc_fusion_solve_host_kmb(loc_fusion* f, int32_t epochs, const float* dist_kmb_host, const float* err_kmb_host,
                              const double* imu_host, double* out_pose_host, double* out_chi2_host, uint8_t* out_trials_host);
int loc_fusion_last_kernel_ms(loc_fusion* f, double* ms);
/* per-launch HIP-event pairs on the launch stream, as loc_snapshot_timing_* */
int loc_fusion_timing_begin(loc_fusion* f, int32_t max_launches);
int loc_fusion_timing_end(loc_fusion* f, int32_t* n_launches, double* total_ms, double* avg_ms);

#ifdef __cplusplus
}
#endif
#endif /* LOCALIZATION_AMD_H */
