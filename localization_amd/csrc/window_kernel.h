// Internal interface of the sliding-window graph solver kernel (window_kernel.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace locamd {

// Per-batch capacities; every instance owns fixed-size slices of the arrays below.
struct WindowCaps {
    int nv_max;  // moving poses per instance
    int nr_max;  // range edges
    int np_max;  // unary SE3 priors
    int ns_max;  // binary SE3 edges   (edge tables and records must fit 160 KiB of LDS)
    int bw_max;  // widest coupling between poses of one instance, in pose slots (|vi - vj| of any binary edge)
};

// Layout of one instance (all arrays are [B][...]):
//   counts  int32 [B][4]            nv, nr, np, ns
//   poses   double[B][nv_max][12]   R row-major (9), t (3)            in: estimate, out: optimised
//   r_idx   int32 [B][nr_max][2]    v0 (moving slot), v1 (moving slot, or -1 - anchor index)
//   r_val   double[B][nr_max][5]    measurement, information, lever arm of endpoint 0 (xyz)
//   p_idx   int32 [B][np_max]       v
//   p_val   double[B][np_max][18]   Z^-1 as R(9), t(3); information diagonal (6)
//   s_idx   int32 [B][ns_max][4]    vi, vj, robust, pad
//   s_val   double[B][ns_max][48]   Z^-1 as R(9), t(3); information 6x6 row-major (36)
//   result  double[B][8]            chi2 (all edges, last evaluated), robust chi2, lambda, outer iterations,
//                                   LM trials, terminated, number of binary edges that share their pair of poses with another edge,
//                                   (nv_max <= 512) elimination-tree levels * 65536 + blocks of the factor + root-supernode poses / 16
struct WindowArgs {
    const int32_t* counts;
    const double* poses_in;  // initial estimates (may alias `poses`: every instance reads its poses before it writes them)
    double* poses;           // optimised estimates
    const int32_t* r_idx; const double* r_val;
    const double* r_off1;   // optional [B][nr_max][3]: lever arm of endpoint 1 (types_edge_se3range.h:73 offset[1]); nullptr = none (general kernel only)
    const int32_t* p_idx; const double* p_val;
    const int32_t* s_idx; const double* s_val;
    const double* anchors;  // [n_anchors][3] fixed vertices (identity rotation), shared by all instances
    double* result;
    double* workspace;  // nullptr: skyline H/L in LDS; else [B][window_workspace_doubles] in HBM (large windows)
    int n_anchors;
    int B;
    int iterations;
    int jacobian;       // 0: analytic range Jacobians, 1: g2o's central differences (delta = 1e-9)
    int natural_order;  // != 0: windows of <= 512 poses are eliminated in the caller's pose order (diagnostics / tests)
    WindowCaps caps;
};

size_t window_lds_bytes(const WindowCaps& c, bool global_a);
size_t window_workspace_doubles(const WindowCaps& c);
hipError_t launch_window(const WindowArgs& a, hipStream_t stream);
// chain windows, one lane per window (window_kernel.hip: chain_lm_kernel): workspace size in doubles, launch
size_t window_chain_workspace_doubles(const WindowCaps& c, long long B);
hipError_t launch_window_chain(const WindowArgs& a, double* chain_ws, hipStream_t stream);

// translation-only chain windows, one lane per window (chain3_kernel.hip)
size_t window_chain3_workspace_doubles(const WindowCaps& c, long long B);
int window_chain3_lds_mode(const WindowCaps& c, long long B, int n_cus);   // bit 0: (G, y) in LDS, bit 1: the translations in LDS
hipError_t launch_window_chain3(const WindowArgs& a, double* ws, hipStream_t stream);

// translation-only chain windows of <= 64 poses, one wave per window (wave3_kernel.hip): the node's own solve, small batches
size_t window_wave3_lds_bytes(const WindowCaps& c);
hipError_t launch_window_wave3(const WindowArgs& a, hipStream_t stream);

// 6-DoF chain windows of <= 64 poses without EdgeSE3 factors and with at most one range edge per pair of consecutive poses, one wave
// per window (wave6_kernel.hip): the node's own solve with IMU priors / lever arms, small batches
constexpr size_t kWave6MaxLds = 128 * 1024;   // per window (one workgroup of one wave)
size_t window_wave6_lds_bytes(const WindowCaps& c, bool se3 = false);   // se3: with the EdgeSE3 records of wave6_lm_kernel<JAC, true> (an EdgeSE3 per consecutive pair)
hipError_t launch_window_wave6(const WindowArgs& a, bool se3, hipStream_t stream);

// translation-only chain + dense border windows (arrow3_kernel.hip): four waves per window.  The host cuts the chain into up to
// four segments at separator poses (which join the border), orders the rows (chain rows by segment, then border rows) and packs
// every row's edges and priors as records [chunk of 64 rows][slot][lane] once per upload (capi_window.cpp: build_arrow_aux).
constexpr int kArrowMaxAnchors = 256;   // fixed anchors an edge of such a window may name: the table is staged in LDS (6 KB)
struct ArrowAux {
    const int32_t* hdr;     // [B][8]  nb (border poses incl. separators), nseg, n (chain rows), seg[0 .. 4] (chain rows of segment s: [seg[s], seg[s+1]))
    const int32_t* rslot;   // [B][nv_max] pose slot of row r (rows 0 .. n-1: chain, n .. n+nb-1: border)
    const double* rec;      // [B][nchunk][jmax][64][3]  code, measurement, information; code < 0: none; else (index << 3) | (kind << 1) | own-is-endpoint-0,
                            //                           kind 0: fixed anchor `index`, 1: the previous chain row, 2: border pose `index`
    const double* prec;     // [B][nchunk][jpmax][64][7] flag (> 0: a prior), Z^-1.t (3), information diagonal (3)
    double* ws;             // [B][window_arrow3_workspace_doubles]
    int nb_max, jmax, jpmax, nchunk;
    int jch[16], jpch[16];  // per chunk of 64 rows: the most edge / prior records any of its rows has, over the batch (<= jmax, jpmax; chunks from 16 on: jmax, jpmax)
};
size_t window_arrow3_workspace_doubles(const WindowCaps& c, int nb_max);
size_t window_arrow3_lds_bytes(const WindowCaps& c, int nb_max);
hipError_t launch_window_arrow3(const WindowArgs& a, const ArrowAux& x, hipStream_t stream);

// forest windows of ONE shared topology, one lane per window (tree_kernel.hip: tree_lm_kernel) or one wave per window (tree_wave_kernel.hip): the elimination schedule the
// host builds once per upload (capi_window.cpp: build_tree_sched); all pointers are device arrays shared by the whole batch
struct TreeSched {
    const int32_t* node;    // [nv]   pose slot of the k-th node in elimination order (children before their parent)
    const int32_t* par;     // [nv]   position (in that order) of the node's parent, -1 for a root
    const int32_t* r_off;   // [nv+1] the node's range edges (to anchors, or to its parent) = r_list[r_off[k] .. r_off[k+1])
    const int32_t* r_list;  // [nr]   edge numbers in schedule order (the kernel copies the values into its workspace in this order)
    const int32_t* p_off; const int32_t* p_list;   // priors of the node
    const int32_t* s_off; const int32_t* s_list;   // EdgeSE3 factors between the node and its parent
    const int32_t* r_idx;   // [nr][2], [ns][4]: the index tables of instance 0 (= of every instance)
    const int32_t* s_idx;
    // the same structure by POSE SLOT, for tree_wave_kernel (one wave per window, lane = pose): parent slot (-1: root), height above
    // the leaves, children (w_klist[w_koff[v] .. w_koff[v+1])), and the edges of the pose (unary ones, and those to its parent)
    const int32_t *w_par, *w_height, *w_koff, *w_klist, *w_roff, *w_rlist, *w_poff, *w_plist, *w_soff, *w_slist;
    const int32_t* w_kleaf;   // [nv] how many of a pose's children are leaves (they come first in w_klist)
    const int32_t* w_ulist;   // [nu] the inner nodes (height >= 1) by pose slot, PARENTS FIRST (decreasing height)
    const int32_t* w_kpos;    // [nv] a pose's position in w_klist (= the LDS column its hand-over to the parent uses: a node's children are
                              //      consecutive columns); roots: nv - nroots, nv - nroots + 1, ...
    int nv, nr, np, ns, depth, nroots, nlev, max_se3_per_node, nu, max_r_per_node;
};
size_t window_tree_workspace_doubles(const WindowCaps& c, long long B);
hipError_t launch_window_tree(const WindowArgs& a, const TreeSched& ts, double* ws, hipStream_t stream);
hipError_t launch_window_tree_wave(const WindowArgs& a, const TreeSched& ts, hipStream_t stream);

}  // namespace locamd
