// gfx950 (MI355X / CDNA4): tree_wave_kernel — forest windows of ONE shared topology (<= 64 poses), one WAVE per window, LANE = POSE.
// BASELINE config 5: the key-frame star of Localization::addPoseEdge (localization.cpp:254-290: every pose hangs on an older key
// pose by an EdgeSE3) plus one anchor range per pose (addRangeEdge, localization.cpp:297-376), g2o's Levenberg-Marquardt
// (Localization::solve, localization.cpp:164-170).  The same residuals and the same LM as every other window
// kernel here; the elimination schedule (parent, height, children and edge lists per pose slot) is built once per batch by the host
// (capi_window.cpp: build_tree_sched) and shared by every window.
//
// tree_lm_kernel (tree_kernel.hip) walks a window's 64 nodes one after the other in one lane.  Here the 64 nodes are the 64 lanes:
//   * linearisation and trial scoring: every lane evaluates ITS node's edges (its EdgeSE3 to the parent, its ranges, its priors);
//   * elimination by HEIGHT: all leaves factor their 6x6 block together, then the nodes whose children are done, ...; a node hands its
//     Schur update (27 numbers) to its parent through LDS; back-substitution the other way round, x handed down through LDS;
//   * H_nn, the coupling block with the parent, b, the factor, y, x and both poses of a node live in that lane's registers.
//
// Round 4 — what the cycle stamps (make treetiming, tools/dev/probe_tree.py) said about the round-3 version: 51 % of a solve was the
// linearisation, 11 % trial scoring, 9 % the sums over a key's leaf children — not arithmetic but ROUND TRIPS: with one wave per SIMD
// nothing hides a memory latency, and every sweep re-read its EdgeSE3 record (48 doubles per lane, 64 cache lines per load
// instruction) and its range record from HBM, chased the schedule tables through dependent global loads (edge number -> index
// table -> value), and spilled 0.3 KB per lane to scratch around the 48-double record.  Now NOTHING is read from memory after the
// prologue:
//   * a lane's EdgeSE3 record sits in LDS ([entry][lane]: conflict-free ds_read_b64, read where it is used — the 36 information
//     entries never occupy registers together), its first range record and that range's fixed endpoint in registers;
//   * every schedule look-up a lane needs (parent, height, its edges' numbers and flags, its first two non-leaf children) is resolved
//     once into registers;
//   * a node hands its 27 numbers up in the LDS COLUMN of its position in the children list, so that a node's children are consecutive
//     columns (leaf children first): sums over children run with LANE = ENTRY (27 lanes, one entry of the 21 + 6 each) over a column
//     range that one v_readlane of a packed per-node word yields — eight independent LDS reads in flight per inner node instead of 189
//     dependent reads (and a table look-up per child) in the node's own lane.
// HBM traffic per window: the inputs once (poses, the edge records) and the poses out.
// LDS: 48 x 64 + 27 x 65 doubles = 38 680 B per wave, four waves per CU.  Nodes with more than two range edges, or priors, take
// (correct, slower) generic loops over the tables in memory.
#include "se3_edge_device.h"

namespace locamd {
namespace {

// -DLOCAMD_TREE_TIMING: cycle stamps per phase, reported INSTEAD of result[0 .. 7] (tools/dev/probe_tree.py; never benchmarked)
#ifdef LOCAMD_TREE_TIMING
#define TW_T0() unsigned long long tw_tc = __builtin_readcyclecounter(), tw_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long tw_start = tw_tc
#define TW_T(k) do { const unsigned long long n_ = __builtin_readcyclecounter(); tw_ph[k] += n_ - tw_tc; tw_tc = n_; } while (0)
#else
#define TW_T0() do {} while (0)
#define TW_T(k) do {} while (0)
#endif

// what a lane knows about its node after the prologue (registers; nothing of this is looked up in memory again)
// Row stride of the [entry][column] hand-over / pose block in LDS, in doubles.  ODD: the block is read with lane = column (one entry,
// 64 columns: consecutive addresses) AND with lane = entry (the sums over children: one column, 27 entries) — with a stride of 64 the
// second pattern put all 27 lanes on one bank (a 27-way conflict per read: the sums took 7 k cycles per call instead of ~1 k).
constexpr int DS = 65;

struct LaneNode {
    int par, height;        // parent slot (-1: root), height above the leaves
    int r0, r1, q0, q1;     // this node's range edges = w_rlist[r0 .. r1), priors = w_plist[q0 .. q1)
    int se;                 // edge number of the EdgeSE3 to the parent, -1: none
    bool se_robust, se_icur;   // its robust flag; endpoint i of that edge is this node
    int sci, scj;           // ... the pose slots of its endpoints i and j (this node and its parent, either way round)
    int rv1;                // first range edge: endpoint 1 (pose slot, or -1 - anchor)
    bool r_first_is_cur;    // ... endpoint 0 (the one with the lever arm) is this node
    int rc0, rc1;           // ... the pose slots of its endpoints (rc1 = rc0 for a fixed endpoint 1)
    int r2v1, r2c0, r2c1;   // the SECOND range edge likewise (a chain pose: one anchor range + the smoothness range to its neighbour)
    bool r2_first_is_cur;
    int col;                // the LDS column this node's hand-over to its parent uses = its position in the children list
    int kn0, k1;            // columns of this node's NON-LEAF children: [kn0, k1) (a node's children are consecutive columns, leaves first)
};

__device__ __forceinline__ int lane_read(int v, int l) { return __builtin_amdgcn_readlane(v, l); }

// One range edge of this lane's node: chi sums; FULL: own block + b (D), the parent's share (Dp), coupling O (rows: the parent's).
// The poses of the edge's two endpoints are read from the pose block in LDS (xp: [entry][slot]; c0, c1: the slots) — never selected
// between two arrays in registers: `c ? A[k] : B[k]` over two local arrays turns into a load through a selected POINTER, which pins
// both arrays in scratch memory (and every sweep then waits for scratch loads).
template <bool FULL, int JAC>
__device__ __forceinline__ void range_edge_terms(const double meas, const double info, const double off0, const double off1, const double off2,
                                                 const bool first_is_cur, const bool binary, const double f0, const double f1, const double f2,
                                                 const double* xp, const int c0, const int c1, double& rsum, double& csum,
                                                 double* D, double* Dp, double* O, int& nbin) {
    const double off[3] = {off0, off1, off2};
    double X0[12], X1[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) { X0[k] = xp[k * DS + c0]; X1[k] = xp[k * DS + c1]; }
    double p0[3], p1[3];
    mat_vec(X0, off, p0);
    p0[0] += X0[9]; p0[1] += X0[10]; p0[2] += X0[11];
    p1[0] = binary ? X1[9] : f0; p1[1] = binary ? X1[10] : f1; p1[2] = binary ? X1[11] : f2;
    double u[3] = {p0[0] - p1[0], p0[1] - p1[1], p0[2] - p1[2]};
    const double nn = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    const double err = JAC == 0 ? meas - nn : range_error_plain(X0, X0 + 9, off, p1, meas);
    const double chi = err * (info * err);
    const double aux = 1.0 + chi;
    rsum += fast_log_ge1(aux);
    csum += chi;
    if (FULL) {
        double J0[6], J1[3];
        if (JAC == 0) {
            const double inv = nn > 0.0 ? 1.0 / nn : 0.0;
            u[0] *= inv; u[1] *= inv; u[2] *= inv;
            double uR[3];
            mat_tvec(X0, u, uR);
            J0[0] = -uR[0]; J0[1] = -uR[1]; J0[2] = -uR[2];
            J0[3] = 2.0 * (uR[1] * off[2] - uR[2] * off[1]);
            J0[4] = 2.0 * (uR[2] * off[0] - uR[0] * off[2]);
            J0[5] = 2.0 * (uR[0] * off[1] - uR[1] * off[0]);
            if (binary) mat_tvec(X1, u, J1); else { J1[0] = 0; J1[1] = 0; J1[2] = 0; }
        } else {
            J0[0] = range_jac_numeric<0>(X0, off, X1, p1, 0, meas);
            J0[1] = range_jac_numeric<1>(X0, off, X1, p1, 0, meas);
            J0[2] = range_jac_numeric<2>(X0, off, X1, p1, 0, meas);
            // With a zero lever arm a rotation of endpoint 0 does not move the ranged point at all: both perturbed evaluations of g2o's central
            // difference return the same number and the column is exactly +0 — the three columns are skipped when no lane of the wave has a lever arm.
            if (__any(off0 != 0.0 || off1 != 0.0 || off2 != 0.0)) {
                J0[3] = range_jac_numeric<3>(X0, off, X1, p1, 0, meas);
                J0[4] = range_jac_numeric<4>(X0, off, X1, p1, 0, meas);
                J0[5] = range_jac_numeric<5>(X0, off, X1, p1, 0, meas);
            } else { J0[3] = 0.0; J0[4] = 0.0; J0[5] = 0.0; }
            if (binary) {
                J1[0] = range_jac_numeric<0>(X0, off, X1, p1, 1, meas);
                J1[1] = range_jac_numeric<1>(X0, off, X1, p1, 1, meas);
                J1[2] = range_jac_numeric<2>(X0, off, X1, p1, 1, meas);
            } else { J1[0] = 0; J1[1] = 0; J1[2] = 0; }
        }
        const double wr = info / aux, wre = -wr * err;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
#pragma unroll
            for (int cc = 0; cc <= r; ++cc) {
                const double h = wr * J0[r] * J0[cc];
                D[r * (r + 1) / 2 + cc] += first_is_cur ? h : 0.0;
                Dp[r * (r + 1) / 2 + cc] += first_is_cur ? 0.0 : h;
            }
            const double bb = J0[r] * wre;
            D[21 + r] += first_is_cur ? bb : 0.0;
            Dp[21 + r] += first_is_cur ? 0.0 : bb;
        }
        if (binary) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int cc = 0; cc <= r; ++cc) {
                    const double h = wr * J1[r] * J1[cc];
                    D[r * (r + 1) / 2 + cc] += first_is_cur ? 0.0 : h;
                    Dp[r * (r + 1) / 2 + cc] += first_is_cur ? h : 0.0;
                }
                const double bb = J1[r] * wre;
                D[21 + r] += first_is_cur ? 0.0 : bb;
                Dp[21 + r] += first_is_cur ? bb : 0.0;
            }
            double jr[6], jc[6];   // rows: the parent's J, columns: this node's
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                jr[r] = first_is_cur ? (r < 3 ? J1[r] : 0.0) : J0[r];
                jc[r] = first_is_cur ? J0[r] : (r < 3 ? J1[r] : 0.0);
            }
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int cc = 0; cc < 6; ++cc) O[6 * cc + r] += wr * jr[r] * jc[cc];
            ++nbin;
        }
    }
}

// every edge of this lane's node, the poses taken from the pose block in LDS (xp: [entry][slot]): chi sums; FULL: own block + b (D), the
// parent's share (Dp), coupling O.  sv: this lane's column of the EdgeSE3 block in LDS; rm .. rf2: the first range edge's record
// (measurement, information, lever arm) and fixed endpoint, in registers.
template <bool FULL, int JAC>
__device__ __forceinline__ void wave_node_edges(const WindowArgs& a, const TreeSched& ts, long long inst, int lane, const LaneNode& nd,
                                                const double* sv, const double* xp, const double rm, const double rinfo, const double ro0, const double ro1,
                                                const double ro2, const double rf0, const double rf1, const double rf2,
                                                const double qm, const double qinfo, const double qo0, const double qo1, const double qo2,
                                                const double qf0, const double qf1, const double qf2,
                                                double& rsum, double& csum, double* D, double* Dp, double* O, int& nbin) {
    const WindowCaps& c = a.caps;
    if (FULL) {
#pragma unroll
        for (int k = 0; k < 27; ++k) { D[k] = 0.0; Dp[k] = 0.0; }
#pragma unroll
        for (int k = 0; k < 36; ++k) O[k] = 0.0;
    }
    nbin = 0;
    // The EdgeSE3 factor to the parent FIRST (at most one per node here: the host sends batches with more to tree_lm_kernel): it
    // writes its blocks straight into D / Dp / O — the 78 doubles of H_ii, H_jj and the coupling block next to the Jacobians would spill
    if (nd.se >= 0) {
        const bool i_is_cur = nd.se_icur;
        double rterm, chi;
        double Xi[12], Xj[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) { Xi[k] = xp[k * DS + nd.sci]; Xj[k] = xp[k * DS + nd.scj]; }
        if (FULL) {
            double Hii[21], Hjj[21], bi[6], bj[6];
            chi = chain_se3_terms<FULL, 64>(Xi, Xj, sv, nd.se_robust, i_is_cur, Hii, Hjj, O, bi, bj, rterm);
#pragma unroll
            for (int k = 0; k < 21; ++k) { D[k] = i_is_cur ? Hii[k] : Hjj[k]; Dp[k] = i_is_cur ? Hjj[k] : Hii[k]; }
#pragma unroll
            for (int k = 0; k < 6; ++k) { D[21 + k] = i_is_cur ? bi[k] : bj[k]; Dp[21 + k] = i_is_cur ? bj[k] : bi[k]; }
            ++nbin;
        } else {
            chi = chain_se3_terms<FULL, 64>(Xi, Xj, sv, nd.se_robust, i_is_cur, nullptr, nullptr, nullptr, nullptr, nullptr, rterm);
        }
        rsum += rterm;
        csum += chi;
    }
    {
        // the first two range edges' records are in registers; a node with more takes the others from memory, every sweep
        double meas = rm, info = rinfo, o0 = ro0, o1 = ro1, o2 = ro2, f0 = rf0, f1 = rf1, f2 = rf2;
        bool first_is_cur = nd.r_first_is_cur, binary = nd.rv1 >= 0;
        int c0 = nd.rc0, c1 = nd.rc1;
        for (int ri = nd.r0; ri < nd.r1; ++ri) {
            if (ri == nd.r0 + 1) {
                meas = qm; info = qinfo; o0 = qo0; o1 = qo1; o2 = qo2; f0 = qf0; f1 = qf1; f2 = qf2;
                first_is_cur = nd.r2_first_is_cur; binary = nd.r2v1 >= 0;
                c0 = nd.r2c0; c1 = nd.r2c1;
            } else if (ri != nd.r0) {
                const int e = ts.w_rlist[ri];
                const int v0 = ts.r_idx[2 * e], v1 = ts.r_idx[2 * e + 1];
                const double* val = a.r_val + ((size_t)inst * c.nr_max + e) * 5;
                meas = val[0]; info = val[1]; o0 = val[2]; o1 = val[3]; o2 = val[4];
                first_is_cur = v0 == lane; binary = v1 >= 0;
                c0 = v0; c1 = v1 >= 0 ? v1 : v0;
                const double* an = a.anchors + (size_t)(v1 >= 0 ? 0 : -1 - v1) * 3;
                f0 = an[0]; f1 = an[1]; f2 = an[2];
            }
            range_edge_terms<FULL, JAC>(meas, info, o0, o1, o2, first_is_cur, binary, f0, f1, f2, xp, c0, c1, rsum, csum, D, Dp, O, nbin);
        }
    }
    for (int qi = nd.q0; qi < nd.q1; ++qi) {
        const int e = ts.w_plist[qi];
        const double* pv = a.p_val + ((size_t)inst * c.np_max + e) * 18;
        double Zi[12], Wd[6], Xc[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) Xc[k] = xp[k * DS + lane];
#pragma unroll
        for (int k = 0; k < 12; ++k) Zi[k] = pv[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) Wd[k] = pv[12 + k];
        double RE[9], tE[3], qq[4];
        mat_mul(Zi, Xc, RE);
        mat_vec(Zi, Xc + 9, tE);
        tE[0] += Zi[9]; tE[1] += Zi[10]; tE[2] += Zi[11];
        mat_to_quat(RE, qq);
        quat_normalize_sign(qq);
        const double err[6] = {tE[0], tE[1], tE[2], qq[1], qq[2], qq[3]};
        double chi = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) chi += err[i] * (Wd[i] * err[i]);
        rsum += chi;
        csum += chi;
        if (FULL) {
            double J[36];
#pragma unroll
            for (int i = 0; i < 36; ++i) J[i] = 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) J[i * 6 + j] = RE[i * 3 + j];
            quat_right_jac(qq, 1.0, J, 6);
#pragma unroll
            for (int r = 0; r < 6; ++r) {
#pragma unroll
                for (int cc = 0; cc <= r; ++cc) {
                    double h = 0.0;
                    if ((r < 3) == (cc < 3)) {
#pragma unroll
                        for (int i = (r < 3 ? 0 : 3); i < (r < 3 ? 3 : 6); ++i) h += J[i * 6 + r] * Wd[i] * J[i * 6 + cc];
                    }
                    D[r * (r + 1) / 2 + cc] += h;
                }
                double bb = 0.0;
#pragma unroll
                for (int i = (r < 3 ? 0 : 3); i < (r < 3 ? 3 : 6); ++i) bb += J[i * 6 + r] * (-Wd[i] * err[i]);
                D[21 + r] += bb;
            }
        }
    }
}

template <int JAC>
__global__ void __launch_bounds__(64, 1) tree_wave_kernel(const WindowArgs a, const TreeSched ts) {
    __shared__ double sv_lds[48 * 64];    // [entry][lane]: the EdgeSE3 record of the lane's node (Z^-1 as R t, information 6x6)
    __shared__ double dep[27 * DS + 8];   // [entry][column], row stride DS (+ 8: sum_children reads eight columns at a time): what a node hands to its parent (linearisation share / Schur update);
                                          // between those uses: rows 0 .. 11 the poses (for the children), rows 12 .. 17 x (back-substitution)
#define DEP(k, l) dep[(k) * DS + (l)]
#define XP(k, l) dep[(k) * DS + (l)]
#define XX(k, l) dep[(12 + (k)) * DS + (l)]
    const int lane = threadIdx.x;
    const long long inst = blockIdx.x;
    const WindowCaps& c = a.caps;
    const int nv = ts.nv;
    const bool node = lane < nv;
    // ---- prologue: everything a sweep needs, once -------------------------------------------------------------------------------------
    LaneNode nd = {-1, -1, 0, 0, 0, 0, -1, false, false, 0, 0, -1, false, 0, 0, -1, 0, 0, false, lane, 0, 0};
    int kleaf = 0;
    double rm = 0, rinfo = 0, ro0 = 0, ro1 = 0, ro2 = 0, rf0 = 0, rf1 = 0, rf2 = 0;   // first range edge: measurement, information, lever arm; its fixed endpoint
    double qm = 0, qinfo = 0, qo0 = 0, qo1 = 0, qo2 = 0, qf0 = 0, qf1 = 0, qf2 = 0;   // second range edge
    // the tables the wave walks, one entry per lane: children list, first child / number of children / of leaf children per slot, inner nodes parents first
    int t_upack = 0;   // lane ui: the ui-th inner node (parents first): first child column | children << 8 | leaf children << 16 | own column << 24
    if (node) {
        nd.par = ts.w_par[lane]; nd.height = ts.w_height[lane];
        const int k0 = ts.w_koff[lane], k1 = ts.w_koff[lane + 1];
        kleaf = ts.w_kleaf[lane];
        nd.col = ts.w_kpos[lane];
        nd.kn0 = k0 + kleaf; nd.k1 = k1;
        nd.r0 = ts.w_roff[lane]; nd.r1 = ts.w_roff[lane + 1];
        nd.q0 = ts.w_poff[lane]; nd.q1 = ts.w_poff[lane + 1];
        const int s0 = ts.w_soff[lane], s1 = ts.w_soff[lane + 1];
        if (s0 < s1) {
            nd.se = ts.w_slist[s0];
            nd.sci = ts.s_idx[4 * nd.se]; nd.scj = ts.s_idx[4 * nd.se + 1];
            nd.se_icur = nd.sci == lane;
            nd.se_robust = ts.s_idx[4 * nd.se + 2] != 0;
        }
        if (nd.r0 < nd.r1) {
            const int e = ts.w_rlist[nd.r0];
            nd.rc0 = ts.r_idx[2 * e];
            nd.r_first_is_cur = nd.rc0 == lane;
            nd.rv1 = ts.r_idx[2 * e + 1];
            nd.rc1 = nd.rv1 >= 0 ? nd.rv1 : nd.rc0;
            const double* val = a.r_val + ((size_t)inst * c.nr_max + e) * 5;
            rm = val[0]; rinfo = val[1]; ro0 = val[2]; ro1 = val[3]; ro2 = val[4];
            if (nd.rv1 < 0) {
                const double* an = a.anchors + (size_t)(-1 - nd.rv1) * 3;
                rf0 = an[0]; rf1 = an[1]; rf2 = an[2];
            }
        }
        if (nd.r0 + 1 < nd.r1) {
            const int e = ts.w_rlist[nd.r0 + 1];
            nd.r2c0 = ts.r_idx[2 * e];
            nd.r2_first_is_cur = nd.r2c0 == lane;
            nd.r2v1 = ts.r_idx[2 * e + 1];
            nd.r2c1 = nd.r2v1 >= 0 ? nd.r2v1 : nd.r2c0;
            const double* val = a.r_val + ((size_t)inst * c.nr_max + e) * 5;
            qm = val[0]; qinfo = val[1]; qo0 = val[2]; qo1 = val[3]; qo2 = val[4];
            if (nd.r2v1 < 0) {
                const double* an = a.anchors + (size_t)(-1 - nd.r2v1) * 3;
                qf0 = an[0]; qf1 = an[1]; qf2 = an[2];
            }
        }
    }
    if (lane < ts.nu) {
        const int u = ts.w_ulist[lane];
        const int k0 = ts.w_koff[u];
        t_upack = k0 | ((ts.w_koff[u + 1] - k0) << 8) | (ts.w_kleaf[u] << 16) | (ts.w_kpos[u] << 24);
    }
    {
        // the lane's EdgeSE3 record -> LDS (the only pass over it: 48 doubles per lane)
        const double* val = a.s_val + ((size_t)inst * c.ns_max + (nd.se >= 0 ? nd.se : 0)) * 48;
        if (nd.se >= 0) {
#pragma unroll
            for (int k = 0; k < 48; ++k) sv_lds[k * 64 + lane] = val[k];
        }
    }
    const double* sv = sv_lds + lane;
    const double* gin = a.poses_in + ((size_t)inst * c.nv_max + (node ? lane : 0)) * 12;
    double* gout = a.poses + ((size_t)inst * c.nv_max + (node ? lane : 0)) * 12;
    double Xa[12], Xb[12];   // state / trial state
#pragma unroll
    for (int k = 0; k < 12; ++k) { Xa[k] = gin[k]; Xb[k] = Xa[k]; }
    double HD[21], HB[6], HO[36], Gs[15], ig[6], y[6], x[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { x[k] = 0.0; y[k] = 0.0; ig[k] = 0.0; HB[k] = 0.0; }
#pragma unroll
    for (int k = 0; k < 21; ++k) HD[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 36; ++k) HO[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 15; ++k) Gs[k] = 0.0;
    constexpr double tau = 1e-5, good_lo = 1.0 / 3.0, good_hi = 2.0 / 3.0;
    constexpr int max_trials = 10;
    double lambda = 0.0, ni = 2.0, cur_chi = 0.0, last_plain = 0.0;
    int it = 0, trials = 0, terminated = 0, shared_edges = 0;
    const bool active = nv > 0 && ts.nr + ts.np + ts.ns > 0 && a.iterations > 0;
    // (the lambdas are always_inline: inlined before the first SROA / InstCombine round — as ordinary lambdas the pose arrays they take by
    //  pointer stayed in memory long enough for `c ? A[k] : B[k]` to become a load through a selected POINTER, which pins both arrays in scratch)
    auto wsync = []() __attribute__((always_inline)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    // pose X of every node -> the pose block in LDS (rows 0 .. 11 of `dep`), where the edges read their endpoints'
    auto publish_pose = [&](const double* X) __attribute__((always_inline)) {
        wsync();
#pragma unroll
        for (int k = 0; k < 12; ++k) XP(k, lane) = X[k];
        wsync();
    };
    // For every inner node: the sum over its children (LEAVES_ONLY: its leaf children, which come first) of what they put into their `dep`
    // columns, left in the node's own column.  EIGHT inner nodes per round (parents first), eight lanes per node, each lane four of the
    // 21 + 6 entries: one ds_bpermute for the node's packed word, then up to 32 independent LDS reads in flight per lane (a node's children
    // are consecutive columns: immediate offsets), summed in list order (fixed order: bit-reproducible).  A node's own column is
    // overwritten only after its parent has read it: within a round every read is issued before the first write, and parents come in
    // the same or an earlier round.
    auto sum_children = [&](const bool leaves_only) __attribute__((always_inline)) {
        const int slot = lane >> 3, sub = lane & 7;
        for (int r0 = 0; r0 < ts.nu; r0 += 8) {
            const int ui = r0 + slot;
            const int pk = __builtin_amdgcn_ds_bpermute(4 * (ui < 64 ? ui : 63), t_upack);
            const bool mine = ui < ts.nu;
            const int k0 = pk & 255, cnt = mine ? (leaves_only ? (pk >> 16) & 255 : (pk >> 8) & 255) : 0, own = (pk >> 24) & 255;
            double s[4] = {0.0, 0.0, 0.0, 0.0};
            for (int c0 = 0; __any(c0 < cnt); c0 += 8) {
                double v[4][8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = sub + 8 * i;
                    const double* src = &DEP(e < 27 ? e : 0, k0 + c0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[i][j] = src[j];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) s[i] += c0 + j < cnt ? v[i][j] : 0.0;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = sub + 8 * i;
                if (mine && e < 27) DEP(e, own) = s[i];
            }
        }
    };
    TW_T0();
    wsync();
    if (active) {
        for (; it < a.iterations;) {
            TW_T(5);
            // ---- linearise at Xa: every lane its node's edges; the parent's shares go up through LDS ------------------------------
            {
                double D[27], Dp[27], rs = 0.0, cs = 0.0;
                int nbin;
                publish_pose(Xa);
                if (node) wave_node_edges<true, JAC>(a, ts, inst, lane, nd, sv, dep, rm, rinfo, ro0, ro1, ro2, rf0, rf1, rf2, qm, qinfo, qo0, qo1, qo2, qf0, qf1, qf2, rs, cs, D, Dp, HO, nbin);
                else {
#pragma unroll
                    for (int k = 0; k < 27; ++k) { D[k] = 0.0; Dp[k] = 0.0; }
#pragma unroll
                    for (int k = 0; k < 36; ++k) HO[k] = 0.0;   // (written on every path: HO is not live across iterations)
                    nbin = 0;
                }
                wsync();   // (every lane is done with the pose block before the hand-over rows overwrite it)
                TW_T(0);
#pragma unroll
                for (int k = 0; k < 27; ++k) DEP(k, nd.col) = Dp[k];
                wsync();
                sum_children(false);
                wsync();
                if (node && nd.height >= 1) {
#pragma unroll
                    for (int k = 0; k < 27; ++k) D[k] += DEP(k, nd.col);
                }
#pragma unroll
                for (int k = 0; k < 21; ++k) HD[k] = D[k];
#pragma unroll
                for (int k = 0; k < 6; ++k) HB[k] = D[21 + k];
                double md = 0.0;
#pragma unroll
                for (int r = 0; r < 6; ++r) md = fmax(md, fabs(HD[r * (r + 1) / 2 + r]));
                cur_chi = wave_sum(rs);
                last_plain = wave_sum(cs);
                if (it == 0) { lambda = tau * wave_max(node ? md : 0.0); ni = 2.0; shared_edges = (int)wave_sum(node && nbin >= 2 ? (double)nbin : 0.0); }
                wsync();
            }
            TW_T(7);
            int q = 0;
            double rho = 0.0;
            do {
                // ---- (H + lambda I) x = b: leaves first, by height; a node's Schur update goes to its parent through LDS ----------------
                bool ok = true;
                // (the factor of a trial is not live across trials / sweeps: without these the conditional writes below keep 27 doubles per
                //  lane alive through the linearisation, where registers are scarcest)
#pragma unroll
                for (int k = 0; k < 15; ++k) Gs[k] = 0.0;
#pragma unroll
                for (int k = 0; k < 6; ++k) { ig[k] = 0.0; y[k] = 0.0; }
                for (int h = 0; h < ts.nlev; ++h) {
                    if (h == 1) {
                        // every inner node's LEAF children summed once, lane = entry; the sum waits in the node's own slot, which the node
                        // only overwrites when it hands its own update up
                        sum_children(true);
                        wsync();
                        TW_T(2);
                    }
                    if (node && nd.height == h) {
                        double A[6][6], rhs[6];
#pragma unroll
                        for (int r = 0; r < 6; ++r) {
#pragma unroll
                            for (int cc = 0; cc <= r; ++cc) A[r][cc] = HD[r * (r + 1) / 2 + cc];
                            A[r][r] += lambda;
                            rhs[r] = HB[r];
                        }
                        if (h >= 1) {
#pragma unroll
                            for (int r = 0; r < 6; ++r) {
#pragma unroll
                                for (int cc = 0; cc <= r; ++cc) A[r][cc] -= DEP(r * (r + 1) / 2 + cc, nd.col);
                                rhs[r] -= DEP(21 + r, nd.col);
                            }
                            for (int ci = nd.kn0; ci < nd.k1; ++ci) {   // the non-leaf children: consecutive columns after the leaves'
#pragma unroll
                                for (int r = 0; r < 6; ++r) {
#pragma unroll
                                    for (int cc = 0; cc <= r; ++cc) A[r][cc] -= DEP(r * (r + 1) / 2 + cc, ci);
                                    rhs[r] -= DEP(21 + r, ci);
                                }
                            }
                        }
#pragma unroll
                        for (int j = 0; j < 6; ++j) {
                            const double g = pivot_rsqrt(A[j][j]);
                            ig[j] = g;
#pragma unroll
                            for (int i2 = j + 1; i2 < 6; ++i2) A[i2][j] *= g;
#pragma unroll
                            for (int i2 = j + 1; i2 < 6; ++i2)
#pragma unroll
                                for (int cc = j + 1; cc <= i2; ++cc) A[i2][cc] = __builtin_fma(-A[i2][j], A[cc][j], A[i2][cc]);
                        }
                        ok = (((ig[0] + ig[1]) + (ig[2] + ig[3])) + (ig[4] + ig[5]) < DBL_MAX);
#pragma unroll
                        for (int cc = 0; cc < 6; ++cc) {
                            rhs[cc] *= ig[cc];
#pragma unroll
                            for (int c2 = cc + 1; c2 < 6; ++c2) rhs[c2] = __builtin_fma(-rhs[cc], A[c2][cc], rhs[c2]);
                        }
                        {
                            int k = 0;
#pragma unroll
                            for (int cc = 0; cc < 5; ++cc)
#pragma unroll
                                for (int r = cc + 1; r < 6; ++r) { Gs[k] = A[r][cc]; ++k; }
                        }
#pragma unroll
                        for (int r = 0; r < 6; ++r) y[r] = rhs[r];
                        if (nd.par >= 0) {
                            double Wm[36];   // W = H_parent,n G^-T, entry (r, c) at 6 c + r
#pragma unroll
                            for (int r = 0; r < 6; ++r) {
                                double w[6];
#pragma unroll
                                for (int cc = 0; cc < 6; ++cc) w[cc] = HO[6 * cc + r];
#pragma unroll
                                for (int cc = 0; cc < 6; ++cc) {
                                    w[cc] *= ig[cc];
#pragma unroll
                                    for (int c2 = cc + 1; c2 < 6; ++c2) w[c2] = __builtin_fma(-w[cc], A[c2][cc], w[c2]);
                                }
#pragma unroll
                                for (int cc = 0; cc < 6; ++cc) Wm[6 * cc + r] = w[cc];
                            }
#pragma unroll
                            for (int r = 0; r < 6; ++r) {
#pragma unroll
                                for (int c2 = 0; c2 <= r; ++c2) {
                                    double s2 = 0.0;
#pragma unroll
                                    for (int k = 0; k < 6; ++k) s2 = __builtin_fma(Wm[6 * k + r], Wm[6 * k + c2], s2);
                                    DEP(r * (r + 1) / 2 + c2, nd.col) = s2;
                                }
                                double s3 = 0.0;
#pragma unroll
                                for (int k = 0; k < 6; ++k) s3 = __builtin_fma(Wm[6 * k + r], rhs[k], s3);
                                DEP(21 + r, nd.col) = s3;
                            }
                        }
                    }
                    wsync();
                    TW_T(h == 0 ? 1 : 3);
                }
                const bool all_ok = __ballot(node && !ok) == 0;
                double sc = 0.0;
                if (all_ok) {
                    // roots first: x_n = G_n^-T (y_n - G_n^-1 (H_parent,n^T x_parent)), x handed down through LDS
                    for (int h = ts.nlev - 1; h >= 0; --h) {
                        if (node && nd.height == h) {
                            double t[6], Gl[6][6];
#pragma unroll
                            for (int r = 0; r < 6; ++r) t[r] = y[r];
                            {
                                int k = 0;
#pragma unroll
                                for (int cc = 0; cc < 5; ++cc)
#pragma unroll
                                    for (int r = cc + 1; r < 6; ++r) { Gl[r][cc] = Gs[k]; ++k; }
                            }
                            if (nd.par >= 0) {
                                double xq[6], v[6];
#pragma unroll
                                for (int r = 0; r < 6; ++r) xq[r] = XX(r, nd.par);
#pragma unroll
                                for (int cc = 0; cc < 6; ++cc) {
                                    double a2 = 0.0;
#pragma unroll
                                    for (int r = 0; r < 6; ++r) a2 = __builtin_fma(HO[6 * cc + r], xq[r], a2);
                                    v[cc] = a2;
                                }
#pragma unroll
                                for (int cc = 0; cc < 6; ++cc) {
                                    v[cc] *= ig[cc];
#pragma unroll
                                    for (int c2 = cc + 1; c2 < 6; ++c2) v[c2] = __builtin_fma(-v[cc], Gl[c2][cc], v[c2]);
                                    t[cc] -= v[cc];
                                }
                            }
#pragma unroll
                            for (int rr = 5; rr >= 0; --rr) {
                                x[rr] = t[rr] * ig[rr];
#pragma unroll
                                for (int q2 = 0; q2 < rr; ++q2) t[q2] = __builtin_fma(-Gl[rr][q2], x[rr], t[q2]);
                            }
#pragma unroll
                            for (int r = 0; r < 6; ++r) XX(r, lane) = x[r];
                        }
                        wsync();
                    }
                }
                TW_T(4);
                // (a failed factorisation leaves every x as it was: g2o applies the stale x all the same)
                if (node) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) sc += x[k] * (lambda * x[k] + HB[k]);
                    double Rd[9];
                    const double ww = 1.0 - (x[3] * x[3] + x[4] * x[4] + x[5] * x[5]);
                    if (ww < 0) { Rd[0] = 1; Rd[1] = 0; Rd[2] = 0; Rd[3] = 0; Rd[4] = 1; Rd[5] = 0; Rd[6] = 0; Rd[7] = 0; Rd[8] = 1; }
                    else { const double qd[4] = {sqrt(ww), x[3], x[4], x[5]}; quat_to_mat(qd, Rd); }
                    double tn[3];
                    mat_mul(Xa, Rd, Xb);
                    mat_vec(Xa, x, tn);
                    Xb[9] = Xa[9] + tn[0]; Xb[10] = Xa[10] + tn[1]; Xb[11] = Xa[11] + tn[2];
                } else {
#pragma unroll
                    for (int k = 0; k < 12; ++k) Xb[k] = Xa[k];
                }
                const double scale = wave_sum(sc) + 1e-3;
                ++trials;
                TW_T(5);
                // ---- score the trial state -------------------------------------------------------------------------------------------------
                double temp_chi;
                {
                    double rs = 0.0, cs = 0.0, D[1], Dp[1], O[1];
                    int nbin;
                    publish_pose(Xb);
                    if (node) wave_node_edges<false, JAC>(a, ts, inst, lane, nd, sv, dep, rm, rinfo, ro0, ro1, ro2, rf0, rf1, rf2, qm, qinfo, qo0, qo1, qo2, qf0, qf1, qf2, rs, cs, D, Dp, O, nbin);
                    temp_chi = wave_sum(rs);
                    last_plain = wave_sum(cs);
                }
                TW_T(6);
                if (!all_ok) temp_chi = DBL_MAX;
                rho = (cur_chi - temp_chi) / scale;
                if (rho > 0.0 && fabs(temp_chi) <= DBL_MAX) {
                    const double r21 = 2.0 * rho - 1.0;
                    double alpha = 1.0 - r21 * r21 * r21;
                    alpha = fmin(alpha, good_hi);
                    lambda *= fmax(good_lo, alpha);
                    ni = 2.0;
                    cur_chi = temp_chi;
#pragma unroll
                    for (int k = 0; k < 12; ++k) Xa[k] = Xb[k];   // the trial state is the state
                } else {
                    lambda *= ni;
                    ni *= 2.0;
                }
                ++q;
            } while (rho < 0.0 && q < max_trials);
            ++it;
            if (q == max_trials || rho == 0.0) { terminated = 1; break; }
        }
    }
    if (node) {
#pragma unroll
        for (int k = 0; k < 12; ++k) gout[k] = Xa[k];
    }
    if (lane == 0) {
        double* res = a.result + (size_t)inst * 8;
        res[0] = last_plain; res[1] = cur_chi; res[2] = lambda; res[3] = (double)it; res[4] = (double)trials;
        res[5] = (double)terminated; res[6] = (double)shared_edges; res[7] = nv > 0 ? (double)(ts.nlev * 65536 + 2 * nv - ts.nroots) : 0.0;
#ifdef LOCAMD_TREE_TIMING
        for (int k = 0; k < 7; ++k) res[k] = (double)tw_ph[k];
        res[6] = (double)tw_ph[6] + 1e9 * (double)tw_ph[7];   // slot 7 (linearisation: hand-over + children sums) rides in res[6]'s upper digits
        res[7] = (double)(__builtin_readcyclecounter() - tw_start) + 1e12 * trials;
#endif
    }
#undef DEP
#undef XP
#undef XX
}

}  // namespace

hipError_t launch_window_tree_wave(const WindowArgs& a, const TreeSched& ts, hipStream_t stream) {
    if (a.B <= 0 || ts.nv <= 0 || ts.nv > 64 || ts.nlev <= 0 || ts.nu < 0 || ts.nu > 64) return hipErrorInvalidValue;
    if (a.jacobian) hipLaunchKernelGGL((tree_wave_kernel<1>), dim3((unsigned)a.B), dim3(64), 0, stream, a, ts);
    else hipLaunchKernelGGL((tree_wave_kernel<0>), dim3((unsigned)a.B), dim3(64), 0, stream, a, ts);
    return hipGetLastError();
}

}  // namespace locamd
