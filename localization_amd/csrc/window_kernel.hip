// gfx950 (MI355X / CDNA4) sliding-window graph solver: the reference's general case.
//
// One instance = one trajectory's window as Localization/Robot build it (reference file:line in brackets):
//   vertices  g2o::VertexSE3 poses of the moving node(s), ring window              [robot.cpp:31-58, 75-110]
//             fixed anchors (identity rotation)                                    [localization.cpp:100-106]
//   edges     EdgeSE3Range  e = d - ||(X0*O0).t - X1.t||, Cauchy                    [types_edge_se3range.cpp:105-114,
//                 range to an anchor (lever arm O0 = antenna offset), and the         localization.cpp:331-340, 608-627]
//                 zero-range smoothness edge between consecutive poses
//             EdgeSE3Prior  e = toVectorMQT(Z^-1 X), diagonal information           [localization.cpp:476-486, 513-525]
//                 (IMU: rotation only; lidar: z only), not robust
//             EdgeSE3       e = toVectorMQT(Z^-1 Xi^-1 Xj), 6x6 information, Cauchy  [localization.cpp:263-281, 588-602]
//   solve     initializeOptimization + optimize(maximum_iteration), g2o LM          [localization.cpp:164-170]
//             then optimizer.chi2()                                                 [localization.cpp:197]
// (g2o semantics: SURVEY.md Appendix A.)
//
// MI355X mapping: one wave per instance up to 64 poses, eight waves (one CU) per instance of 65 .. 512 poses.
//   * edge evaluation: lane e evaluates edge e (error, Jacobian blocks, robust weight) into a record; range Jacobians are
//     analytic, or g2o's central differences (delta = 1e-9, what the reference inherits) when WindowArgs::jacobian says so;
//   * normal equations: a gather over per-pose incidence lists (fixed order => bit-reproducible, no atomics);
//   * windows of up to 512 poses (SPARSE path): the poses are first ordered by multiple-minimum-degree / nested dissection
//     on adjacency bit masks (lane = pose; what CHOLMOD's AMD ordering does for the reference, localization.h:82-84), the
//     exact block structure of the Cholesky factor and its elimination-tree levels are derived, and the factorisation
//     runs LEVEL BY LEVEL: all block columns of a level are independent (6x6 blocks, the right-hand side rides along as
//     one more row).  BASELINE config 5's key-frame tree is 6 levels instead of 64 sequential block columns and has no
//     fill; a 500-pose chain is 10 levels; BASELINE config 4's arrowhead is 8 levels + a dense root supernode.
//       - <= 64 poses, arrays in LDS: factor_and_solve_small (one lane per entry / per row, pre-decoded schedule);
//       - <= 64 poses, arrays in the HBM workspace: factor_and_solve_sparse, one-word masks (column / row modes, pushed updates);
//       - 65 .. 512 poses: eight-word masks in LDS, eight waves, dense blocks summed cooperatively, the trailing clique
//         as one dense system in LDS whose sums run on the f64 matrix cores (root_factor_and_solve);
//   * larger windows (SKYLINE path, 513 .. 1024 poses): H and its factor in envelope form in the caller's pose order, one
//     block column after the other;
//   * small windows keep everything in LDS, larger ones in a per-instance slice of an HBM workspace (index tables in LDS);
//   * all LM control flow is uniform over the instance's threads (sums are DPP reductions, combined across the waves in wave
//     order: every thread holds the same bits).
#include "window_device.h"

#include <atomic>

namespace locamd {

namespace {

// Edge records written by the linearisation.  Unary and SE3 edges store their finished CONTRIBUTIONS to the normal equations
// (the 6x6 products are formed once, by the lane that evaluated the edge, from values it holds in registers), so building H
// and b is one load per (edge, entry); diagonal blocks as 21 lower-triangle entries k = r (r + 1) / 2 + c, r >= c.
constexpr int RREC = 16;   // J0[6] J1[6] wr omega_r (+pad): rank-1, expanded on the fly
constexpr int PREC = 28;   // H_vv[21] b_v[6] (+pad)
constexpr int SREC = 90;   // H_ii[21] H_jj[21] H_(later,earlier)[36, column-major like the storage] b_i[6] b_j[6]
constexpr int S_HJJ = 21, S_OFF = 42, S_BI = 78, S_BJ = 84;


// Stage-A record: out[entry] = in[entry] - sum_k a[k] b[k] for one earlier column K.  dst = index of the block / vector being
// formed (its source, H or b, lies a fixed distance below), a / b = indices of L_iK (or y_K) and L_JK; ctl bit 0 = the record has
// a K at all, bits 8.. = index of the record of the next K of the same block (0 = none).
struct alignas(16) RecA { int dst, a, b, ctl; };
// Stage-B record: dJ = offset of the column's diagonal block in H / L, rb = offset of the off-diagonal block (block jobs) or of
// the column's 6 entries in the vectors (column jobs).
struct alignas(8) RecB { int dJ, rb; };

struct Lds {
    double *Hs, *Ls;  // block-sparse H (lower) and its Cholesky factor (LDS, or an HBM workspace slice for large windows)
    double *diagL, *b, *x, *yrow, *pose, *bak, *rrec, *prec, *srec;
    int *boff;              // per block row: offset of its storage (boff[nv] = stored entries)
    int *fb, *last;         // SKYLINE path: first / last connected block of a block row / column
    // SPARSE path (nv <= 64), everything in elimination-order labels:
    u64 *rowmask, *colmask; //   rowmask[i]: block columns K <= i with L(i,K) != 0 (bit i set); colmask[J]: block rows i > J with L(i,J) != 0
    u64 *scr;               //   nv_max x W words of scratch for the ordering
    int W;                  //   words per mask: 1 (<= 64 poses) or 8 (<= 512 poses)
    int *rowpre;            //   W > 1: per (row, word) the number of blocks in the lower words (rank look-ups)
    u64 *pushw;             //   W > 1: the pushed columns as W words (W = 1: pushmask below)
    int *lvl_mode;          //   W > 1: per level 1 = column mode (W = 1: colmode below)
    int *perm;              //   caller's pose slot -> elimination position
    int *lvl_col, *lvl_blk; //   per elimination-tree level: first entry in colorder / otask (nlev + 1 entries)
    int *colorder;          //   block columns sorted by level
    int *otask;             //   off-diagonal blocks (i << 8 | J) sorted by the level of J
    int nlev;
    u64 colmode;            //   bit l: level l is factored one lane per column (every column has <= 1 off-diagonal block, or the level is wide)
    u64 pushmask;           //   bit K: column K has exactly one off-diagonal block (i, K) and is factored in column mode: its lane
                            //   also forms the update U_K = L_iK L_iK^T, u_K = L_iK y_K that the parent i subtracts ("pushed")
    double *Us, *As;        //   per column 28 doubles: pushed update (U lower triangle 21, u 6); per column: the sum of its children's
    int *ioff, *ilist;      // per pose: its incident edges in fold order (CSR), entry = kind << 28 | role << 27 | edge
    int *shared;            // [0] = count, then the binary edges (range e, or nr + SE3 e) whose pair of poses has another edge, in fold order
    double* blk; // 6x6 scratch: the diagonal block being factored (SKYLINE path)
    double* root;            // 8-word windows: LDS scratch of the dense root supernode ((6 m + 1) x 6 m doubles), or nullptr
    int root_level;          //   first level of the root supernode (nlev: none)
    int *dense, *dense_off;  // 8-word windows: the cooperatively summed (dense) blocks / right-hand sides, by level (compute_sparse_mw)
    int dense_cap, dense_ok; //   capacity; 0 = the list overflowed: nothing is summed cooperatively
    double* red; // several waves per window: NW doubles for the block-wide reductions, then NW x 36 for the cooperative sums
    // small windows (arrays in LDS, one-word masks): pre-decoded task records of the factorisation (build_small_tables)
    double* base;            //   the instance's LDS arrays as one array (record fields are indices into it)
    struct RecA* recA;       //   stage A: per (block, earlier column) pair
    struct RecB* recB;       //   stage B: per off-diagonal block, then per column
    int* rec_count;          //   records handed out so far
    int recA_cap;
#ifdef LOCAMD_WINDOW_TIMING
    long long* tim;  // diagnostic build: cycles in (a) segments, (b) block exchange+factor, (c) row finish, back-substitution
#endif
    // this instance's edge tables (staged from HBM once per launch; the SPARSE path relabels the indices in place)
    int32_t *r_idx, *p_idx, *s_idx;
    const double *r_val, *p_val, *s_val;
};

// Diagnostic build (-DLOCAMD_WINDOW_TIMING, tests/perf/probe_window_phases.py): cycles per phase, accumulated by lane 0 in
// LDS and written INSTEAD of the result record — 0 set-up (ordering, structure, incidence), 1 linearise, 2 build,
// 3 factor phase 1 (SKYLINE: the whole sweep), 4 factor phase 2, 5 back-substitution, 6 update + trial evaluation, 7 total.
#ifdef LOCAMD_WINDOW_TIMING
#define LOCAMD_TIC() const long long locamd_t0 = clock64()
#if LOCAMD_WINDOW_TIMING >= 2
// sub-phase stamps of the column mode (slots 0 loads, 1 updates from earlier columns, 2 Cholesky + right-hand side,
// 6 off-diagonal blocks) instead of the phases that normally use those slots
#define LOCAMD_TOC(slot) do { if (lane == 0 && (slot) != 0 && (slot) != 1 && (slot) != 2 && (slot) != 6) L.tim[slot] += clock64() - locamd_t0; } while (0)
// level 4: the factorisation by class of level: 0 column-mode levels, 1 row-mode levels without dense blocks, 2 / 6 phase 1 / 2 of
// the levels with cooperatively summed (dense) blocks
#define LOCAMD_TOC4(slot) do { if (lane == 0 && LOCAMD_WINDOW_TIMING == 4) L.tim[slot] += clock64() - locamd_t0; } while (0)
// level 5: the set-up: 0 ordering rounds, 1 relabelling + structure + levels + dense list, 2 incidence lists + shared pairs, 6 the rest
#define LOCAMD_STAMP5(slot) do { if (lane == 0 && LOCAMD_WINDOW_TIMING == 5) { const long long t = clock64(); L.tim[slot] += t - locamd_t5; locamd_t5 = t; } } while (0)
#define LOCAMD_T5_DECL long long locamd_t5 = clock64()
#define LOCAMD_SUB(slot) do { if (lane == 0 && LOCAMD_WINDOW_TIMING == 2) { const long long t = clock64(); L.tim[slot] += t - locamd_ts; locamd_ts = t; } } while (0)
// level 3: sub-phases of the back-substitution (0 index look-ups, 1 blocks of the column, 2 triangular solve, 6 stores + barrier)
#define LOCAMD_SUB3(slot) do { if (lane == 0 && LOCAMD_WINDOW_TIMING == 3) { const long long t = clock64(); L.tim[slot] += t - locamd_ts; locamd_ts = t; } } while (0)
#else
#define LOCAMD_TOC(slot) do { if (lane == 0) L.tim[slot] += clock64() - locamd_t0; } while (0)
#define LOCAMD_SUB(slot) do {} while (0)
#define LOCAMD_SUB3(slot) do {} while (0)
#define LOCAMD_TOC4(slot) do {} while (0)
#define LOCAMD_STAMP5(slot) do {} while (0)
#define LOCAMD_T5_DECL do {} while (0)
#endif
#else
#define LOCAMD_TIC() do {} while (0)
#define LOCAMD_TOC(slot) do {} while (0)
#define LOCAMD_SUB(slot) do {} while (0)
#define LOCAMD_SUB3(slot) do {} while (0)
#define LOCAMD_TOC4(slot) do {} while (0)
#define LOCAMD_STAMP5(slot) do {} while (0)
#define LOCAMD_T5_DECL do {} while (0)
#endif


template <bool SP>
__device__ __forceinline__ int blk_off(const Lds& L, int i, int K);

// Evaluate every edge at the current poses: errors + chi sums always, Jacobian/weight records when FULL.
// (FULL: the caller has zeroed H; an SE3 edge that is alone on its pair of poses — s_idx[4 e + 3] == 0, set once per solve —
//  writes its off-diagonal block straight into H, the others leave it in their record for the ordered accumulation.)
template <bool FULL, int JAC, bool SP, int NW>
__device__ __forceinline__ void evaluate_edges(const WindowArgs& a, const Lds& L, int inst, int lane, int nr, int np, int ns,
                               double& robust_chi, double& plain_chi) {
    (void)inst;
    constexpr int NT = 64 * NW;   // (`lane` is the thread's index in its window's workgroup: 0 .. NT - 1)
    double rsum = 0.0, csum = 0.0;
    // ---- range edges ------------------------------------------------------------------------------------------
    for (int e = lane; e < nr; e += NT) {
        const int32_t* idx = L.r_idx + e * 2;
        const double* val = L.r_val + e * 5;
        const int v0 = idx[0], v1 = idx[1];
        const double meas = val[0], info = val[1];
        const double off[3] = {val[2], val[3], val[4]};
        const double* X0 = L.pose + v0 * 12;
        double p0[3], p1[3];
        mat_vec(X0, off, p0);
        p0[0] += X0[9]; p0[1] += X0[10]; p0[2] += X0[11];
        if (v1 >= 0) { p1[0] = L.pose[v1 * 12 + 9]; p1[1] = L.pose[v1 * 12 + 10]; p1[2] = L.pose[v1 * 12 + 11]; }
        else { const double* an = a.anchors + (size_t)(-1 - v1) * 3; p1[0] = an[0]; p1[1] = an[1]; p1[2] = an[2]; }
        // optional lever arm on endpoint 1 (EdgeSE3Range::offset[1], types_edge_se3range.h:73, .cpp:99-103, 111): (X1 * O1).t; a fixed
        // vertex has the identity rotation (localization.cpp:100-106), so its point is the anchor + o1
        double off1[3] = {0.0, 0.0, 0.0};
        const bool has1 = a.r_off1 != nullptr;
        if (has1) {
            const double* o1 = a.r_off1 + ((size_t)inst * a.caps.nr_max + e) * 3;
            off1[0] = o1[0]; off1[1] = o1[1]; off1[2] = o1[2];
            if (v1 >= 0) perturbed_point_plain<0>(L.pose + v1 * 12, L.pose + v1 * 12 + 9, off1, 0.0, p1);   // X1 * o1, a plain CPU build's operation order
            else { p1[0] = off1[0] + p1[0]; p1[1] = off1[1] + p1[1]; p1[2] = off1[2] + p1[2]; }
        }
        double u[3] = {p0[0] - p1[0], p0[1] - p1[1], p0[2] - p1[2]};
        const double n = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        const double err = JAC == 0 ? meas - n : range_error_plain(X0, X0 + 9, off, p1, meas);
        const double chi = err * (info * err);
        const double aux = 1.0 + chi;
        rsum += fast_log_ge1(aux);
        csum += chi;
        if (FULL) {
            double* rec = L.rrec + e * RREC;
            if (JAC == 0) {
                const double inv = n > 0.0 ? 1.0 / n : 0.0;  // coincident endpoints: J = 0
                u[0] *= inv; u[1] *= inv; u[2] *= inv;
                double uR[3];
                mat_tvec(X0, u, uR);  // (u^T R0)^T
                rec[0] = -uR[0]; rec[1] = -uR[1]; rec[2] = -uR[2];
                // dp0/dv = -2 R0 [o]x  =>  de/dv0 = 2 (uR x o)
                rec[3] = 2.0 * (uR[1] * off[2] - uR[2] * off[1]);
                rec[4] = 2.0 * (uR[2] * off[0] - uR[0] * off[2]);
                rec[5] = 2.0 * (uR[0] * off[1] - uR[1] * off[0]);
                if (v1 >= 0) {
                    double uR1[3];
                    mat_tvec(L.pose + v1 * 12, u, uR1);
                    rec[6] = uR1[0]; rec[7] = uR1[1]; rec[8] = uR1[2];
                    // dp1/dv = -2 R1 [o1]x  =>  de/dv1 = -2 (uR1 x o1)
                    rec[9] = -2.0 * (uR1[1] * off1[2] - uR1[2] * off1[1]);
                    rec[10] = -2.0 * (uR1[2] * off1[0] - uR1[0] * off1[2]);
                    rec[11] = -2.0 * (uR1[0] * off1[1] - uR1[1] * off1[0]);
                } else { rec[6] = 0; rec[7] = 0; rec[8] = 0; rec[9] = 0; rec[10] = 0; rec[11] = 0; }
            } else {
                const double* X1 = v1 >= 0 ? L.pose + v1 * 12 : X0;
                rec[0] = range_jac_numeric<0>(X0, off, X1, p1, 0, meas);
                rec[1] = range_jac_numeric<1>(X0, off, X1, p1, 0, meas);
                rec[2] = range_jac_numeric<2>(X0, off, X1, p1, 0, meas);
                rec[3] = range_jac_numeric<3>(X0, off, X1, p1, 0, meas);
                rec[4] = range_jac_numeric<4>(X0, off, X1, p1, 0, meas);
                rec[5] = range_jac_numeric<5>(X0, off, X1, p1, 0, meas);
                const double* o1 = (has1 && v1 >= 0) ? off1 : nullptr;
                if (v1 >= 0) {  // (without a lever arm, rotating endpoint 1 does not move its point: those three columns are exactly 0)
                    rec[6] = range_jac_numeric<0>(X0, off, X1, p1, 1, meas, o1);
                    rec[7] = range_jac_numeric<1>(X0, off, X1, p1, 1, meas, o1);
                    rec[8] = range_jac_numeric<2>(X0, off, X1, p1, 1, meas, o1);
                    if (o1) {
                        rec[9] = range_jac_numeric<3>(X0, off, X1, p1, 1, meas, o1);
                        rec[10] = range_jac_numeric<4>(X0, off, X1, p1, 1, meas, o1);
                        rec[11] = range_jac_numeric<5>(X0, off, X1, p1, 1, meas, o1);
                    } else { rec[9] = 0; rec[10] = 0; rec[11] = 0; }
                } else { rec[6] = 0; rec[7] = 0; rec[8] = 0; rec[9] = 0; rec[10] = 0; rec[11] = 0; }
            }
            const double wr = info / aux;  // rho' * Omega
            rec[12] = wr;
            rec[13] = -wr * err;           // omega_r
        }
    }
    // ---- unary priors -------------------------------------------------------------------------------------------
    for (int e = lane; e < np; e += NT) {
        const int v = L.p_idx[e];
        const double* val = L.p_val + e * 18;
        const double* X = L.pose + v * 12;
        double RE[9], tE[3], q[4];
        mat_mul(val, X, RE);
        mat_vec(val, X + 9, tE);
        tE[0] += val[9]; tE[1] += val[10]; tE[2] += val[11];
        mat_to_quat(RE, q);
        quat_normalize_sign(q);
        const double err[6] = {tE[0], tE[1], tE[2], q[1], q[2], q[3]};
        double chi = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) chi += err[i] * (val[12 + i] * err[i]);
        rsum += chi;  // not robust
        csum += chi;
        if (FULL) {
            double* rec = L.prec + e * PREC;
            double J[36];
#pragma unroll
            for (int i = 0; i < 36; ++i) J[i] = 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) J[i * 6 + j] = RE[i * 3 + j];
            quat_right_jac(q, 1.0, J, 6);
            const double* W = val + 12;
            // J = blockdiag(RE, Q): row i of the Jacobian touches columns of its own 3-block only, so entry (r, cc) of
            // J^T W J is non-zero only inside a diagonal 3-block (IEEE arithmetic cannot drop 0 * x by itself)
#pragma unroll
            for (int r = 0; r < 6; ++r) {
#pragma unroll
                for (int cc = 0; cc <= r; ++cc) {
                    double h = 0.0;
                    if ((r < 3) == (cc < 3)) {
#pragma unroll
                        for (int i = (r < 3 ? 0 : 3); i < (r < 3 ? 3 : 6); ++i) h += J[i * 6 + r] * W[i] * J[i * 6 + cc];
                    }
                    rec[r * (r + 1) / 2 + cc] = h;
                }
                double bb = 0.0;
#pragma unroll
                for (int i = (r < 3 ? 0 : 3); i < (r < 3 ? 3 : 6); ++i) bb += J[i * 6 + r] * (-W[i] * err[i]);
                rec[21 + r] = bb;
            }
        }
    }
    // ---- binary SE3 edges ---------------------------------------------------------------------------------------
    for (int e = lane; e < ns; e += NT) {
        const int32_t* idx = L.s_idx + e * 4;
        const double* val = L.s_val + e * 48;
        const int vi = idx[0], vj = idx[1], robust = idx[2];
        const double* Xi = L.pose + vi * 12;
        const double* Xj = L.pose + vj * 12;
        double RB[9], tB[3], dt[3] = {Xj[9] - Xi[9], Xj[10] - Xi[10], Xj[11] - Xi[11]};
        mat_tmul(Xi, Xj, RB);   // Ri^T Rj
        mat_tvec(Xi, dt, tB);   // Ri^T (tj - ti)
        double RE[9], tE[3];
        mat_mul(val, RB, RE);
        mat_vec(val, tB, tE);
        tE[0] += val[9]; tE[1] += val[10]; tE[2] += val[11];
        double qE[4];
        mat_to_quat(RE, qE);
        quat_normalize_sign(qE);
        const double err[6] = {tE[0], tE[1], tE[2], qE[1], qE[2], qE[3]};
        const double* Om = val + 12;
        double Oe[6];
        double chi = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            double r = 0.0;
#pragma unroll
            for (int j = 0; j < 6; ++j) r += Om[i * 6 + j] * err[j];
            Oe[i] = r;
            chi += err[i] * r;
        }
        const double aux = 1.0 + chi;
        rsum += robust ? fast_log_ge1(aux) : chi;
        csum += chi;
        if (FULL) {
            double* rec = L.srec + e * SREC;
            const double w = robust ? 1.0 / aux : 1.0;
            double J0[36], J1[36];
#pragma unroll
            for (int i = 0; i < 36; ++i) { J0[i] = 0.0; J1[i] = 0.0; }
            // Jj : E' = E * Delta
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) J1[i * 6 + j] = RE[i * 3 + j];
            quat_right_jac(qE, 1.0, J1, 6);
            // Ji : E' = A * Delta^-1 * B
            const double S[9] = {0, -tB[2], tB[1], tB[2], 0, -tB[0], -tB[1], tB[0], 0};
            double RAS[9];
            mat_mul(val, S, RAS);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) { J0[i * 6 + j] = -val[i * 3 + j]; J0[i * 6 + 3 + j] = 2.0 * RAS[i * 3 + j]; }
            double qA[4], qB[4], qAB[4];
            mat_to_quat(val, qA);
            mat_to_quat(RB, qB);
            quat_mul(qA, qB, qAB);
            const double s = qAB[0] < 0 ? -1.0 : 1.0;
            const double nrm = 1.0 / sqrt(qAB[0] * qAB[0] + qAB[1] * qAB[1] + qAB[2] * qAB[2] + qAB[3] * qAB[3]);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                double ek[4] = {0, 0, 0, 0}, r1[4], r2[4];
                ek[1 + k] = 1.0;
                quat_mul(qA, ek, r1);
                quat_mul(r1, qB, r2);
#pragma unroll
                for (int i = 0; i < 3; ++i) J0[(3 + i) * 6 + 3 + k] = -s * nrm * r2[1 + i];
            }
            // contributions: WJ = w Omega J;  H_ii = J0^T WJ0, H_jj = J1^T WJ1, the off-diagonal block with the rows of the
            // later-labelled pose (J1^T WJ0 if vj > vi, else J0^T WJ1), b = J^T omega_r with omega_r = -w Omega e.
            // Structure (IEEE arithmetic cannot drop 0 * x by itself, so it is spelled out):
            //   J1 = [[RE, 0], [0, Q1]]: column c has rows [0,3) if c < 3, else [3,6);  J0 = [[-A, 2 R A S], [0, Q0]]: column c
            //   has rows [0,3) if c < 3, else [0,6).
#define LOCAMD_J1_LO(c) ((c) < 3 ? 0 : 3)
#define LOCAMD_J1_HI(c) ((c) < 3 ? 3 : 6)
#define LOCAMD_J0_HI(c) ((c) < 3 ? 3 : 6)
            double WJ[36];
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int cc = 0; cc < 6; ++cc) {
                    double s0 = 0.0;
#pragma unroll
                    for (int j = 0; j < LOCAMD_J0_HI(cc); ++j) s0 += Om[i * 6 + j] * J0[j * 6 + cc];
                    WJ[i * 6 + cc] = w * s0;
                }
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int cc = 0; cc <= r; ++cc) {
                    double h = 0.0;
#pragma unroll
                    for (int i = 0; i < LOCAMD_J0_HI(r); ++i) h += J0[i * 6 + r] * WJ[i * 6 + cc];
                    rec[r * (r + 1) / 2 + cc] = h;
                }
            double* const hoff = idx[3] ? rec + S_OFF : L.Hs + blk_off<SP>(L, max(vi, vj), min(vi, vj));
            if (vj > vi) {
#pragma unroll
                for (int r = 0; r < 6; ++r)
#pragma unroll
                    for (int cc = 0; cc < 6; ++cc) {
                        double h = 0.0;
#pragma unroll
                        for (int i = LOCAMD_J1_LO(r); i < LOCAMD_J1_HI(r); ++i) h += J1[i * 6 + r] * WJ[i * 6 + cc];
                        hoff[6 * cc + r] = h;
                    }
            }
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int cc = 0; cc < 6; ++cc) {
                    double s1 = 0.0;
#pragma unroll
                    for (int j = LOCAMD_J1_LO(cc); j < LOCAMD_J1_HI(cc); ++j) s1 += Om[i * 6 + j] * J1[j * 6 + cc];
                    WJ[i * 6 + cc] = w * s1;
                }
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int cc = 0; cc <= r; ++cc) {
                    double h = 0.0;
#pragma unroll
                    for (int i = LOCAMD_J1_LO(r); i < LOCAMD_J1_HI(r); ++i) h += J1[i * 6 + r] * WJ[i * 6 + cc];
                    rec[S_HJJ + r * (r + 1) / 2 + cc] = h;
                }
            if (vi > vj) {
#pragma unroll
                for (int r = 0; r < 6; ++r)
#pragma unroll
                    for (int cc = 0; cc < 6; ++cc) {
                        double h = 0.0;
#pragma unroll
                        for (int i = 0; i < LOCAMD_J0_HI(r); ++i) h += J0[i * 6 + r] * WJ[i * 6 + cc];
                        hoff[6 * cc + r] = h;
                    }
            }
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                double bi = 0.0, bj = 0.0;
#pragma unroll
                for (int i = 0; i < LOCAMD_J0_HI(r); ++i) bi += J0[i * 6 + r] * (-w * Oe[i]);
#pragma unroll
                for (int i = LOCAMD_J1_LO(r); i < LOCAMD_J1_HI(r); ++i) bj += J1[i * 6 + r] * (-w * Oe[i]);
                rec[S_BI + r] = bi;
                rec[S_BJ + r] = bj;
            }
#undef LOCAMD_J1_LO
#undef LOCAMD_J1_HI
#undef LOCAMD_J0_HI
        }
    }
    robust_chi = block_sum<NW>(rsum, L.red, lane);
    plain_chi = block_sum<NW>(csum, L.red, lane);
}

// ---- storage ---------------------------------------------------------------------------------------------------------
// H (lower triangle) and its Cholesky factor are stored by 6x6 blocks, the blocks of a block row one after the other in
// column order, each block COLUMN-MAJOR: entry (6 i + r, 6 K + c) sits at off(i, K) + 6 c + r.  Which blocks a row has:
//   SKYLINE (nv_max > 64): columns fb[i]..i, fb[i] = the leftmost block i is connected to; Cholesky never leaves that
//     envelope, so storage and work are O(n band^2): a 500-pose chain (cfg/uwb_pose.yaml) is 3000 rows of ~12 entries;
//   SPARSE  (nv_max <= 64): exactly the blocks of the factor's structure under the in-kernel elimination order (rowmask).
// Capacity in both cases: the envelope bound of the caller's bw_max (the exact structure of the natural order lies inside
// it; an elimination order that would need more falls back to the natural one).
__host__ __device__ inline size_t sky_nnz_bound(int nv, int bw) {
    size_t s = 0;
    for (int v = 0; v < nv; ++v) s += 36 * ((size_t)(v < bw ? v : bw) + 1);
    return s;
}
__host__ __device__ inline bool window_sparse_path(const WindowCaps& c) { return c.nv_max <= 512; }
// Capacity of H / L in entries.  Windows of 65 .. 512 poses keep them in the HBM workspace, where room is cheap: they get the
// envelope of a band of at least 3, so that a nested-dissection order of a long chain (one fill block per eliminated pose:
// 3 n blocks where the natural order has 2 n) fits and the window factors in ~log2(n) levels instead of n (a 500-pose chain
// of cfg/uwb_pose.yaml's length: 500 levels, 71 ms per solve with the caller's band of 1).
__host__ __device__ inline size_t window_nnz_capacity(const WindowCaps& c) {
    const int bw = (c.nv_max > 64 && c.nv_max <= 512 && c.bw_max < 3) ? (c.nv_max > 3 ? 3 : c.nv_max - 1) : c.bw_max;
    return sky_nnz_bound(c.nv_max, bw);
}
__host__ __device__ inline int window_mask_words(const WindowCaps& c) { return c.nv_max <= 64 ? 1 : 8; }

// entries of the per-pose incidence lists: every edge once per moving endpoint
__host__ __device__ inline size_t window_incidences(const WindowCaps& c) {
    return 2 * (size_t)c.nr_max + (size_t)c.np_max + 2 * (size_t)c.ns_max;
}
__host__ __device__ inline size_t ints_as_doubles(size_t n) { return (n + 1) / 2; }
// the edge-index tables and incidence lists of one instance, in doubles; in HBM-workspace mode they move to LDS as well when
// they are small (every gather starts with a look-up in them; from HBM that is a full memory round trip of pure latency)
__host__ __device__ inline size_t window_index_doubles(const WindowCaps& c) {
    return ints_as_doubles(window_incidences(c)) + ints_as_doubles((size_t)c.nr_max + c.ns_max + 1) +
           ints_as_doubles(2 * (size_t)c.nr_max) + ints_as_doubles((size_t)c.np_max) + ints_as_doubles(4 * (size_t)c.ns_max);
}
constexpr size_t INDEX_TABLES_LDS_MAX = 16 * 1024;        // windows of <= 64 poses: keeps 8+ waves per CU
constexpr size_t INDEX_TABLES_LDS_MAX_BIG = 72 * 1024;    // larger windows: their structure tables already limit the CU to 1-2 waves
__host__ __device__ inline size_t window_table_bytes(const WindowCaps& c);
__host__ __device__ inline bool window_index_in_lds(const WindowCaps& c) {
    const size_t bytes = window_index_doubles(c) * 8;
    if (c.nv_max <= 64) return bytes <= INDEX_TABLES_LDS_MAX;
    return bytes <= INDEX_TABLES_LDS_MAX_BIG && bytes + window_table_bytes(c) + 1024 <= 160 * 1024 - 512;
}

// 8-word windows: the trailing clique of the elimination order (BASELINE config 4's ten unknown anchors; the last separator of
// a nested dissection in general) is factored as ONE dense supernode of up to ROOT_MAX poses in LDS — when the LDS has room
constexpr int ROOT_MAX = 10;
__host__ __device__ inline size_t window_root_lds_doubles(const WindowCaps& c) {
    if (!window_sparse_path(c) || window_mask_words(c) == 1) return 0;
    const size_t need = (size_t)(6 * ROOT_MAX + 1) * (6 * ROOT_MAX);
    const size_t used = (window_table_bytes(c) + 7) / 8 * 8 + (window_index_in_lds(c) ? window_index_doubles(c) * 8 : 0);
    return used + need * 8 <= 160 * 1024 - 4096 ? need : 0;
}

// doubles of one instance's MAIN arrays (block-sparse pair, dense vectors, poses, edge records, incidence lists, a
// writable copy of the edge index tables) — the layout the kernel carves, in LDS or in the HBM workspace
__host__ __device__ inline size_t window_instance_doubles(const WindowCaps& c) {
    const size_t n_max = 6 * (size_t)c.nv_max;
    return 2 * window_nnz_capacity(c) + 4 * n_max + 2 * (size_t)c.nv_max * 12 + (size_t)c.nr_max * RREC + (size_t)c.np_max * PREC +
           (size_t)c.ns_max * SREC + window_index_doubles(c);
}
// workspace mode only: the pushed updates and their per-parent sums (28 doubles per column each), and for 8-word windows the
// off-diagonal task list
__host__ __device__ inline size_t window_push_doubles(const WindowCaps& c) {
    if (!window_sparse_path(c)) return 0;
    return 56 * (size_t)c.nv_max + (window_mask_words(c) > 1 ? ints_as_doubles(window_nnz_capacity(c) / 36) : 0);
}
// bytes of the small index tables that always live in LDS
__host__ __device__ inline size_t window_table_bytes(const WindowCaps& c) {
    const size_t nv = (size_t)c.nv_max;
    if (!window_sparse_path(c)) return (4 * nv + 2) * sizeof(int);  // fb, last, boff[nv + 1], ioff[nv + 1]
    const size_t nb_max = window_nnz_capacity(c) / 36;
    if (window_mask_words(c) == 1)
        return 3 * nv * sizeof(u64) + (6 * nv + 4 + nb_max) * sizeof(int);  // rowmask, colmask, scr; perm, boff, ioff, lvl_col, lvl_blk, colorder, otask
    // 8-word masks: rowmask, colmask, scr [nv][8], pushw [8]; rowpre [nv][8], perm, boff, ioff, lvl_col, lvl_blk, lvl_mode, colorder
    // (the off-diagonal task list of such windows lives in the workspace)
    // + dense_off [nv + 2], dense [2 nv]
    return (3 * nv * 8 + 8) * sizeof(u64) + (8 * nv + 7 * nv + 6 + 3 * nv + 2) * sizeof(int);
}

// small windows: records of the pre-decoded factorisation schedule — first-level stage-A records (two per column, one per
// off-diagonal block), continuation records for further earlier columns (capacity: one per block; a window that needs more
// runs the generic look-ups), stage-B records (one per block), the hand-out counter
__host__ __device__ inline size_t small_recA_capacity(const WindowCaps& c) {
    const size_t nb_max = window_nnz_capacity(c) / 36;
    return (size_t)c.nv_max + 2 * nb_max;
}
__host__ __device__ inline size_t small_table_doubles(const WindowCaps& c) {
    if (!window_sparse_path(c) || window_mask_words(c) != 1) return 0;
    return 2 * small_recA_capacity(c) + window_nnz_capacity(c) / 36 + 2;
}

// offset of block (i, K), K <= i, in the storage of H / L
template <bool SP>
__device__ __forceinline__ int blk_off(const Lds& L, int i, int K) {
    if (SP) {
        if (L.W == 1) return L.boff[i] + 36 * __popcll(L.rowmask[i] & ((1ull << K) - 1));
        const int w = K >> 6;
        return L.boff[i] + 36 * (L.rowpre[i * L.W + w] + __popcll(L.rowmask[i * L.W + w] & ((1ull << (K & 63)) - 1)));
    }
    return L.boff[i] + 36 * (K - L.fb[i]);
}
// address of H/L entry (row, col), col <= row, inside row's structure
template <bool SP>
__device__ __forceinline__ int sky(const Lds& L, int row, int col) {
    const int i = row / 6, K = col / 6;
    return blk_off<SP>(L, i, K) + 6 * (col - 6 * K) + (row - 6 * i);
}

constexpr int WIDE_LEVEL = 12;  // columns per level from which one lane takes a whole column whatever the columns look like
constexpr int INC_KIND_SHIFT = 28, INC_ROLE_SHIFT = 27, INC_EDGE_MASK = (1 << 27) - 1;

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int m = 32; m; m >>= 1) v = min(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int m = 32; m; m >>= 1) v = max(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ int wave_excl_scan_i(int v, int lane) {
    int s = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(s, d, 64); if (lane >= d) s += t; }
    return s - v;
}

// SKYLINE path, once per solve: fb[], last[], boff[] in the caller's pose order.
__device__ __forceinline__ void compute_skyline(const Lds& L, int lane, int nv, int nr, int ns) {
    if (lane == 0) {
        for (int v = 0; v < nv; ++v) { L.fb[v] = v; L.last[v] = v; }
        for (int e = 0; e < nr; ++e) {
            const int32_t* idx = L.r_idx + e * 2;
            if (idx[1] >= 0) { const int lo = min(idx[0], idx[1]), hi = max(idx[0], idx[1]); L.fb[hi] = min(L.fb[hi], lo); }
        }
        for (int e = 0; e < ns; ++e) {
            const int32_t* idx = L.s_idx + e * 4;
            const int lo = min(idx[0], idx[1]), hi = max(idx[0], idx[1]);
            L.fb[hi] = min(L.fb[hi], lo);
        }
        // last[J] = last block whose envelope reaches block column J
        for (int v = 0; v < nv; ++v) for (int J = L.fb[v]; J <= v; ++J) L.last[J] = max(L.last[J], v);
        int off = 0;
        for (int v = 0; v < nv; ++v) { L.boff[v] = off; off += 36 * (v - L.fb[v] + 1); }
        L.boff[nv] = off;
    }
    __syncthreads();
}

// SPARSE path, once per solve (nv <= 64, lane = pose): elimination order, structure of the factor, levels, task lists.
//   1. adjacency masks of the poses (binary edges), caller's labels;
//   2. multiple minimum degree: per round every pose of minimum degree that has no lower-numbered minimum-degree
//      neighbour is eliminated (an independent set, so the poses of a round do not interact); their remaining neighbours
//      become cliques.  `natural`: one pose per round in the caller's order instead (the fallback whose structure is
//      guaranteed to fit the envelope capacity);
//   3. relabel by elimination position; rowmask / colmask of the factor; block offsets; elimination-tree levels
//      (level(J) = 1 + max level of the columns in row J); columns and off-diagonal blocks sorted by level.
// Returns the number of blocks of the factor (the caller checks it against the capacity).
#ifndef LOCAMD_COLMODE_LDS
#define LOCAMD_COLMODE_LDS 1
#endif
#define LOCAMD_COLMODE_LDS_OK(push) ((push) || LOCAMD_COLMODE_LDS)
template <bool PUSH>
__device__ __forceinline__ int compute_sparse(Lds& L, int lane, int nv, int nv_max, int nb_max, int nr, int ns, bool natural, int& my_level) {
    if (lane < nv) L.scr[lane] = 0;
    __syncthreads();
    for (int e = lane; e < nr; e += 64) {
        const int v0 = L.r_idx[2 * e], v1 = L.r_idx[2 * e + 1];
        if (v1 >= 0) { atomicOr(&L.scr[v0], 1ull << v1); atomicOr(&L.scr[v1], 1ull << v0); }
    }
    for (int e = lane; e < ns; e += 64) {
        const int v0 = L.s_idx[4 * e], v1 = L.s_idx[4 * e + 1];
        atomicOr(&L.scr[v0], 1ull << v1); atomicOr(&L.scr[v1], 1ull << v0);
    }
    __syncthreads();
    const u64 me = 1ull << lane, below = me - 1;
    u64 a = lane < nv ? (L.scr[lane] & ~me) : 0;
    u64 remaining = nv >= 64 ? ~0ull : (1ull << nv) - 1;
    int pos = 0, mypos = 0;
    u64 mystruct = 0;
    while (remaining) {
        const bool alive = (remaining & me) != 0;
        u64 S;
        if (natural) {
            S = remaining & (~remaining + 1);
        } else {
            const int deg = alive ? __popcll(a & remaining) : 1000;
            const int mind = wave_min_i(deg);
            const u64 cand = __ballot(deg == mind);
            S = __ballot(alive && deg == mind && (a & cand & below) == 0);
        }
        const bool sel = (S & me) != 0;
        if (sel) { mypos = pos + __popcll(S & below); mystruct = a & remaining; }
        __syncthreads();
        if (sel) L.scr[lane] = mystruct;
        __syncthreads();
        if (alive && !sel) {
            u64 m = a & S;
            while (m) { const int v = __ffsll((long long)m) - 1; m &= m - 1; a |= L.scr[v]; }
            a &= ~me;
        }
        remaining &= ~S;
        pos += __popcll(S);
    }
    __syncthreads();
    if (lane < nv) L.perm[lane] = mypos;
    __syncthreads();
    {
        u64 cm = 0, m = mystruct;
        while (m) { const int u = __ffsll((long long)m) - 1; m &= m - 1; cm |= 1ull << L.perm[u]; }
        if (lane < nv) L.colmask[mypos] = cm;
    }
    __syncthreads();
    // from here on lane = elimination position
    const u64 mycol = lane < nv ? L.colmask[lane] : 0;
    u64 rm = lane < nv ? me : 0;
    for (int J = 0; J < nv; ++J) rm |= ((L.colmask[J] >> lane) & 1ull) << J;
    if (lane >= nv) rm = 0;
    if (lane < nv) L.rowmask[lane] = rm;
    const int cnt = __popcll(rm);
    const int off = wave_excl_scan_i(cnt, lane);
    const int nb = __shfl(off + cnt, 63, 64);
    if (lane < nv) L.boff[lane] = 36 * off;
    if (lane == 0) L.boff[nv] = 36 * nb;
    if (nb > nb_max) return nb;  // (uniform) does not fit the capacity: the caller falls back to the natural order
    int lev = 0;
    for (int J = 0; J < nv; ++J) {
        const int lJ = __shfl(lev, J, 64);
        if (((rm >> J) & 1ull) && J != lane) lev = max(lev, lJ + 1);
    }
    const int nlev = wave_max_i(lane < nv ? lev : 0) + 1;
    my_level = lane < nv ? lev : -1;
    int rank = 0, base = 0;
    const int nbc = __popcll(mycol);  // off-diagonal blocks in my column
    u64 colmode = 0, pushmask = 0;
    for (int l = 0; l < nlev; ++l) {
        const bool mine = lane < nv && lev == l;
        const u64 bl = __ballot(mine);
        if (lane == 0) L.lvl_col[l] = base;
        if (mine) rank = base + __popcll(bl & below);
        base += __popcll(bl);
        // column mode: every column of the level has at most one off-diagonal block (trees, chains), or the level is wide
        if (LOCAMD_COLMODE_LDS_OK(PUSH) && (__ballot(mine && nbc > 1) == 0 || __popcll(bl) >= WIDE_LEVEL)) {
            colmode |= 1ull << l;
            if (PUSH) pushmask |= __ballot(mine && nbc == 1);   // (in LDS the parent re-forms the product as cheaply as it loads it)
        }
    }
    L.colmode = colmode;
    L.pushmask = pushmask;
    if (lane == 0) L.lvl_col[nlev] = base;
    if (lane < nv) L.colorder[rank] = lane;
    // off-diagonal blocks of the columns, in level order
    int* tmp = reinterpret_cast<int*>(L.scr);  // 2 nv_max ints
    __syncthreads();
    if (lane < nv) tmp[rank] = nbc;
    __syncthreads();
    const int c_r = lane < nv ? tmp[lane] : 0;
    const int st = wave_excl_scan_i(c_r, lane);
    if (lane < nv) tmp[nv_max + lane] = st;
    __syncthreads();
    const int blkbase = lane < nv ? tmp[nv_max + rank] : 0;
    for (int l = lane; l < nlev; l += 64) L.lvl_blk[l] = tmp[nv_max + L.lvl_col[l]];
    if (lane == 0) L.lvl_blk[nlev] = nb - nv;
    {
        u64 m = mycol;
        int k = 0;
        while (m) { const int i = __ffsll((long long)m) - 1; m &= m - 1; L.otask[blkbase + k++] = (i << 16) | lane; }
    }
    L.nlev = nlev;
    __syncthreads();
    return nb;
}

// ---- structure masks of W 64-bit words per pose (W = 1: windows of <= 64 poses, masks in registers where possible;
//      W = 8: windows of <= 512 poses, masks in LDS) -------------------------------------------------------------------------
template <int W> __device__ __forceinline__ u64 rm_word(const Lds& L, int i, int w) { return L.rowmask[i * W + w]; }
template <int W> __device__ __forceinline__ u64 cm_word(const Lds& L, int J, int w) { return L.colmask[J * W + w]; }
// the bits of word w that lie below position J
template <int W> __device__ __forceinline__ u64 below_word(int J, int w) {
    if (W == 1) return (1ull << J) - 1;
    const int jw = J >> 6;
    return w < jw ? ~0ull : (w == jw ? (1ull << (J & 63)) - 1 : 0ull);
}
template <int W> __device__ __forceinline__ u64 push_word(const Lds& L, int w) { return W == 1 ? L.pushmask : L.pushw[w]; }
template <int W> __device__ __forceinline__ bool pushed(const Lds& L, int K) {
    if (W == 1) return (L.pushmask >> K) & 1ull;
    return (L.pushw[K >> 6] >> (K & 63)) & 1ull;
}
template <int W> __device__ __forceinline__ bool has_pushed_child(const Lds& L, int J) {
    u64 any = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) any |= rm_word<W>(L, J, w) & below_word<W>(J, w) & push_word<W>(L, w);
    return any != 0;
}
// number of blocks of row i left of column K
template <int W> __device__ __forceinline__ int row_rank(const Lds& L, int i, int K) {
    if (W == 1) return __popcll(L.rowmask[i] & ((1ull << K) - 1));
    const int w = K >> 6;
    return L.rowpre[i * W + w] + __popcll(L.rowmask[i * W + w] & ((1ull << (K & 63)) - 1));
}
// number of columns K < J that rows i and J share (the length of the product sum of block (i, J))
template <int W> __device__ __forceinline__ int common_count(const Lds& L, int i, int J) {
    int n = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) n += __popcll(rm_word<W>(L, i, w) & rm_word<W>(L, J, w) & below_word<W>(J, w));
    return n;
}
// A block whose product sum runs over at least DENSE_K earlier columns (the border of an arrowhead: BASELINE config 4's ten
// unknown anchors see all 256 tag poses, so each anchor-anchor block is a sum of 256 6x6 products) is summed by the WHOLE
// WAVE, lane = earlier column K, before the level's row tasks run: a row task would walk the 256 products one memory round
// trip after the other (it did: 90 % of the factorisation time of that shape, in the skyline sweep and here alike).
constexpr int DENSE_K = 24;
template <int W> __device__ __forceinline__ bool block_is_presummed(const Lds& L, int i, int J) {
    return W > 1 && L.dense_ok && common_count<W>(L, i, J) >= DENSE_K;
}
// P = sum_K L_iK L_JK^T over the common earlier columns, left in the (i, J) block of Ls (entry (r, c) at 6 c + r) for the row
// tasks to pick up; for the diagonal (i == J) only r >= c is meaningful.  One WAVE per block (the dense blocks of a level are
// independent: with several waves per window each wave takes its own); the caller's barrier follows the level's last block.
template <int W>
__device__ __forceinline__ void presum_block(const Lds& L, int lane, int i, int J, int Kend) {   // lane: 0 .. 63 within the wave; columns K < Kend <= J
    double acc[36];
#pragma unroll
    for (int q = 0; q < 36; ++q) acc[q] = 0.0;
    const int rowi = L.boff[i], rowJ = L.boff[J];
    for (int K0 = 0; K0 < Kend; K0 += 64) {
        const int K = K0 + lane;
        const int w = K >> 6;
        // (a pushed child's update reaches its parent through the reduction step: not again here)
        const bool has = K < Kend && (((L.rowmask[i * W + w] & L.rowmask[J * W + w]) >> (K & 63)) & 1ull) && !pushed<W>(L, K);
        if (has) {
            const double* bki = L.Ls + rowi + 36 * row_rank<W>(L, i, K);
            const double* bkj = L.Ls + rowJ + 36 * row_rank<W>(L, J, K);
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                double li[6], lj[6];
#pragma unroll
                for (int r = 0; r < 6; ++r) { li[r] = bki[6 * k + r]; lj[r] = bkj[6 * k + r]; }
#pragma unroll
                for (int c = 0; c < 6; ++c)
#pragma unroll
                    for (int r = 0; r < 6; ++r) acc[6 * c + r] = __builtin_fma(li[r], lj[c], acc[6 * c + r]);
            }
        }
    }
    double* dst = L.Ls + rowi + 36 * row_rank<W>(L, i, J);
#pragma unroll
    for (int q = 0; q < 36; ++q) {
        const double tot = wave_sum(acc[q]);
        if (lane == q) dst[q] = tot;
    }
}

// The right-hand side of a dense row likewise: u = sum_K L_JK y_K over the (non-pushed) earlier columns of row J, lane = K,
// left in yrow[6 J ..] (free until the row's right-hand-side task overwrites it with y_J).
template <int W> __device__ __forceinline__ bool rhs_is_presummed(const Lds& L, int J) {
    if (W == 1 || !L.dense_ok) return false;
    int n = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) n += __popcll(rm_word<W>(L, J, w) & below_word<W>(J, w));
    return n >= DENSE_K;
}
template <int W>
__device__ __forceinline__ void presum_rhs(const Lds& L, int lane, int J, int Kend) {   // lane: 0 .. 63 within the wave; columns K < Kend <= J
    double acc[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) acc[c] = 0.0;
    const int rowJ = L.boff[J];
    for (int K0 = 0; K0 < Kend; K0 += 64) {
        const int K = K0 + lane;
        const int w = K >> 6;
        const bool has = K < Kend && ((L.rowmask[J * W + w] >> (K & 63)) & 1ull) && !pushed<W>(L, K);
        if (has) {
            const double* bkj = L.Ls + rowJ + 36 * row_rank<W>(L, J, K);
            double yk[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) yk[k] = L.yrow[6 * K + k];
#pragma unroll
            for (int c = 0; c < 6; ++c)
#pragma unroll
                for (int k = 0; k < 6; ++k) acc[c] = __builtin_fma(yk[k], bkj[6 * k + c], acc[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const double tot = wave_sum(acc[c]);
        if (lane == c) L.yrow[6 * J + c] = tot;
    }
}


template <int W> __device__ __forceinline__ bool level_is_column_mode(const Lds& L, int l) {
    return W == 1 ? ((L.colmode >> l) & 1ull) != 0 : L.lvl_mode[l] != 0;
}

// Windows of 65 .. 512 poses: the same ordering / structure / levels as compute_sparse, with W-word masks kept in LDS and
// the poses handled in chunks of 64 (lane = pose within the chunk).  Two differences in the ordering, both for long chains
// with a dense border (BASELINE config 4: 256 tag poses, each tied to its two neighbours and to all 10 unknown anchors):
//   * a round takes every pose whose degree is within 1 of the minimum (the chain's interior poses have one neighbour more
//     than its ends), and
//   * among adjacent candidates the one whose RANK among the round's candidates (in index order) has the FEWER trailing zero
//     bits goes first (ties: the lower rank) — every second pose of a chain in the first round, every second of the rest in
//     the next, ...: a nested dissection, log2(n) levels for the chain instead of n / 2; the dense border vertices (huge
//     degree) come last.
// A round is a level (the poses of a round are an independent set, so their columns do not interact); elimination positions
// are handed out round by round, so the columns of a level are consecutive and colorder is the identity.
template <int W, bool PUSH, bool SOLO>
__device__ __forceinline__ int compute_sparse_mw(Lds& L, int lane, int nv, int nv_max, int nb_max, int nr, int ns, bool natural) {
    u64* adj = L.scr;      // [nv][W] current adjacency, caller's labels
    u64* st = L.rowmask;   // [nv][W] structure at elimination, caller's labels (the row masks overwrite it later)
    const int NC = (nv + 63) >> 6;
    LOCAMD_T5_DECL;
    for (int i = lane; i < nv * W; i += 64) adj[i] = 0;
    sync_<SOLO>();
    for (int e = lane; e < nr; e += 64) {
        const int v0 = L.r_idx[2 * e], v1 = L.r_idx[2 * e + 1];
        if (v1 >= 0) { atomicOr(&adj[v0 * W + (v1 >> 6)], 1ull << (v1 & 63)); atomicOr(&adj[v1 * W + (v0 >> 6)], 1ull << (v0 & 63)); }
    }
    for (int e = lane; e < ns; e += 64) {
        const int v0 = L.s_idx[4 * e], v1 = L.s_idx[4 * e + 1];
        atomicOr(&adj[v0 * W + (v1 >> 6)], 1ull << (v1 & 63)); atomicOr(&adj[v1 * W + (v0 >> 6)], 1ull << (v0 & 63));
    }
    sync_<SOLO>();
    for (int v = lane; v < nv; v += 64) adj[v * W + (v >> 6)] &= ~(1ull << (v & 63));
    sync_<SOLO>();
    const u64 me = 1ull << lane, below = me - 1;
    u64 rem[W];
#pragma unroll
    for (int w = 0; w < W; ++w) { const int lo = 64 * w; rem[w] = nv >= lo + 64 ? ~0ull : (nv > lo ? (1ull << (nv - lo)) - 1 : 0ull); }
    int pos = 0, round = 0;
    for (;;) {
        u64 any = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) any |= rem[w];
        if (!any) break;
        if (lane == 0) L.lvl_col[round] = pos;
        int degs[W];
        int mind = 1 << 30;
#pragma unroll
        for (int t = 0; t < W; ++t) {
            const int v = 64 * t + lane;
            degs[t] = 1 << 30;
            if (t < NC && ((rem[t] >> lane) & 1ull)) {
                int d = 0;
#pragma unroll
                for (int w = 0; w < W; ++w) d += __popcll(adj[v * W + w] & rem[w]);
                degs[t] = d;
            }
            mind = min(mind, degs[t]);
        }
        mind = wave_min_i(mind);
        u64 cand[W], S[W];
        if (natural) {
            bool found = false;
#pragma unroll
            for (int w = 0; w < W; ++w) { cand[w] = found ? 0ull : rem[w] & (~rem[w] + 1); found = found || rem[w] != 0; S[w] = cand[w]; }
        } else {
#pragma unroll
            for (int t = 0; t < W; ++t) cand[t] = __ballot(degs[t] <= mind + 1);
#pragma unroll
            for (int t = 0; t < W; ++t) {
                const int v = 64 * t + lane;
                bool sel = false;
                // priority among the candidates: by the candidate's RANK r in index order (so it does not matter how the chain
                // is numbered: the key poses 7, 15, 23, ... of a key-frame chain alternate just like 0, 1, 2, ...) — fewer
                // trailing zero bits of r first, then the lower rank
                auto prio = [&](int x) {
                    int r = 0;
#pragma unroll
                    for (int w = 0; w < W; ++w) r += __popcll(cand[w] & below_word<W>(x, w));
                    return ((r ? __ffs(r) - 1 : 31) << 16) | r;
                };
                if (t < NC && ((cand[t] >> lane) & 1ull)) {
                    sel = true;
                    const int pv = prio(v);
#pragma unroll
                    for (int w = 0; w < W; ++w) {
                        u64 m = adj[v * W + w] & cand[w];
                        while (m) { const int u = (w << 6) + __ffsll((long long)m) - 1; m &= m - 1; if (prio(u) < pv) sel = false; }
                    }
                }
                S[t] = __ballot(sel);
            }
        }
        // elimination positions of this round, structure at elimination
        int before = pos;
#pragma unroll
        for (int t = 0; t < W; ++t) {
            const int v = 64 * t + lane;
            if (t < NC && ((S[t] >> lane) & 1ull)) {
                L.perm[v] = before + __popcll(S[t] & below);
#pragma unroll
                for (int w = 0; w < W; ++w) st[v * W + w] = adj[v * W + w] & rem[w];
            }
            before += __popcll(S[t]);
        }
        sync_<SOLO>();
        // the remaining neighbours of an eliminated pose become a clique
#pragma unroll
        for (int t = 0; t < W; ++t) {
            const int u = 64 * t + lane;
            if (t < NC && ((rem[t] >> lane) & 1ull) && !((S[t] >> lane) & 1ull)) {
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    u64 m = adj[u * W + w] & S[w];
                    while (m) {
                        const int v = (w << 6) + __ffsll((long long)m) - 1;
                        m &= m - 1;
#pragma unroll
                        for (int w2 = 0; w2 < W; ++w2) adj[u * W + w2] |= st[v * W + w2];
                    }
                }
                adj[u * W + t] &= ~me;
            }
        }
        sync_<SOLO>();
#pragma unroll
        for (int w = 0; w < W; ++w) rem[w] &= ~S[w];
        pos = before;
        ++round;
    }
    const int nlev = round;
    LOCAMD_STAMP5(0);
    if (lane == 0) L.lvl_col[nlev] = pos;
    // relabel: colmask[position of v] = positions of the poses in st[v]
    for (int i = lane; i < nv * W; i += 64) L.colmask[i] = 0;
    sync_<SOLO>();
    for (int v = lane; v < nv; v += 64) {
        const int p = L.perm[v];
#pragma unroll
        for (int w = 0; w < W; ++w) {
            u64 m = st[v * W + w];
            while (m) { const int u = (w << 6) + __ffsll((long long)m) - 1; m &= m - 1; const int q = L.perm[u]; L.colmask[p * W + (q >> 6)] |= 1ull << (q & 63); }
        }
    }
    sync_<SOLO>();
    // row masks (transpose + diagonal), prefix counts, block offsets — lane = elimination position from here on
    int carry = 0;
    for (int c0 = 0; c0 < nv; c0 += 64) {
        const int i = c0 + lane;
        int cnt = 0;
        if (i < nv) {
            for (int wJ = 0; wJ < W; ++wJ) {
                u64 acc = 0;
                const int jn = min(64, nv - 64 * wJ);
                for (int b = 0; b < jn; ++b) acc |= ((L.colmask[(64 * wJ + b) * W + (i >> 6)] >> (i & 63)) & 1ull) << b;
                if (wJ == (i >> 6)) acc |= 1ull << (i & 63);
                L.rowmask[i * W + wJ] = acc;   // (st is dead: every colmask row is complete)
                L.rowpre[i * W + wJ] = cnt;
                cnt += __popcll(acc);
            }
        }
        const int ex = wave_excl_scan_i(cnt, lane) + carry;
        carry = __shfl(ex + cnt, 63, 64);
        if (i < nv) L.boff[i] = 36 * ex;
    }
    const int nb = carry;
    if (lane == 0) L.boff[nv] = 36 * nb;
    sync_<SOLO>();
    if (nb > nb_max) return nb;
    // levels: column mode or row mode, pushed columns; off-diagonal block tasks in level (= position) order
    for (int w = lane; w < W; w += 64) L.pushw[w] = 0;
    sync_<SOLO>();
    for (int l = 0; l < nlev; ++l) {
        const int j0 = L.lvl_col[l], j1 = L.lvl_col[l + 1];
        bool multi = false;
        for (int J = j0 + lane; J < j1; J += 64) {
            int nbc = 0, nk = 0;
#pragma unroll
            for (int w = 0; w < W; ++w) { nbc += __popcll(L.colmask[J * W + w]); nk += __popcll(L.rowmask[J * W + w] & below_word<W>(J, w)); }
            // (a column with a dense row — many earlier columns — wants the cooperative sums of the row mode: one lane walking
            //  an unknown anchor's 264 earlier columns of BASELINE config 4 was a sixth of that solve)
            multi = multi || nbc > 1 || nk >= DENSE_K;
        }
        // (with several waves per window a wide level is better off as row tasks over all of them: a lane per column leaves
        //  most of them idle — BASELINE config 4's first four levels took a third of the solve that way)
        const bool cm = __ballot(multi) == 0 || (!SOLO && (j1 - j0) >= WIDE_LEVEL);
        if (lane == 0) L.lvl_mode[l] = cm ? 1 : 0;
        if (cm && PUSH) {
            for (int J = j0 + lane; J < j1; J += 64) {
                int nbc = 0;
#pragma unroll
                for (int w = 0; w < W; ++w) nbc += __popcll(L.colmask[J * W + w]);
                if (nbc == 1) atomicOr(&L.pushw[J >> 6], 1ull << (J & 63));
            }
        }
    }
    carry = 0;
    for (int c0 = 0; c0 < nv; c0 += 64) {
        const int J = c0 + lane;
        int nbc = 0;
        if (J < nv) {
#pragma unroll
            for (int w = 0; w < W; ++w) nbc += __popcll(L.colmask[J * W + w]);
        }
        const int start = wave_excl_scan_i(nbc, lane) + carry;
        carry = __shfl(start + nbc, 63, 64);
        if (J < nv) {
            L.colorder[J] = J;
            L.ioff[J] = start;   // (ioff is free until compute_incidence: the level pointers are read from it just below)
            int k = 0;
            for (int w = 0; w < W; ++w) {
                u64 m = L.colmask[J * W + w];
                // (bits 25 .. 31: the number of earlier columns rows i and J share, capped at 127 — the row task of the block then
                //  knows without walking the eight mask words twice whether it has a product sum at all, and whether it is dense)
                while (m) {
                    const int i = (w << 6) + __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const int nc = min(common_count<W>(L, i, J), 127);
                    L.otask[start + k++] = (int)(((unsigned)nc << 25) | (unsigned)(i << 16) | (unsigned)J);
                }
            }
        }
    }
    sync_<SOLO>();
    for (int l = lane; l < nlev; l += 64) L.lvl_blk[l] = L.ioff[L.lvl_col[l]];
    if (lane == 0) L.lvl_blk[nlev] = nb - nv;
    L.nlev = nlev;
    L.colmode = 0; L.pushmask = 0;
    sync_<SOLO>();
    // the cooperatively summed (dense) blocks and right-hand sides, listed by level: block (i << 16 | J), right-hand side
    // (1 << 31 | J).  (The factorisation used to find them again on every level of every LM trial, one uniform look-up in the
    // task list — which lives in the HBM workspace — per block: a full memory round trip each.)
    {
        L.dense_ok = 1;
        int dcount = 0;
        for (int l = 0; l < nlev; ++l) {
            if (lane == 0) L.dense_off[l] = dcount;
            const int j0 = L.lvl_col[l], j1 = L.lvl_col[l + 1];
            for (int c0 = j0; c0 < j1; c0 += 64) {
                const int J = c0 + lane;
                const bool d = J < j1 && block_is_presummed<W>(L, J, J), r = J < j1 && rhs_is_presummed<W>(L, J);
                const u64 bd = __ballot(d), br = __ballot(r);
                const int pd = dcount + __popcll(bd & below);
                if (d && pd < L.dense_cap) L.dense[pd] = (J << 16) | J;
                dcount += __popcll(bd);
                const int pr = dcount + __popcll(br & below);
                if (r && pr < L.dense_cap) L.dense[pr] = (int)(0x80000000u | (unsigned)J);
                dcount += __popcll(br);
            }
            const int t1 = L.lvl_blk[l + 1];
            for (int t0 = L.lvl_blk[l]; t0 < t1; t0 += 64) {
                const int t = t0 + lane;
                const int code = t < t1 ? L.otask[t] : 0;
                const bool d = t < t1 && block_is_presummed<W>(L, (code >> 16) & 511, code & 65535);
                const u64 bd = __ballot(d);
                const int pd = dcount + __popcll(bd & below);
                if (d && pd < L.dense_cap) L.dense[pd] = code & 0x01FFFFFF;   // (without the count bits: bit 31 marks right-hand sides)
                dcount += __popcll(bd);
            }
        }
        if (lane == 0) L.dense_off[nlev] = dcount;
        if (dcount > L.dense_cap) L.dense_ok = 0;
        sync_<SOLO>();
    }
    // root supernode: the trailing run of single-column levels whose columns hold ALL later poses (a clique: the ten unknown
    // anchors of BASELINE config 4 — and the last tag pose, which the ordering eliminates among them —, the top separator of a
    // nested dissection), at most ROOT_MAX of them, none with a pushed child (whose update arrives by the reduction step)
    {
        int lr = nlev;
        if (L.root != nullptr && L.dense_ok) {
            while (lr > 0 && nlev - (lr - 1) <= ROOT_MAX) {
                const int l = lr - 1;
                if (L.lvl_col[l + 1] - L.lvl_col[l] != 1) break;
                const int J = L.lvl_col[l];
                int cnt = 0;
#pragma unroll
                for (int w = 0; w < W; ++w) cnt += __popcll(L.colmask[J * W + w]);
                if (cnt != nv - 1 - J) break;
                lr = l;
            }
            const int jr = lr < nlev ? L.lvl_col[lr] : nv, m = nv - jr;
            bool good = m >= 2;
            for (int a = 0; a < m && good; ++a) {
                u64 pushed_children = 0;
#pragma unroll
                for (int w = 0; w < W; ++w) pushed_children |= L.rowmask[(jr + a) * W + w] & below_word<W>(jr, w) & L.pushw[w];
                good = pushed_children == 0;
            }
            if (!good) lr = nlev;
        }
        L.root_level = lr;
    }
    LOCAMD_STAMP5(1);
    return nb;
}

// Once per solve (the topology does not change between iterations): the incidence lists, and the list of binary edges
// that share their pair of poses with another edge.  Everything wave-parallel:
//   incidence lists   one lane per pose walks the edge tables (uniform, broadcast loads) twice: count, then fill in fold
//                     order (ranges, priors, SE3; ascending edge number) — the order every sum over a pose's edges uses;
//   shared pairs      every binary edge stamps its number on the first entry of its off-diagonal block (H is not in use
//                     yet) and clears the second; an edge that does not read its own stamp back raises the pair's flag
//                     (second entry = -1); the edges of flagged pairs then take the minimum of their stamps (atomic min on
//                     the bit patterns of positive doubles), which names the pair's first edge; the flagged edges are
//                     listed in fold order by ballot compaction.
template <bool SP, bool SOLO>
__device__ __forceinline__ void compute_incidence(const Lds& L, int lane, int nv, int nr, int np, int ns) {
    LOCAMD_T5_DECL;
    for (int v0 = 0; v0 < nv; v0 += 64) {
        const int v = v0 + lane;
        int cnt = 0;
        for (int e = 0; e < nr; ++e) cnt += (L.r_idx[2 * e] == v) + (L.r_idx[2 * e + 1] == v);
        for (int e = 0; e < np; ++e) cnt += (L.p_idx[e] == v);
        for (int e = 0; e < ns; ++e) cnt += (L.s_idx[4 * e] == v) + (L.s_idx[4 * e + 1] == v);
        if (v < nv) L.ioff[v + 1] = cnt;
    }
    sync_<SOLO>();
    int carry = 0;
    for (int v0 = 0; v0 < nv; v0 += 64) {
        const int v = v0 + lane;
        const int c = v < nv ? L.ioff[v + 1] : 0;
        const int ex = wave_excl_scan_i(c, lane) + carry;
        carry = __shfl(ex + c, 63, 64);
        sync_<SOLO>();
        if (v < nv) L.ioff[v] = ex;
        sync_<SOLO>();
    }
    if (lane == 0) L.ioff[nv] = carry;
    sync_<SOLO>();
    for (int v0 = 0; v0 < nv; v0 += 64) {
        const int v = v0 + lane;
        int cur = v < nv ? L.ioff[v] : 0;
        for (int e = 0; e < nr; ++e) {
            if (L.r_idx[2 * e] == v) L.ilist[cur++] = e;
            if (L.r_idx[2 * e + 1] == v) L.ilist[cur++] = (1 << INC_ROLE_SHIFT) | e;
        }
        for (int e = 0; e < np; ++e) if (L.p_idx[e] == v) L.ilist[cur++] = (1 << INC_KIND_SHIFT) | e;
        for (int e = 0; e < ns; ++e) {
            if (L.s_idx[4 * e] == v) L.ilist[cur++] = (2 << INC_KIND_SHIFT) | e;
            if (L.s_idx[4 * e + 1] == v) L.ilist[cur++] = (2 << INC_KIND_SHIFT) | (1 << INC_ROLE_SHIFT) | e;
        }
    }
    sync_<SOLO>();
    auto block_entry = [&](int t) {
        const bool is_r = t < nr;
        const int va = is_r ? L.r_idx[2 * t] : L.s_idx[4 * (t - nr)], vb = is_r ? L.r_idx[2 * t + 1] : L.s_idx[4 * (t - nr) + 1];
        return vb >= 0 ? blk_off<SP>(L, max(va, vb), min(va, vb)) : -1;
    };
    const int nbin = nr + ns;
    for (int t = lane; t < nbin; t += 64) { const int a0 = block_entry(t); if (a0 >= 0) { L.Hs[a0] = (double)(t + 1); L.Hs[a0 + 1] = 0.0; } }
    sync_<SOLO>();
    for (int t = lane; t < nbin; t += 64) { const int a0 = block_entry(t); if (a0 >= 0 && L.Hs[a0] != (double)(t + 1)) L.Hs[a0 + 1] = -1.0; }
    sync_<SOLO>();
    for (int t = lane; t < nbin; t += 64) {
        const int a0 = block_entry(t);
        if (a0 >= 0 && L.Hs[a0 + 1] == -1.0) atomicMin(reinterpret_cast<u64*>(L.Hs + a0), (u64)__double_as_longlong((double)(t + 1)));
    }
    sync_<SOLO>();
    int cnt = 0;
    for (int t0 = 0; t0 < nbin; t0 += 64) {
        const int t = t0 + lane;
        const int a0 = t < nbin ? block_entry(t) : -1;
        const bool sh = a0 >= 0 && L.Hs[a0 + 1] == -1.0;
        const bool first = sh && L.Hs[a0] == (double)(t + 1);
        const u64 bal = __ballot(sh);
        if (sh) L.shared[1 + cnt + __popcll(bal & ((1ull << lane) - 1))] = (first ? (1 << INC_KIND_SHIFT) : 0) | t;
        if (t >= nr && t < nbin) L.s_idx[4 * (t - nr) + 3] = sh ? 1 : 0;   // (evaluate_edges: who writes its block straight into H)
        cnt += __popcll(bal);
    }
    if (lane == 0) L.shared[0] = cnt;
    sync_<SOLO>();
    // the two marker entries of every edge block go back to zero: H is zeroed once per launch BEFORE this pass, and nothing may
    // depend on every later linearisation overwriting all 36 entries of every block that has an edge
    for (int t = lane; t < nbin; t += 64) { const int a0 = block_entry(t); if (a0 >= 0) { L.Hs[a0] = 0.0; L.Hs[a0 + 1] = 0.0; } }
    sync_<SOLO>();
    LOCAMD_STAMP5(2);
}

// The incidence lists of a window that runs on several waves (NW > 1; called by ALL threads, real barriers): edge-parallel.
// One lane per pose walking every edge table twice (compute_incidence) is 2 x (poses / 64) x edges dependent look-ups on one
// wave — 15 % of a BASELINE config 4 solve (2815 range edges, 266 poses, the edge tables in the HBM workspace).  Here
//   counts      every edge bumps its endpoints' counters (LDS atomics), wave 0 scans them;
//   filling     every edge takes the next free slot of its endpoints' lists (atomic cursor: arbitrary order) in a scratch copy
//               (the edge-record region, unused until the first linearisation) and notes the owner next to it;
//   fold order  every entry ranks itself inside its own list by (kind, edge number) — exactly the order the one-lane-per-pose
//               walk produces — and moves to its place: the sums over a pose's edges stay bit-reproducible.
template <bool SP, int NW>
__device__ __forceinline__ void compute_incidence_wide(const Lds& L, int tid, int nv, int nr, int np, int ns) {
    constexpr int NT = 64 * NW;
    LOCAMD_T5_DECL;
    const int lane = tid;   // (the timing macros)
    int* cur = reinterpret_cast<int*>(L.scr);        // nv cursors (the ordering's scratch is free)
    int* tmp = reinterpret_cast<int*>(L.rrec);       // [ninc] codes, then [ninc] owners
    for (int v = tid; v <= nv; v += NT) L.ioff[v] = 0;
    __syncthreads();
    for (int e = tid; e < nr; e += NT) {
        atomicAdd(&L.ioff[L.r_idx[2 * e] + 1], 1);
        const int v1 = L.r_idx[2 * e + 1];
        if (v1 >= 0) atomicAdd(&L.ioff[v1 + 1], 1);
    }
    for (int e = tid; e < np; e += NT) atomicAdd(&L.ioff[L.p_idx[e] + 1], 1);
    for (int e = tid; e < ns; e += NT) { atomicAdd(&L.ioff[L.s_idx[4 * e] + 1], 1); atomicAdd(&L.ioff[L.s_idx[4 * e + 1] + 1], 1); }
    __syncthreads();
    if (tid < 64) {   // exclusive scan, wave 0
        int carry = 0;
        for (int v0 = 0; v0 < nv; v0 += 64) {
            const int v = v0 + tid;
            const int c = v < nv ? L.ioff[v + 1] : 0;
            const int ex = wave_excl_scan_i(c, tid) + carry;
            carry = __shfl(ex + c, 63, 64);
            sync_<true>();
            if (v < nv) { L.ioff[v] = ex; cur[v] = ex; }
            sync_<true>();
        }
        if (tid == 0) L.ioff[nv] = carry;
    }
    __syncthreads();
    const int ninc = L.ioff[nv];
    auto put = [&](int v, int code) { const int p = atomicAdd(&cur[v], 1); tmp[p] = code; tmp[ninc + p] = v; };
    for (int e = tid; e < nr; e += NT) {
        put(L.r_idx[2 * e], e);
        const int v1 = L.r_idx[2 * e + 1];
        if (v1 >= 0) put(v1, (1 << INC_ROLE_SHIFT) | e);
    }
    for (int e = tid; e < np; e += NT) put(L.p_idx[e], (1 << INC_KIND_SHIFT) | e);
    for (int e = tid; e < ns; e += NT) { put(L.s_idx[4 * e], (2 << INC_KIND_SHIFT) | e); put(L.s_idx[4 * e + 1], (2 << INC_KIND_SHIFT) | (1 << INC_ROLE_SHIFT) | e); }
    __syncthreads();
    constexpr int KEY = ~(1 << INC_ROLE_SHIFT);
    for (int p = tid; p < ninc; p += NT) {
        const int code = tmp[p], v = tmp[ninc + p];
        const int key = code & KEY;
        const int q0 = L.ioff[v], q1 = L.ioff[v + 1];
        int rank = 0;
        for (int q = q0; q < q1; ++q) rank += (tmp[q] & KEY) < key;
        L.ilist[q0 + rank] = code;
    }
    __syncthreads();
    LOCAMD_STAMP5(2);
    // binary edges that share their pair of poses with another edge (see compute_incidence)
    auto block_entry = [&](int t) {
        const bool is_r = t < nr;
        const int va = is_r ? L.r_idx[2 * t] : L.s_idx[4 * (t - nr)], vb = is_r ? L.r_idx[2 * t + 1] : L.s_idx[4 * (t - nr) + 1];
        return vb >= 0 ? blk_off<SP>(L, max(va, vb), min(va, vb)) : -1;
    };
    const int nbin = nr + ns;
    for (int t = tid; t < nbin; t += NT) { const int a0 = block_entry(t); if (a0 >= 0) { L.Hs[a0] = (double)(t + 1); L.Hs[a0 + 1] = 0.0; } }
    __syncthreads();
    for (int t = tid; t < nbin; t += NT) { const int a0 = block_entry(t); if (a0 >= 0 && L.Hs[a0] != (double)(t + 1)) L.Hs[a0 + 1] = -1.0; }
    __syncthreads();
    for (int t = tid; t < nbin; t += NT) {
        const int a0 = block_entry(t);
        if (a0 >= 0 && L.Hs[a0 + 1] == -1.0) atomicMin(reinterpret_cast<u64*>(L.Hs + a0), (u64)__double_as_longlong((double)(t + 1)));
    }
    __syncthreads();
    if (tid < 64) {   // compaction in fold order, wave 0
        int cnt = 0;
        for (int t0 = 0; t0 < nbin; t0 += 64) {
            const int t = t0 + tid;
            const int a0 = t < nbin ? block_entry(t) : -1;
            const bool sh = a0 >= 0 && L.Hs[a0 + 1] == -1.0;
            const bool first = sh && L.Hs[a0] == (double)(t + 1);
            const u64 bal = __ballot(sh);
            if (sh) L.shared[1 + cnt + __popcll(bal & ((1ull << tid) - 1))] = (first ? (1 << INC_KIND_SHIFT) : 0) | t;
            if (t >= nr && t < nbin) L.s_idx[4 * (t - nr) + 3] = sh ? 1 : 0;
            cnt += __popcll(bal);
        }
        if (tid == 0) L.shared[0] = cnt;
    }
    __syncthreads();
    // (markers back to zero, as in compute_incidence)
    for (int t = tid; t < nbin; t += NT) { const int a0 = block_entry(t); if (a0 >= 0) { L.Hs[a0] = 0.0; L.Hs[a0 + 1] = 0.0; } }
    __syncthreads();
}

// Fold the edge records into H (skyline lower triangle) and b: H_vv = sum J_v^T (rho' Omega) J_v etc., every entry summed
// over its edges in one fixed order (ranges, priors, SE3; bit-reproducible, no atomics).  It is a gather: task
// (pose v, entry) walks v's incidence list and sums that entry of the diagonal block / of b in a register, one store at
// the end; an off-diagonal block that belongs to one edge is written by it; the few pairs of poses with several edges
// (a key-frame pose edge landing on the previous pose next to the smoothness edge) are accumulated edge by edge at the
// end.  No read-modify-write chains through memory for the bulk, all tasks independent.
template <bool SP, int NW>
__device__ __forceinline__ void build_system(const Lds& L, int lane, int n, int nr, int ns) {
    constexpr int NT = 64 * NW;
    const int nv = n / 6;
    // (H was zeroed before the linearisation, which has already written the off-diagonal blocks of the unshared SE3 edges)
    // diagonal blocks and b: one lane per pose, its 21 + 6 sums in registers, ONE pass over the pose's incidence list — the
    // 27 loads of an incidence are independent of each other, so a list costs one memory round trip per edge, not per entry
    for (int v = lane; v < nv; v += NT) {
        double acc[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) acc[k] = 0.0;
        const int p1 = L.ioff[v + 1];
        for (int p = L.ioff[v]; p < p1; ++p) {
            const int code = L.ilist[p];
            const int kind = code >> INC_KIND_SHIFT, role = (code >> INC_ROLE_SHIFT) & 1, e = code & INC_EDGE_MASK;
            if (kind == 0) {
                const double* rec = L.rrec + e * RREC;
                double J[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) J[i] = rec[6 * role + i];
                const double wr = rec[12], wre = rec[13];
#pragma unroll
                for (int r = 0; r < 6; ++r) {
#pragma unroll
                    for (int cc = 0; cc <= r; ++cc) acc[r * (r + 1) / 2 + cc] += wr * J[r] * J[cc];
                    acc[21 + r] += J[r] * wre;
                }
            } else {
                const double* rec = kind == 1 ? L.prec + e * PREC : L.srec + e * SREC + S_HJJ * role;
                const double* recb = kind == 1 ? L.prec + e * PREC + 21 : L.srec + e * SREC + S_BI + 6 * role;
#pragma unroll
                for (int k = 0; k < 21; ++k) acc[k] += rec[k];
#pragma unroll
                for (int r = 0; r < 6; ++r) acc[21 + r] += recb[r];
            }
        }
        const int dv = L.boff[v + 1] - 36;  // the diagonal block is the last one of its block row (both storage schemes)
#pragma unroll
        for (int r = 0; r < 6; ++r) {
#pragma unroll
            for (int cc = 0; cc <= r; ++cc) L.Hs[dv + 6 * cc + r] = acc[r * (r + 1) / 2 + cc];
            L.b[v * 6 + r] = acc[21 + r];
        }
    }
    // off-diagonal blocks: one task per (binary edge, entry q = 6 cc + r of the block, column-major like the storage)
    auto offdiag = [&](int t, int q, int& addr) {
        double h;
        if (t < nr) {
            const int r = q % 6, cc = q / 6;
            const int v0 = L.r_idx[2 * t], v1 = L.r_idx[2 * t + 1];
            const double* rec = L.rrec + t * RREC;
            const double wr = rec[12];
            if (v0 > v1) { h = wr * rec[r] * rec[6 + cc]; addr = blk_off<SP>(L, v0, v1) + q; }
            else         { h = wr * rec[6 + r] * rec[cc]; addr = blk_off<SP>(L, v1, v0) + q; }
        } else {
            const int e = t - nr;
            const int vi = L.s_idx[4 * e], vj = L.s_idx[4 * e + 1];
            h = L.srec[e * SREC + S_OFF + q];
            addr = (vi > vj ? blk_off<SP>(L, vi, vj) : blk_off<SP>(L, vj, vi)) + q;
        }
        return h;
    };
    // one lane per binary edge writes the edge's whole 6x6 block (13 loads for a range edge, then 36 stores): the index
    // look-ups and the record are read once per edge, not once per entry
    for (int t = lane; t < nr + ns; t += NT) {
        if (t < nr ? L.r_idx[2 * t + 1] < 0 : L.s_idx[4 * (t - nr) + 3] == 0) continue;  // range to a fixed anchor: no block; unshared SE3 edge: written already
        if (t < nr) {
            const int v0 = L.r_idx[2 * t], v1 = L.r_idx[2 * t + 1];
            const double* rec = L.rrec + t * RREC;
            double Jr[6], Jc[6];   // rows: the later-labelled pose, columns: the earlier one
            const bool swap = v0 <= v1;
#pragma unroll
            for (int k = 0; k < 6; ++k) { Jr[k] = rec[swap ? 6 + k : k]; Jc[k] = rec[swap ? k : 6 + k]; }
            const double wr = rec[12];
            double* dst = L.Hs + (swap ? blk_off<SP>(L, v1, v0) : blk_off<SP>(L, v0, v1));
#pragma unroll
            for (int cc = 0; cc < 6; ++cc)
#pragma unroll
                for (int r = 0; r < 6; ++r) dst[6 * cc + r] = wr * Jr[r] * Jc[cc];
        } else {
            const int e = t - nr;
            const int vi = L.s_idx[4 * e], vj = L.s_idx[4 * e + 1];
            const double* src = L.srec + e * SREC + S_OFF;
            double* dst = L.Hs + (vi > vj ? blk_off<SP>(L, vi, vj) : blk_off<SP>(L, vj, vi));
#pragma unroll
            for (int q = 0; q < 36; ++q) dst[q] = src[q];
        }
    }
    __syncthreads();
    // pairs of poses with several edges: their blocks are accumulated edge by edge, in fold order
    const int n_shared = L.shared[0];
    for (int i = 1; i <= n_shared; ++i) {
        const int code = L.shared[i];
        const int t = code & INC_EDGE_MASK;
        const bool first = (code >> INC_KIND_SHIFT) != 0;
        if (lane < 36) {
            int addr;
            const double h = offdiag(t, lane, addr);
            L.Hs[addr] = first ? h : L.Hs[addr] + h;
        }
    }
    __syncthreads();
}

// SKYLINE path: (H + lambda I) x = b in one sweep over 6x6 block columns (poses are 6-DoF blocks, n = 6 nv):
//   per block column J the ACTIVE rows are those of blocks J..last[J] whose envelope reaches J, plus the right-hand side
//   (it rides along as one more row of the factor: forward substitution is one more Cholesky row).  In chunks of 64:
//   (a) every lane forms the 6-entry segment S_i = H[i][J] - sum_K L[i][K] L[J][K]^T of its row (K over the band),
//   (b) the six diagonal rows publish theirs (6x6 in LDS) and every lane factors that block in registers (redundantly:
//       no broadcast of the factor), pivots from the rsq seed,
//   (c) every lane finishes its row with a 6x6 triangular solve and stores 6 entries.
//   Back-substitution walks block rows with y in LDS.  Inverse pivots go to diagL, x to L.x.
__device__ __forceinline__ bool factor_and_solve(const Lds& L, int lane, int n, double lambda) {
    const int nvb = n / 6;
    for (int J = 0; J < nvb; ++J) {
        const int c0 = 6 * J;
        const int fJ = L.fb[J];
        const int nrows = 6 * (L.last[J] - J + 1);  // matrix rows c0 .. c0 + nrows - 1, then the rhs
        const int total = nrows + 1;
        double G[6][6], ig[6];
        for (int base = 0; base < total; base += 64) {
            const int idx = base + lane;
            const bool is_rhs = idx == nrows;
            const int row = is_rhs ? n : c0 + idx;
            const int fbi = (is_rhs || idx > nrows) ? 0 : L.fb[row / 6];
            const bool part = (idx < total) && (is_rhs || fbi <= J);
            // a matrix row's base and stride: entry (row, col) at roff + rs * col (the rhs row is a plain vector: stride 1)
            const int roff = part && !is_rhs ? L.boff[row / 6] - 36 * fbi + row % 6 : 0;
            const int rs = is_rhs ? 1 : 6;
            const int jb = L.boff[J] - 36 * fJ;  // block row J: entry (c0 + c, col) at jb + 6 col + c
            double S[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) S[c] = 0.0;
            if (part) {
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const int col = c0 + c;
                    double v;
                    if (is_rhs) v = L.b[col];
                    else if (col <= row) v = L.Hs[roff + 6 * col];
                    else v = L.Hs[jb + 6 * row + c];  // symmetric entry inside the diagonal block
                    if (row == col) v += lambda;
                    S[c] = v;
                }
                const double* ri = is_rhs ? L.yrow : L.Ls;
                for (int K = (fJ > fbi ? fJ : fbi); K < J; ++K) {
                    double li[6];
#pragma unroll
                    for (int k = 0; k < 6; ++k) li[k] = ri[roff + rs * (6 * K + k)];
                    const double* rj = L.Ls + jb + 36 * K;  // the 6x6 block (rows c0.., columns 6K..) is contiguous: [k][c]
#pragma unroll
                    for (int c = 0; c < 6; ++c) {
                        double acc = 0.0;
#pragma unroll
                        for (int k = 0; k < 6; ++k) acc = __builtin_fma(li[k], rj[6 * k + c], acc);
                        S[c] -= acc;
                    }
                }
            }
            if (base == 0) {
                // (b) publish the diagonal block (rows idx 0..5), factor it everywhere
                if (idx < 6) {
#pragma unroll
                    for (int c = 0; c < 6; ++c) L.blk[idx * 6 + c] = S[c];
                }
                __syncthreads();
                bool ok = true;
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    double dj = L.blk[j * 6 + j];
#pragma unroll
                    for (int k = 0; k < j; ++k) dj = __builtin_fma(-G[j][k], G[j][k], dj);
                    ok = ok && (dj > 0.0) && (dj < DBL_MAX);
                    double g, igj;
                    sqrt_and_rsqrt(fmax(dj, 1e-300), g, igj);
                    igj = __builtin_fma(igj, __builtin_fma(-g, igj, 1.0), igj);  // one Newton step: 1/g to ~1e-16
                    G[j][j] = g;
                    ig[j] = igj;
#pragma unroll
                    for (int i2 = j + 1; i2 < 6; ++i2) {
                        double v = L.blk[i2 * 6 + j];
#pragma unroll
                        for (int k = 0; k < j; ++k) v = __builtin_fma(-G[i2][k], G[j][k], v);
                        G[i2][j] = v * ig[j];
                    }
                }
                if (!ok) return false;  // uniform: every lane factored the same block
            }
            // (c) finish the rows
            if (part && (is_rhs || idx >= 6)) {
                double x[6];
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    double v = S[c];
#pragma unroll
                    for (int k = 0; k < c; ++k) v = __builtin_fma(-x[k], G[c][k], v);
                    x[c] = v * ig[c];
                }
                double* dst = is_rhs ? L.yrow : L.Ls;
#pragma unroll
                for (int c = 0; c < 6; ++c) dst[roff + rs * (c0 + c)] = x[c];
            } else if (base == 0 && idx < 6) {
                // (static indices only: a runtime row index would push G into scratch memory)
#pragma unroll
                for (int rr = 0; rr < 6; ++rr) {
                    if (idx == rr) {
#pragma unroll
                        for (int c = 0; c < rr; ++c) L.Ls[roff + 6 * (c0 + c)] = G[rr][c];
                        L.diagL[row] = ig[rr];  // the INVERSE pivot: back-substitution multiplies
                    }
                }
            }
        }
        __syncthreads();
    }
    LOCAMD_TIC();
    // back substitution, block rows from the bottom; y (= row n of the factor) is updated in place in LDS
    for (int J = nvb - 1; J >= 0; --J) {
        const int c0 = 6 * J;
        const int fJ6 = 6 * L.fb[J];
        const int jb = L.boff[J] - 6 * fJ6;  // block row J: entry (c0 + c, col) at jb + 6 col + c
        double yj[6], x[6], G[6][6], igd[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) yj[c] = L.yrow[c0 + c];
#pragma unroll
        for (int rr = 0; rr < 6; ++rr) {
            igd[rr] = L.diagL[c0 + rr];
#pragma unroll
            for (int c = 0; c < rr; ++c) G[rr][c] = L.Ls[jb + 6 * (c0 + c) + rr];
        }
#pragma unroll
        for (int rr = 5; rr >= 0; --rr) {
            double v = yj[rr];
#pragma unroll
            for (int s2 = rr + 1; s2 < 6; ++s2) v = __builtin_fma(-G[s2][rr], x[s2], v);
            x[rr] = v * igd[rr];
        }
        if (lane < 6) {
#pragma unroll
            for (int c = 0; c < 6; ++c) if (lane == c) L.x[c0 + c] = x[c];
        }
        for (int k = fJ6 + lane; k < c0; k += 64) {
            double acc = L.yrow[k];
#pragma unroll
            for (int c = 0; c < 6; ++c) acc = __builtin_fma(-L.Ls[jb + 6 * k + c], x[c], acc);
            L.yrow[k] = acc;
        }
        __syncthreads();
    }
    LOCAMD_TOC(5);
    return true;
}

// The root supernode (see compute_sparse_mw): the trailing clique of m <= ROOT_MAX poses as ONE dense (6 m) x (6 m) system in
// LDS — S = H_root + lambda I - sum over the columns BEFORE the root of L L^T (the cooperative sums, one wave per block, all
// blocks in one batch), a right-looking Cholesky with every thread of the workgroup on the trailing update (the right-hand
// side rides along as one more row), the back-substitution, x_root.  As eleven sequential levels of one column each (their
// sums, a diagonal-row phase, a row-task phase, three barriers and a handful of lanes per level) the ten anchors of BASELINE
// config 4 were a third of its solve.
template <int W, int NW>
__device__ __forceinline__ bool root_factor_and_solve(const Lds& L, int tid, int nv, double lambda) {
    constexpr int NT = 64 * NW;
    const int jr = L.lvl_col[L.root_level], m = nv - jr, n = 6 * m;
    double* S = L.root;   // rows 0 .. n - 1: the lower triangle; row n: the right-hand side; row-major, n per row
    const int npair = m * (m + 1) / 2, nent = npair + m;
    auto pair_of = [&](int e, int& a, int& b) { a = 0; while ((a + 1) * (a + 2) / 2 <= e) ++a; b = e - a * (a + 1) / 2; };
    const int lane = tid;   // (the timing macros)
    (void)lane; (void)nent; (void)pair_of;
    LOCAMD_TIC();
    // S = H_root + lambda I (lower triangle), right-hand side row = b_root
    for (int q = tid; q < npair * 36 + n; q += NT) {
        if (q < npair * 36) {
            int a, b;
            pair_of(q / 36, a, b);
            const int k = q % 36, r = k % 6, c = k / 6;
            if (a == b && r < c) continue;
            double v = L.Hs[blk_off<true>(L, jr + a, jr + b) + k];
            if (a == b && r == c) v += lambda;
            S[(6 * a + r) * n + 6 * b + c] = v;
        } else {
            const int k = q - npair * 36;
            S[n * n + k] = L.b[6 * jr + k];
        }
    }
    // U = sum over the columns K before the root of M_K M_K^T, M_K = the (n + 1) x 6 slab [L_aK (a in the root); y_K^T]: a
    // (n + 1) x (6 jr) x (n + 1) GEMM — on the f64 matrix cores.  v_mfma_f64_16x16x4_f64 (layout found with
    // tools/mfma_f64_probe.hip): lane l supplies A[l % 16][l / 16] and B[l / 16][l % 16], and holds D[(l / 16) + 4 v][l % 16],
    // v = 0 .. 3.  The rows are cut into four tiles of 16; a tile's A fragment IS its B fragment (U is M M^T), so a (K, slice
    // of four k) costs four gathers and ten MFMAs for the lower triangle of tiles.  The waves split the columns K and subtract
    // their partial U from S one after the other (wave order: bit-reproducible).
    {
        typedef double v4f64 __attribute__((ext_vector_type(4)));
        const int wv = tid >> 6, l = tid & 63, li = l & 15, lk = l >> 4;
        int frow[4], foff[4];   // per tile: the root pose's row of blocks (or -1: right-hand side / -2: padding), the row inside the block
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            const int R = 16 * ti + li;
            frow[ti] = R < n ? jr + R / 6 : (R == n ? -1 : -2);
            foff[ti] = R < n ? R % 6 : 0;
        }
        v4f64 acc[10];
#pragma unroll
        for (int t = 0; t < 10; ++t) acc[t] = v4f64{0.0, 0.0, 0.0, 0.0};
        constexpr int KU = 4;   // columns per step: their 32 gathers are in flight together (a step is one memory round trip)
        for (int K0 = KU * wv; K0 < jr; K0 += KU * NW) {
            int boffs[KU][4];   // offset of block (root pose, K) + row; -2: the right-hand side's y_K; -1: nothing
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const int K = K0 + u;
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) {
                    const int a = frow[ti];
                    boffs[u][ti] = -1;
                    if (K < jr) {
                        if (a >= 0) {
                            if ((L.rowmask[a * W + (K >> 6)] >> (K & 63)) & 1ull) boffs[u][ti] = L.boff[a] + 36 * row_rank<W>(L, a, K) + foff[ti];
                        } else if (a == -1) boffs[u][ti] = -2;
                    }
                }
            }
            double f[KU][2][4];
#pragma unroll
            for (int u = 0; u < KU; ++u)
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    const int k = 4 * sl + lk;
#pragma unroll
                    for (int ti = 0; ti < 4; ++ti) {
                        f[u][sl][ti] = 0.0;
                        if (k < 6) {
                            if (boffs[u][ti] >= 0) f[u][sl][ti] = L.Ls[boffs[u][ti] + 6 * k];
                            else if (boffs[u][ti] == -2) f[u][sl][ti] = L.yrow[6 * (K0 + u) + k];
                        }
                    }
                }
#pragma unroll
            for (int u = 0; u < KU; ++u)
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    int t = 0;
#pragma unroll
                    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
                        for (int tj = 0; tj <= ti; ++tj) { acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[u][sl][ti], f[u][sl][tj], acc[t], 0, 0, 0); ++t; }
                }
        }
        __syncthreads();   // (S is assembled)
        for (int turn = 0; turn < NW; ++turn) {
            if (wv == turn) {
                int t = 0;
#pragma unroll
                for (int ti = 0; ti < 4; ++ti)
#pragma unroll
                    for (int tj = 0; tj <= ti; ++tj) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const int R = 16 * ti + lk + 4 * v, C = 16 * tj + li;
                            if (C <= R && R <= n && C < n) S[R * n + C] -= acc[t][v];
                        }
                        ++t;
                    }
            }
            __syncthreads();
        }
    }
    LOCAMD_TOC4(0);
    bool ok = true;
    // blocked right-looking Cholesky, 6 columns per step, two barriers per step: every thread factors the step's diagonal
    // block itself (registers), the rows below it (and the right-hand side, row n) are solved one per thread, the trailing
    // update is a wave per row and a lane per column
    for (int bj = 0; bj < m; ++bj) {
        const int j0 = 6 * bj;
        double A[6][6], ig[6];
#pragma unroll
        for (int c = 0; c < 6; ++c)
#pragma unroll
            for (int r = c; r < 6; ++r) A[r][c] = S[(j0 + r) * n + j0 + c];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const double g = pivot_rsqrt(A[j][j]);
            ig[j] = g;
#pragma unroll
            for (int i2 = j + 1; i2 < 6; ++i2) A[i2][j] *= g;
#pragma unroll
            for (int i2 = j + 1; i2 < 6; ++i2)
#pragma unroll
                for (int c = j + 1; c <= i2; ++c) A[i2][c] = __builtin_fma(-A[i2][j], A[c][j], A[i2][c]);
        }
        ok = ok && (((ig[0] + ig[1]) + (ig[2] + ig[3])) + (ig[4] + ig[5]) < DBL_MAX);   // (a pivot <= 0 or not finite: NaN / inf)
        const int i = j0 + 6 + tid;   // rows below the block, the right-hand side last
        if (i <= n) {
            double sv[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) sv[c] = S[i * n + j0 + c];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                sv[c] *= ig[c];
#pragma unroll
                for (int c2 = c + 1; c2 < 6; ++c2) sv[c2] = __builtin_fma(-sv[c], A[c2][c], sv[c2]);
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) S[i * n + j0 + c] = sv[c];
        }
        __syncthreads();
        if (tid == 0) {   // (nobody reads the diagonal block any more: its factor and the inverse pivots, for the back-substitution)
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                S[(j0 + c) * n + j0 + c] = ig[c];
#pragma unroll
                for (int r = c + 1; r < 6; ++r) S[(j0 + r) * n + j0 + c] = A[r][c];
            }
        }
        const int c = j0 + 6 + (tid & 63);
        for (int i2 = j0 + 6 + (tid >> 6); i2 <= n; i2 += NW) {
            if (c <= i2 && c < n) {
                double acc = S[i2 * n + c];
#pragma unroll
                for (int k = 0; k < 6; ++k) acc = __builtin_fma(-S[i2 * n + j0 + k], S[c * n + j0 + k], acc);
                S[i2 * n + c] = acc;
            }
        }
        __syncthreads();
    }
    if (block_any<NW>(!ok, L.red, tid)) return false;
    for (int j = n - 1; j >= 0; --j) {
        const double xj = S[n * n + j] * S[j * n + j];
        if (tid < j) S[n * n + tid] = __builtin_fma(-S[j * n + tid], xj, S[n * n + tid]);
        if (tid == 0) L.x[6 * jr + j] = xj;
        __syncthreads();
    }
    return true;
}

// SPARSE path: (H + lambda I) x = b, level by level over the elimination tree.  Per level (its block columns J are
// mutually independent):
//   phase 1  the six rows of every diagonal block form S_JJ = H_JJ + lambda I - sum_K L_JK L_JK^T and publish it RAW into
//            the upper triangle (+ diagonal) of L's (J, J) block — the Cholesky factor G_J of that block only needs the
//            strict lower triangle (its inverse pivots go to diagL), so both fit the 36 slots;
//   phase 2  one lane per row of every block of the level's columns (diagonal rows, off-diagonal rows, and the right-hand
//            side as one more row per column): factor G_J from the raw block in registers (redundantly: no broadcast of
//            the factor), then  diagonal rows store their row of G_J;  the others form their 6-entry segment
//            S = H_iJ - sum_{K in row i and row J} L_iK L_JK^T  and finish it with the 6x6 triangular solve.
//   Back-substitution walks the levels downwards, one lane per column (a gather over the column's blocks).
// Pushed updates: a column K with exactly one off-diagonal block (i, K) that is factored in column mode leaves
// U_K = L_iK L_iK^T (lower triangle, 21) and u_K = L_iK y_K (6) in Us[28 K ..]; its parent i then subtracts 27 numbers instead of
// re-forming the products from the 36-entry block (and, with the arrays in the HBM workspace, the sums over a parent's children
// are formed first by a wave-parallel reduction step — one lane per (parent, entry), four loads in flight — so a key pose with
// eight children costs two memory round trips, not eight).
template <bool GLOBAL_A, int W, int NW>
__device__ __forceinline__ bool factor_and_solve_sparse(const Lds& L, int lane, double lambda) {
    constexpr int NT = 64 * NW;   // (`lane` is the thread's index in its window's workgroup)
    // (8-word windows: the levels from root_level on are the root supernode, factored densely in LDS after the others)
    const int lr = (W > 1 && GLOBAL_A) ? L.root_level : L.nlev;
    for (int l = 0; l < lr; ++l) {
        const int c0 = L.lvl_col[l], ncol = L.lvl_col[l + 1] - c0;
        const int b0 = L.lvl_blk[l], nblk = L.lvl_blk[l + 1] - b0;
        if (GLOBAL_A) {
            // reduction step: As[J] = sum over J's pushed children K (ascending) of Us[K], for the columns of this level
            LOCAMD_TIC();
            bool any = false;
            for (int base = 0; base < 27 * ncol; base += NT) {
                const int idx = base + lane;
                if (idx < 27 * ncol) {
                    const int J = L.colorder[c0 + idx / 27], e = idx % 27;
                    bool mine = false;
                    double acc = 0.0;
                    for (int w = 0; w < W; ++w) {
                        u64 m = rm_word<W>(L, J, w) & below_word<W>(J, w) & push_word<W>(L, w);
                        mine = mine || m != 0;
                        while (m) {
                            // four children per round: the loads are independent, the sum stays in ascending order
                            double v[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                v[u] = 0.0;
                                if (m) { const int K = (w << 6) + __ffsll((long long)m) - 1; m &= m - 1; v[u] = L.Us[28 * K + e]; }
                            }
                            acc = ((acc + v[0]) + v[1]) + v[2];
                            acc += v[3];
                        }
                    }
                    if (mine) { any = true; L.As[28 * J + e] = acc; }
                }
            }
            if (NW > 1 || __ballot(any)) __syncthreads();
            LOCAMD_TOC(3);
        }
        if (level_is_column_mode<W>(L, l)) {
            // COLUMN MODE: one lane does the whole column — diagonal block, its Cholesky factor, the right-hand side, the
            // off-diagonal block(s) and the pushed update — so G_J is factored once and the level costs one pass, one barrier.
            LOCAMD_TIC();
            bool okw = true;
            for (int base = 0; base < ncol; base += NT) {
                const int idx = base + lane;
                if (idx < ncol) {
                    const int J = L.colorder[c0 + idx];
                    const int rowJ = L.boff[J], dJ = L.boff[J + 1] - 36;
#if defined(LOCAMD_WINDOW_TIMING) && LOCAMD_WINDOW_TIMING >= 2
                    long long locamd_ts = clock64();
#endif
                    double G[6][6], ig[6], y[6];
#pragma unroll
                    for (int r = 0; r < 6; ++r) {
#pragma unroll
                        for (int c = 0; c <= r; ++c) G[r][c] = L.Hs[dJ + 6 * c + r];
                        G[r][r] += lambda;
                        y[r] = L.b[6 * J + r];
                    }
                    if (GLOBAL_A && has_pushed_child<W>(L, J)) {
#pragma unroll
                        for (int r = 0; r < 6; ++r) {
#pragma unroll
                            for (int c = 0; c <= r; ++c) G[r][c] -= L.As[28 * J + r * (r + 1) / 2 + c];
                            y[r] -= L.As[28 * J + 21 + r];
                        }
                    }
                    LOCAMD_SUB(0);
                    int kb = -1;
                    for (int w = 0; w < W; ++w) {
                        u64 m = rm_word<W>(L, J, w) & below_word<W>(J, w);
                        while (m) {
                            const int K = (w << 6) + __ffsll((long long)m) - 1;
                            m &= m - 1;
                            ++kb;
                            if (pushed<W>(L, K)) {
                                if (!GLOBAL_A) {
#pragma unroll
                                    for (int r = 0; r < 6; ++r) {
#pragma unroll
                                        for (int c = 0; c <= r; ++c) G[r][c] -= L.Us[28 * K + r * (r + 1) / 2 + c];
                                        y[r] -= L.Us[28 * K + 21 + r];
                                    }
                                }
                                continue;
                            }
                            const double* blk = L.Ls + rowJ + 36 * kb;
                            double bl[36], yk[6];
#pragma unroll
                            for (int q = 0; q < 36; ++q) bl[q] = blk[q];
#pragma unroll
                            for (int k = 0; k < 6; ++k) yk[k] = L.yrow[6 * K + k];
#pragma unroll
                            for (int r = 0; r < 6; ++r) {
#pragma unroll
                                for (int c = 0; c <= r; ++c) {
                                    double acc = 0.0;
#pragma unroll
                                    for (int k = 0; k < 6; ++k) acc = __builtin_fma(bl[6 * k + r], bl[6 * k + c], acc);
                                    G[r][c] -= acc;
                                }
                                double acc = 0.0;
#pragma unroll
                                for (int k = 0; k < 6; ++k) acc = __builtin_fma(bl[6 * k + r], yk[k], acc);
                                y[r] -= acc;
                            }
                        }
                    }
                    LOCAMD_SUB(1);
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        double dj = G[j][j];
#pragma unroll
                        for (int k = 0; k < j; ++k) dj = __builtin_fma(-G[j][k], G[j][k], dj);
                        okw = okw && (dj > 0.0) && (dj < DBL_MAX);
                        double g, igj;
                        sqrt_and_rsqrt(fmax(dj, 1e-300), g, igj);
                        igj = __builtin_fma(igj, __builtin_fma(-g, igj, 1.0), igj);
                        ig[j] = igj;
#pragma unroll
                        for (int i2 = j + 1; i2 < 6; ++i2) {
                            double v = G[i2][j];
#pragma unroll
                            for (int k = 0; k < j; ++k) v = __builtin_fma(-G[i2][k], G[j][k], v);
                            G[i2][j] = v * igj;
                        }
                        L.diagL[6 * J + j] = igj;
                    }
#pragma unroll
                    for (int r = 1; r < 6; ++r)
#pragma unroll
                        for (int c = 0; c < r; ++c) L.Ls[dJ + 6 * c + r] = G[r][c];
#pragma unroll
                    for (int c = 0; c < 6; ++c) {
                        double v = y[c];
#pragma unroll
                        for (int k = 0; k < c; ++k) v = __builtin_fma(-y[k], G[c][k], v);
                        y[c] = v * ig[c];
                        L.yrow[6 * J + c] = y[c];
                    }
                    LOCAMD_SUB(2);
                    const bool push = pushed<W>(L, J);   // exactly one off-diagonal block
                    for (int wi = 0; wi < W; ++wi) {
                    u64 mi = cm_word<W>(L, J, wi);
                    while (mi) {
                        const int i = (wi << 6) + __ffsll((long long)mi) - 1;
                        mi &= mi - 1;
                        const int rowi = L.boff[i];
                        const int bi = rowi + 36 * row_rank<W>(L, i, J);
                        double S[36];
#pragma unroll
                        for (int q = 0; q < 36; ++q) S[q] = L.Hs[bi + q];
                        for (int w = 0; w < W; ++w) {
                        u64 m = rm_word<W>(L, i, w) & rm_word<W>(L, J, w) & below_word<W>(J, w);
                        while (m) {
                            const int K = (w << 6) + __ffsll((long long)m) - 1;
                            m &= m - 1;
                            const double* bki = L.Ls + rowi + 36 * row_rank<W>(L, i, K);
                            const double* bkj = L.Ls + rowJ + 36 * row_rank<W>(L, J, K);
                            for (int k = 0; k < 6; ++k) {
                                double li[6], lj[6];
#pragma unroll
                                for (int r = 0; r < 6; ++r) { li[r] = bki[6 * k + r]; lj[r] = bkj[6 * k + r]; }
#pragma unroll
                                for (int c = 0; c < 6; ++c)
#pragma unroll
                                    for (int r = 0; r < 6; ++r) S[6 * c + r] = __builtin_fma(-li[r], lj[c], S[6 * c + r]);
                            }
                        }
                        }
#pragma unroll
                        for (int r = 0; r < 6; ++r) {
#pragma unroll
                            for (int c = 0; c < 6; ++c) {
                                double v = S[6 * c + r];
#pragma unroll
                                for (int k = 0; k < c; ++k) v = __builtin_fma(-S[6 * k + r], G[c][k], v);
                                v *= ig[c];
                                S[6 * c + r] = v;          // S becomes X = L_iJ, entry (r, c)
                                L.Ls[bi + 6 * c + r] = v;
                            }
                        }
                        if (push) {
#pragma unroll
                            for (int r = 0; r < 6; ++r) {
#pragma unroll
                                for (int c = 0; c <= r; ++c) {
                                    double acc = 0.0;
#pragma unroll
                                    for (int k = 0; k < 6; ++k) acc = __builtin_fma(S[6 * k + r], S[6 * k + c], acc);
                                    L.Us[28 * J + r * (r + 1) / 2 + c] = acc;
                                }
                                double acc = 0.0;
#pragma unroll
                                for (int k = 0; k < 6; ++k) acc = __builtin_fma(S[6 * k + r], y[k], acc);
                                L.Us[28 * J + 21 + r] = acc;
                            }
                        }
                    }
                    }
                    LOCAMD_SUB(6);
                }
            }
            __syncthreads();
            LOCAMD_TOC(4);
            LOCAMD_TOC4(0);
            if (block_any<NW>(!okw, L.red, lane)) return false;
            continue;
        }
        LOCAMD_TIC();
        bool dense_level = false;
        (void)dense_level;
        if (W > 1 && L.dense_ok) {
            // dense blocks / right-hand sides of this level (listed once per solve): summed cooperatively first, one wave each
            const int d0 = L.dense_off[l], d1 = L.dense_off[l + 1];
            dense_level = d1 > d0;
            for (int e0 = d0; e0 < d1; e0 += NW) {
                const int e = e0 + (lane >> 6);
                if (e < d1) {
                    const int code = L.dense[e];
                    const int J = code & 65535, i = (code >> 16) & 511;
                    if (code < 0) presum_rhs<W>(L, lane & 63, J, J);
                    else presum_block<W>(L, lane & 63, i, J, J);
                }
            }
            if (dense_level) __syncthreads();
        }
        for (int base = 0; base < 6 * ncol; base += NT) {
            const int idx = base + lane;
            if (idx < 6 * ncol) {
                const int J = L.colorder[c0 + idx / 6], r = idx % 6;
                const int rowJ = L.boff[J], dJ = L.boff[J + 1] - 36;  // the diagonal block is the last one of its row
                double S[6];
#pragma unroll
                for (int c = 0; c < 6; ++c) S[c] = c <= r ? L.Hs[dJ + 6 * c + r] : 0.0;
                if (GLOBAL_A && has_pushed_child<W>(L, J)) {   // the pushed children's updates were summed by the reduction step
#pragma unroll
                    for (int c = 0; c < 6; ++c) if (c <= r) S[c] -= L.As[28 * J + r * (r + 1) / 2 + c];
                }
                const bool pre = block_is_presummed<W>(L, J, J);
                if (pre) {
#pragma unroll
                    for (int c = 0; c < 6; ++c) if (c <= r) S[c] -= L.Ls[dJ + 6 * c + r];
                }
                int kb = -1;
                for (int w = 0; w < W && !pre; ++w) {
                u64 mK = rm_word<W>(L, J, w) & below_word<W>(J, w);
                while (mK) {
                    const int K = (w << 6) + __ffsll((long long)mK) - 1;
                    mK &= mK - 1;
                    ++kb;
                    if (pushed<W>(L, K)) {
                        if (!GLOBAL_A) {
#pragma unroll
                            for (int c = 0; c < 6; ++c) if (c <= r) S[c] -= L.Us[28 * K + r * (r + 1) / 2 + c];
                        }
                        continue;
                    }
                    const double* blk = L.Ls + rowJ + 36 * kb;
                    double li[6], bj[36];   // (all loads first; the row's own entries by address: r is not a compile-time index)
#pragma unroll
                    for (int k = 0; k < 6; ++k) li[k] = blk[6 * k + r];
#pragma unroll
                    for (int q = 0; q < 36; ++q) bj[q] = W > 1 ? blk[q] : 0.0;   // (one-word windows: the products read in place, as measured best there)
#pragma unroll
                    for (int c = 0; c < 6; ++c) {
                        double acc = 0.0;
#pragma unroll
                        for (int k = 0; k < 6; ++k) acc = __builtin_fma(li[k], W > 1 ? bj[6 * k + c] : blk[6 * k + c], acc);
                        S[c] -= acc;
                    }
                }
                }
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    if (c == r) S[c] += lambda;
                    if (c <= r) L.Ls[dJ + 6 * r + c] = S[c];  // entry (r, c) in the transposed slot
                }
            }
        }
        __syncthreads();
        LOCAMD_TOC(3);
        LOCAMD_TOC4(dense_level ? 2 : 1);
        const int ntask = 7 * ncol + 6 * nblk;
        bool ok = true;
        const long long locamd_t1 = clock64();
        (void)locamd_t1;
        for (int base = 0; base < ntask; base += NT) {
            const int idx = base + lane;
            if (idx < ntask) {
                int kind, i, J, r, nc = -1;  // 0: row r of the diagonal block of J; 1: row r of block (i, J); 2: right-hand side of column J
                if (idx < 6 * ncol) { kind = 0; J = L.colorder[c0 + idx / 6]; i = J; r = idx % 6; }
                else if (idx < 6 * (ncol + nblk)) {
                    const int t = idx - 6 * ncol;
                    const int code = L.otask[b0 + t / 6];
                    kind = 1; i = (code >> 16) & 511; J = code & 65535; r = t % 6;
                    nc = W > 1 ? (int)((unsigned)code >> 25) : -1;   // (8-word windows: the common earlier columns were counted once)
                } else { kind = 2; J = L.colorder[c0 + idx - 6 * (ncol + nblk)]; i = J; r = 0; }
                const int rowJ = L.boff[J], dJ = L.boff[J + 1] - 36;
                // the row's own segment first (its loads overlap the factorisation below)
                double S[6];
#pragma unroll
                for (int c = 0; c < 6; ++c) S[c] = 0.0;
                int bi = 0;
                if (kind == 1) {
                    const int rowi = L.boff[i];
                    bi = rowi + 36 * row_rank<W>(L, i, J);
#pragma unroll
                    for (int c = 0; c < 6; ++c) S[c] = L.Hs[bi + 6 * c + r];
                    const bool pre = nc >= 0 ? (L.dense_ok && nc >= DENSE_K) : block_is_presummed<W>(L, i, J);
                    if (pre) {
#pragma unroll
                        for (int c = 0; c < 6; ++c) S[c] -= L.Ls[bi + 6 * c + r];
                    }
                    int left = nc >= 0 && nc < 127 ? nc : 1 << 30;   // products still to find (8-word windows stop looking then)
                    for (int w = 0; w < W && !pre && left > 0; ++w) {
                    u64 m = rm_word<W>(L, i, w) & rm_word<W>(L, J, w) & below_word<W>(J, w);
                    while (m) {
                        const int K = (w << 6) + __ffsll((long long)m) - 1;
                        m &= m - 1;
                        --left;
                        const double* bki = L.Ls + rowi + 36 * row_rank<W>(L, i, K);
                        const double* bkj = L.Ls + rowJ + 36 * row_rank<W>(L, J, K);
                        double li[6], bj[36];   // (all loads first: from the workspace every wait is a memory round trip)
#pragma unroll
                        for (int k = 0; k < 6; ++k) li[k] = bki[6 * k + r];
#pragma unroll
                        for (int q = 0; q < 36; ++q) bj[q] = W > 1 ? bkj[q] : 0.0;
#pragma unroll
                        for (int c = 0; c < 6; ++c) {
                            double acc = 0.0;
#pragma unroll
                            for (int k = 0; k < 6; ++k) acc = __builtin_fma(li[k], W > 1 ? bj[6 * k + c] : bkj[6 * k + c], acc);
                            S[c] -= acc;
                        }
                    }
                    }
                } else if (kind == 2) {
#pragma unroll
                    for (int c = 0; c < 6; ++c) S[c] = L.b[6 * J + c];
                    if (GLOBAL_A && has_pushed_child<W>(L, J)) {
#pragma unroll
                        for (int c = 0; c < 6; ++c) S[c] -= L.As[28 * J + 21 + c];
                    }
                    const bool pre = rhs_is_presummed<W>(L, J);
                    if (pre) {
#pragma unroll
                        for (int c = 0; c < 6; ++c) S[c] -= L.yrow[6 * J + c];
                    }
                    int kb = 0;
                    for (int w = 0; w < W && !pre; ++w) {
                    u64 m = rm_word<W>(L, J, w) & below_word<W>(J, w);
                    while (m) {
                        const int K = (w << 6) + __ffsll((long long)m) - 1;
                        m &= m - 1;
                        const double* bkj = L.Ls + rowJ + 36 * kb++;
                        if (pushed<W>(L, K)) {
                            if (!GLOBAL_A) {
#pragma unroll
                                for (int c = 0; c < 6; ++c) S[c] -= L.Us[28 * K + 21 + c];
                            }
                            continue;
                        }
                        double li[6], bj[36];
#pragma unroll
                        for (int k = 0; k < 6; ++k) li[k] = L.yrow[6 * K + k];
#pragma unroll
                        for (int q = 0; q < 36; ++q) bj[q] = W > 1 ? bkj[q] : 0.0;
#pragma unroll
                        for (int c = 0; c < 6; ++c) {
                            double acc = 0.0;
#pragma unroll
                            for (int k = 0; k < 6; ++k) acc = __builtin_fma(li[k], W > 1 ? bj[6 * k + c] : bkj[6 * k + c], acc);
                            S[c] -= acc;
                        }
                    }
                    }
                }
                // G_J from the raw diagonal block: entry (p, q), p >= q, at dJ + 6 p + q
                double G[6][6], ig[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    double dj = L.Ls[dJ + 7 * j];
#pragma unroll
                    for (int k = 0; k < j; ++k) dj = __builtin_fma(-G[j][k], G[j][k], dj);
                    ok = ok && (dj > 0.0) && (dj < DBL_MAX);
                    double g, igj;
                    sqrt_and_rsqrt(fmax(dj, 1e-300), g, igj);
                    igj = __builtin_fma(igj, __builtin_fma(-g, igj, 1.0), igj);  // one Newton step: 1/g to ~1e-16
                    G[j][j] = g;
                    ig[j] = igj;
#pragma unroll
                    for (int i2 = j + 1; i2 < 6; ++i2) {
                        double v = L.Ls[dJ + 6 * i2 + j];
#pragma unroll
                        for (int k = 0; k < j; ++k) v = __builtin_fma(-G[i2][k], G[j][k], v);
                        G[i2][j] = v * ig[j];
                    }
                }
                if (kind == 0) {
                    // (static indices only: a runtime row index would push G into scratch memory)
#pragma unroll
                    for (int rr = 0; rr < 6; ++rr) {
                        if (r == rr) {
#pragma unroll
                            for (int c = 0; c < rr; ++c) L.Ls[dJ + 6 * c + rr] = G[rr][c];
                            L.diagL[6 * J + rr] = ig[rr];  // the INVERSE pivot: back-substitution multiplies
                        }
                    }
                } else {
                    double x[6];
#pragma unroll
                    for (int c = 0; c < 6; ++c) {
                        double v = S[c];
#pragma unroll
                        for (int k = 0; k < c; ++k) v = __builtin_fma(-x[k], G[c][k], v);
                        x[c] = v * ig[c];
                    }
                    if (kind == 1) {
#pragma unroll
                        for (int c = 0; c < 6; ++c) L.Ls[bi + 6 * c + r] = x[c];
                    } else {
#pragma unroll
                        for (int c = 0; c < 6; ++c) L.yrow[6 * J + c] = x[c];
                    }
                }
            }
        }
        __syncthreads();
#ifdef LOCAMD_WINDOW_TIMING
        if (lane == 0) L.tim[4] += clock64() - locamd_t1;
        if (lane == 0 && LOCAMD_WINDOW_TIMING == 4) L.tim[dense_level ? 6 : 1] += clock64() - locamd_t1;
#endif
        if (block_any<NW>(!ok, L.red, lane)) return false;
    }
    if (W > 1 && GLOBAL_A && lr < L.nlev) {
        LOCAMD_TIC();
        const bool okr = root_factor_and_solve<W, NW>(L, lane, L.lvl_col[L.nlev], lambda);
        LOCAMD_TOC(3);
        LOCAMD_TOC4(2);
        if (!okr) return false;
    }
    LOCAMD_TIC();
    // back substitution: x_J = G_J^-T (y_J - sum_{i in column J} L_iJ^T x_i), levels downwards
    for (int l = lr - 1; l >= 0; --l) {
#if defined(LOCAMD_WINDOW_TIMING) && LOCAMD_WINDOW_TIMING >= 2
        long long locamd_ts = clock64();
#endif
        const int c0 = L.lvl_col[l], ncol = L.lvl_col[l + 1] - c0;
        for (int base = 0; base < ncol; base += NT) {
            const int idx = base + lane;
            if (idx < ncol) {
                const int J = L.colorder[c0 + idx];
                const int dJ = L.boff[J + 1] - 36;
                double t[6];
#pragma unroll
                for (int c = 0; c < 6; ++c) t[c] = L.yrow[6 * J + c];
                LOCAMD_SUB3(0);
                for (int w = 0; w < W; ++w) {
                u64 m = cm_word<W>(L, J, w);
                while (m) {
                    const int i = (w << 6) + __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const double* bk = L.Ls + L.boff[i] + 36 * row_rank<W>(L, i, J);
                    double xi[6];
#pragma unroll
                    for (int r = 0; r < 6; ++r) xi[r] = L.x[6 * i + r];
#pragma unroll
                    for (int c = 0; c < 6; ++c) {
                        double acc = 0.0;
#pragma unroll
                        for (int r = 0; r < 6; ++r) acc = __builtin_fma(bk[6 * c + r], xi[r], acc);
                        t[c] -= acc;
                    }
                }
                }
                LOCAMD_SUB3(1);
                double xs[6];
#pragma unroll
                for (int rr = 5; rr >= 0; --rr) {
                    double v = t[rr];
#pragma unroll
                    for (int s2 = rr + 1; s2 < 6; ++s2) v = __builtin_fma(-L.Ls[dJ + 6 * rr + s2], xs[s2], v);
                    xs[rr] = v * L.diagL[6 * J + rr];
                }
                LOCAMD_SUB3(2);
#pragma unroll
                for (int c = 0; c < 6; ++c) L.x[6 * J + c] = xs[c];
            }
        }
        __syncthreads();
        LOCAMD_SUB3(6);
    }
    LOCAMD_TOC(5);
    return true;
}

// ---- small windows: arrays in LDS, one-word masks (T = 10 ... 30 poses: the reference's own sliding windows) ------------------
// The same factorisation, levels and storage as factor_and_solve_sparse, scheduled for LATENCY: such a window occupies a handful
// of lanes and its time is the dependent chain per level — index look-ups, the 6x6 Cholesky, the row solves — not arithmetic
// throughput (measured on a 10-pose chain: the one-lane-per-column mode spent 5 k cycles per level, 650 FMAs by one lane and
// four look-up round trips; a trial solve was 49 k of the 57 k cycles of an LM trial).  Here every level is
//   stage A (levels >= 1): all updates from earlier columns, one lane per ROW of a block / per right-hand side, one uniform
//            code path:  out[c] = src[c] - sum_K sum_k a_K[k] L_JK[c][k]  with a_K = row r of L_iK (blocks) or y_K (rhs);
//   stage B: one lane per row of an off-diagonal block and one per column (right-hand side, stores G): each factors its
//            column's 6x6 diagonal block itself (right-looking, reciprocal square roots by one cubic Newton step) and solves its
//            own row against it — again one code path;
// and the back-substitution keeps lane = column with the column's first block address cached in registers (SmallPlan).
#ifndef LOCAMD_SMALL_FACTOR
#define LOCAMD_SMALL_FACTOR 1
#endif

struct SmallPlan {
    int level;      // elimination-tree level of column `lane` (-1: no such column)
    int dJ;         // offset of the diagonal block of column `lane`
    u64 colmask;    // rows below the diagonal in column `lane`
    int off0;       // offset of the first of them
    bool tables;    // the schedule records fit (build_small_tables); else the window runs the generic factor_and_solve_sparse
};

__device__ __forceinline__ bool build_small_tables(const Lds& L, int lane, int nv);
__device__ __forceinline__ SmallPlan make_small_plan(const Lds& L, int lane, int nv, int level) {
    SmallPlan p;
    p.level = level; p.dJ = 0; p.colmask = 0; p.off0 = 0;
    p.tables = build_small_tables(L, lane, nv);
    if (lane < nv) {
        p.dJ = L.boff[lane + 1] - 36;
        p.colmask = L.colmask[lane];
        if (p.colmask) {
            const int i = __ffsll((long long)p.colmask) - 1;
            p.off0 = L.boff[i] + 36 * __popcll(L.rowmask[i] & ((1ull << lane) - 1));
        }
    }
    return p;
}

// The schedule of factor_and_solve_small, decoded once per launch: which block, which rows of the factor, where the result goes.
// (Without it every task of every LM trial walked the structure masks again — 64-bit mask arithmetic, population counts and
// two dependent look-ups before its first useful load: 57 % of the VALU instructions of a 10-pose window.)
// Record order of a level l: two per column (diagonal block, right-hand side) in colorder order, then one per off-diagonal
// block in otask order; the first record of level l is 2 lvl_col[l] + lvl_blk[l].  Returns false (uniformly) when a window has
// more (block, earlier column) pairs than the continuation records hold.
__device__ __forceinline__ bool build_small_tables(const Lds& L, int lane, int nv) {
    const int nb = L.boff[nv] / 36, noff = nb - nv;
    const int iLs = (int)(L.Ls - L.base), iY = (int)(L.yrow - L.base);
    if (lane == 0) *L.rec_count = nv + nb;
    __syncthreads();
    bool overflow = false;
    for (int p = lane; p < nv + noff; p += 64) {
        const bool col = p < nv;
        const int q = p - nv;
        int l = 0;
        for (int k = 1; k < L.nlev; ++k) l += col ? (p >= L.lvl_col[k]) : (q >= L.lvl_blk[k]);
        const int code = col ? L.colorder[p] : L.otask[q];
        const int J = code & 65535, i = col ? J : code >> 16;
        const u64 rmJ = L.rowmask[J], rmi = L.rowmask[i], belowJ = (1ull << J) - 1;
        const int rowJ = L.boff[J], rowi = L.boff[i], dJ = L.boff[J + 1] - 36;
        const int bi = rowi + 36 * __popcll(rmi & belowJ);
        // stage B: block jobs first (otask order), column jobs after them (colorder order)
        RecB rb;
        rb.dJ = dJ; rb.rb = col ? 6 * J : bi;
        L.recB[col ? noff + p : q] = rb;
        // stage A
        int slot = col ? 2 * L.lvl_col[l] + L.lvl_blk[l] + 2 * (p - L.lvl_col[l]) : 2 * L.lvl_col[l + 1] + q;
        RecA d, r;   // (r: the right-hand side's record, columns only)
        d.dst = iLs + bi; d.a = 0; d.b = 0; d.ctl = 0;
        r.dst = iY + 6 * J; r.a = 0; r.b = 0; r.ctl = 0;
        u64 m = rmi & rmJ & belowJ;
        bool first = true;
        int prev = slot;
        if (!m) { L.recA[slot] = d; if (col) L.recA[slot + 1] = r; }
        while (m) {
            const int K = __ffsll((long long)m) - 1;
            m &= m - 1;
            const u64 belowK = (1ull << K) - 1;
            d.a = iLs + rowi + 36 * __popcll(rmi & belowK);
            d.b = iLs + rowJ + 36 * __popcll(rmJ & belowK);
            d.ctl = 1;
            r.a = iY + 6 * K; r.b = d.b; r.ctl = 1;
            int pos = slot;
            if (!first) {
                pos = atomicAdd(L.rec_count, col ? 2 : 1);
                if (pos + (col ? 2 : 1) > L.recA_cap) { overflow = true; break; }
                L.recA[prev].ctl |= pos << 8;
                if (col) L.recA[prev + 1].ctl |= (pos + 1) << 8;
            }
            L.recA[pos] = d;
            if (col) L.recA[pos + 1] = r;
            prev = pos;
            first = false;
        }
    }
    __syncthreads();
    return __ballot(overflow) == 0;
}

__device__ __forceinline__ bool factor_and_solve_small(const Lds& L, int lane, int nv, double lambda, const SmallPlan& plan) {
    bool ok = true;
    const int noff = L.boff[nv] / 36 - nv;
    const int dLH = (int)(L.Ls - L.Hs), dYB = (int)(L.yrow - L.b);
    for (int l = 0; l < L.nlev; ++l) {
        const int c0 = L.lvl_col[l], ncol = L.lvl_col[l + 1] - c0;
        const int b0 = L.lvl_blk[l], nblk = L.lvl_blk[l + 1] - b0;
        // stage A (levels >= 1; a column of level 0 has no earlier column in its row): ONE LANE PER ENTRY.  A wave instruction
        // costs the same whether 2 or 60 lanes are active (and an f64 FMA twice as much as anything else: 8 cycles,
        // tools/lat_probe2.hip), so the updates are spread as thin as they go: per column 21 lanes for the lower triangle of the
        // diagonal block and 6 for the right-hand side (a 32-lane slot), per off-diagonal block 36 — six FMAs per earlier
        // column each, where a lane per block row issued 36 and a lane per column 162.
        if (l > 0) {
            LOCAMD_TIC();
            const int ncs = 32 * ncol, ntask = ncs + 36 * nblk, sb = 2 * c0 + b0;
            for (int base = 0; base < ntask; base += 64) {
                const int idx = base + lane;
                const bool off = idx >= ncs;
                const int t = idx - ncs, bq = t / 36;
                const int e = off ? t - 36 * bq : idx & 31;
                if (idx < ntask && (off || e < 27)) {
                    const bool rhs = !off && e >= 21;
                    RecA rec = L.recA[sb + (off ? 2 * ncol + bq : 2 * (idx >> 5) + (rhs ? 1 : 0))];
                    // entry e of the packed lower triangle (column-major) -> (r, c), three bits each
                    constexpr u64 RT = 0ull | (0ull << 0) | (1ull << 3) | (2ull << 6) | (3ull << 9) | (4ull << 12) | (5ull << 15) | (1ull << 18) |
                                       (2ull << 21) | (3ull << 24) | (4ull << 27) | (5ull << 30) | (2ull << 33) | (3ull << 36) | (4ull << 39) |
                                       (5ull << 42) | (3ull << 45) | (4ull << 48) | (5ull << 51) | (4ull << 54) | (5ull << 57) | (5ull << 60);
                    constexpr u64 CT = 0ull | (1ull << 18) | (1ull << 21) | (1ull << 24) | (1ull << 27) | (1ull << 30) | (2ull << 33) | (2ull << 36) |
                                       (2ull << 39) | (2ull << 42) | (3ull << 45) | (3ull << 48) | (3ull << 51) | (4ull << 54) | (4ull << 57) | (5ull << 60);
                    const int ec = off ? e / 6 : (rhs ? e - 21 : (int)((CT >> (3 * e)) & 7));
                    const int er = off ? e - 6 * ec : (rhs ? 0 : (int)((RT >> (3 * e)) & 7));
                    const int eo = rhs ? ec : 6 * ec + er;
                    double* dst = L.base + rec.dst + eo;
                    double out = dst[-(rhs ? dYB : dLH)];
                    const int ast = rhs ? 1 : 6, ao = rhs ? 0 : er;
                    while (rec.ctl & 1) {
                        const double* ap = L.base + rec.a + ao;
                        const double* bp = L.base + rec.b + ec;
                        double av[6], bv[6];
#pragma unroll
                        for (int k = 0; k < 6; ++k) { av[k] = ap[k * ast]; bv[k] = bp[6 * k]; }
#pragma unroll
                        for (int k = 0; k < 6; ++k) out = __builtin_fma(-av[k], bv[k], out);
                        const int next = rec.ctl >> 8;
                        if (next == 0) break;
                        rec = L.recA[next];
                    }
                    *dst = out;
                }
            }
            __syncthreads();
            LOCAMD_TOC(3);
        }
        {
            LOCAMD_TIC();
            // stage B.  (The right-hand-side tasks come last: a column's block-row tasks have read its diagonal block — in the
            // same pass or an earlier one — before its right-hand-side task overwrites that block with G.)
            const int nrow = 6 * nblk, ntask = nrow + ncol;
            const double* Sbase = l == 0 ? L.Hs : L.Ls;
            const double* ybase = l == 0 ? L.b : L.yrow;
            for (int base = 0; base < ntask; base += 64) {
                const int idx = base + lane;
                if (idx < ntask) {
                    const bool rhs = idx >= nrow;
                    const int bq = idx / 6;
                    const int r = rhs ? 0 : idx - 6 * bq;
                    const RecB rb = L.recB[rhs ? noff + c0 + (idx - nrow) : b0 + bq];
                    const double* src = rhs ? ybase + rb.rb : Sbase + rb.rb + r;
                    double* dst = rhs ? L.yrow + rb.rb : L.Ls + rb.rb + r;
                    const int st = rhs ? 1 : 6;
                    const double* Sd = Sbase + rb.dJ;
                    double A[6][6], s[6], ig[6];
#pragma unroll
                    for (int c = 0; c < 6; ++c)
#pragma unroll
                        for (int rr = c; rr < 6; ++rr) A[rr][c] = Sd[6 * c + rr];
#pragma unroll
                    for (int c = 0; c < 6; ++c) s[c] = src[c * st];
#pragma unroll
                    for (int q = 0; q < 6; ++q) A[q][q] += lambda;
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        // (a pivot <= 0 or not finite turns its reciprocal square root into NaN / inf, which the sum below catches)
                        const double g = pivot_rsqrt(A[j][j]);
                        ig[j] = g;
#pragma unroll
                        for (int i2 = j + 1; i2 < 6; ++i2) A[i2][j] *= g;
#pragma unroll
                        for (int i2 = j + 1; i2 < 6; ++i2)
#pragma unroll
                            for (int c = j + 1; c <= i2; ++c) A[i2][c] = __builtin_fma(-A[i2][j], A[c][j], A[i2][c]);
                    }
                    double x[6];
#pragma unroll
                    for (int c = 0; c < 6; ++c) {
                        x[c] = s[c] * ig[c];
#pragma unroll
                        for (int c2 = c + 1; c2 < 6; ++c2) s[c2] = __builtin_fma(-x[c], A[c2][c], s[c2]);
                    }
                    ok = ok && (((ig[0] + ig[1]) + (ig[2] + ig[3])) + (ig[4] + ig[5]) < DBL_MAX);
#pragma unroll
                    for (int c = 0; c < 6; ++c) dst[c * st] = x[c];
                    if (rhs) {
                        double* Gd = L.Ls + rb.dJ;
#pragma unroll
                        for (int c = 0; c < 5; ++c)
#pragma unroll
                            for (int rr = c + 1; rr < 6; ++rr) Gd[6 * c + rr] = A[rr][c];
#pragma unroll
                        for (int j = 0; j < 6; ++j) L.diagL[rb.rb + j] = ig[j];
                    }
                }
            }
            __syncthreads();
            LOCAMD_TOC(4);
            if (__ballot(!ok)) return false;
        }
    }
    LOCAMD_TIC();
    // back substitution, levels downwards, lane = column: x_J = G_J^-T (y_J - sum_i L_iJ^T x_i)
    for (int l = L.nlev - 1; l >= 0; --l) {
        if (plan.level == l) {
            const int J = lane, dJ = plan.dJ;
            double t[6], G[6][6], ig[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) t[c] = L.yrow[6 * J + c];
#pragma unroll
            for (int c = 0; c < 5; ++c)
#pragma unroll
                for (int rr = c + 1; rr < 6; ++rr) G[rr][c] = L.Ls[dJ + 6 * c + rr];
#pragma unroll
            for (int c = 0; c < 6; ++c) ig[c] = L.diagL[6 * J + c];
            u64 m = plan.colmask;
            bool first = true;
            while (m) {
                const int i = __ffsll((long long)m) - 1;
                m &= m - 1;
                const int off = first ? plan.off0 : L.boff[i] + 36 * __popcll(L.rowmask[i] & ((1ull << J) - 1));
                first = false;
                const double* bk = L.Ls + off;
                double bv[36], xi[6];
#pragma unroll
                for (int q = 0; q < 36; ++q) bv[q] = bk[q];
#pragma unroll
                for (int r = 0; r < 6; ++r) xi[r] = L.x[6 * i + r];
#pragma unroll
                for (int r = 0; r < 6; ++r)
#pragma unroll
                    for (int c = 0; c < 6; ++c) t[c] = __builtin_fma(-bv[6 * c + r], xi[r], t[c]);
            }
            double xs[6];
#pragma unroll
            for (int rr = 5; rr >= 0; --rr) {
                xs[rr] = t[rr] * ig[rr];
#pragma unroll
                for (int q = 0; q < rr; ++q) t[q] = __builtin_fma(-G[rr][q], xs[rr], t[q]);
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) L.x[6 * J + c] = xs[c];
        }
        __syncthreads();
    }
    LOCAMD_TOC(5);
    return true;
}

// GLOBAL_A: the main arrays (H, its factor, vectors, poses, records) live in an HBM workspace slice instead of LDS (large
// windows); a workgroup is one wave on one CU, whose L1 is coherent for its own stores after the workgroup barrier.
// JAC: range-edge Jacobians analytic (0) or g2o's central differences (1).  SP: SPARSE path (nv_max <= 64) or SKYLINE.
// (launch bounds: at least two waves per SIMD, i.e. at most 256 registers — a few rarely used values spill to scratch, which
//  costs far less than the halved occupancy a 257th register would)
template <bool GLOBAL_A, int JAC, bool SP, int W, int NW>
// (experiment kept as a switch: forcing 3 or 4 waves per SIMD for the workspace-mode SPARSE kernel — 168 / 128 registers, 204 /
//  484 values spilled to scratch — made BASELINE config 5 slower, 19.7 -> 26.0 / 33.1 ms per 16 384 windows: the extra waves do
//  not pay for the spill traffic)
#ifndef LOCAMD_WS_WAVES
#define LOCAMD_WS_WAVES 2
#endif
__global__ void __launch_bounds__(64 * NW, (GLOBAL_A && SP) ? LOCAMD_WS_WAVES : 2) window_lm_kernel(const WindowArgs a) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int inst = blockIdx.x;
    // NW waves per window (NW > 1: windows of 65 .. 512 poses): `lane` is the thread's index in the window's workgroup; the
    // set-up below runs on wave 0 alone (SOLO), the LM loop on all NT threads
    constexpr int NT = 64 * NW;
    constexpr bool SOLO = NW > 1;
    const int lane = threadIdx.x;
    const WindowCaps& c = a.caps;
    const int nv = a.counts[inst * 4 + 0], nr = a.counts[inst * 4 + 1], np = a.counts[inst * 4 + 2], ns = a.counts[inst * 4 + 3];
    const int n = 6 * nv;
    const int n_max = 6 * c.nv_max;
    const size_t nnz_max = window_nnz_capacity(c);
    Lds L;
    L.base = lds; L.recA = nullptr; L.recB = nullptr; L.rec_count = nullptr; L.recA_cap = 0;
    L.dense = nullptr; L.dense_off = nullptr; L.dense_cap = 0; L.dense_ok = 0; L.root = nullptr; L.root_level = 0;
    // small windows (arrays in LDS, one-word masks): the latency-scheduled factorisation (factor_and_solve_small)
    constexpr bool SMALL = SP && !GLOBAL_A && W == 1 && LOCAMD_SMALL_FACTOR;
    SmallPlan plan;
    plan.level = -1; plan.dJ = 0; plan.colmask = 0; plan.off0 = 0; plan.tables = false;
    __shared__ double s_blk[36];
    __shared__ double s_red[NW > 1 ? NW * 37 : 1];
    __shared__ int s_meta[4];
    L.red = s_red;
#ifdef LOCAMD_WINDOW_TIMING
    __shared__ long long s_tim[8];
    if (lane < 8) s_tim[lane] = 0;
    L.tim = s_tim;
    const long long t_start = clock64();
#endif
    // main arrays: LDS for small windows, this instance's slice of the HBM workspace (L1/L2-cached) for large ones
    double* p = GLOBAL_A ? a.workspace + (size_t)inst * (window_instance_doubles(c) + window_push_doubles(c)) : lds;
    L.Hs = p; p += nnz_max;
    L.Ls = p; p += nnz_max;
    L.diagL = p; p += n_max;
    L.b = p; p += n_max;
    L.x = p; p += n_max;
    L.yrow = p; p += n_max;
    L.blk = s_blk;
    L.pose = p; p += c.nv_max * 12;
    L.bak = p; p += c.nv_max * 12;
    L.rrec = p; p += c.nr_max * RREC;
    L.prec = p; p += c.np_max * PREC;
    L.srec = p; p += c.ns_max * SREC;
    double* const p_index = p;  // (index tables: here, or in LDS — below)
    p += window_index_doubles(c);
    L.Us = p; L.As = p + 28 * c.nv_max;   // (only used, and only allocated, in workspace mode)
    int* const otask_ws = reinterpret_cast<int*>(p + 56 * c.nv_max);   // 8-word windows: the off-diagonal task list
    if (GLOBAL_A) p += window_push_doubles(c);
    // the small index tables stay in LDS even when everything else is in the HBM workspace: every address in the sweep and
    // in the edge fold starts with a lookup in them, and an HBM round trip there is pure latency
    {
        double* t = GLOBAL_A ? lds : p;
        L.W = W; L.rowpre = nullptr; L.pushw = nullptr; L.lvl_mode = nullptr;
        if (SP) {
            L.rowmask = reinterpret_cast<u64*>(t); L.colmask = L.rowmask + c.nv_max * W; L.scr = L.colmask + c.nv_max * W;
            u64* tw = L.scr + c.nv_max * W;
            if (W > 1) { L.pushw = tw; tw += W; }
            int* ti = reinterpret_cast<int*>(tw);
            if (W > 1) { L.rowpre = ti; ti += c.nv_max * W; }
            L.perm = ti; ti += c.nv_max;
            L.boff = ti; ti += c.nv_max + 1;
            L.ioff = ti; ti += c.nv_max + 1;
            L.lvl_col = ti; ti += c.nv_max + 1;
            L.lvl_blk = ti; ti += c.nv_max + 1;
            if (W > 1) { L.lvl_mode = ti; ti += c.nv_max + 1; }
            L.colorder = ti; ti += c.nv_max;
            if (W > 1) { L.dense_off = ti; ti += c.nv_max + 2; L.dense = ti; L.dense_cap = 2 * c.nv_max; ti += L.dense_cap; }
            L.otask = W > 1 ? otask_ws : ti;
            L.fb = L.last = nullptr;
        } else {
            int* ti = reinterpret_cast<int*>(t);
            L.fb = ti; L.last = ti + c.nv_max; L.boff = ti + 2 * c.nv_max; L.ioff = ti + 3 * c.nv_max + 1;
            L.rowmask = L.colmask = L.scr = nullptr; L.perm = L.lvl_col = L.lvl_blk = L.colorder = L.otask = nullptr;
        }
        double* q = GLOBAL_A && window_index_in_lds(c) ? lds + (window_table_bytes(c) + 7) / 8 : p_index;
        L.ilist = reinterpret_cast<int*>(q); q += ints_as_doubles(window_incidences(c));
        L.shared = reinterpret_cast<int*>(q); q += ints_as_doubles((size_t)c.nr_max + c.ns_max + 1);
        L.r_idx = reinterpret_cast<int32_t*>(q); q += ints_as_doubles(2 * (size_t)c.nr_max);
        L.p_idx = reinterpret_cast<int32_t*>(q); q += ints_as_doubles((size_t)c.np_max);
        L.s_idx = reinterpret_cast<int32_t*>(q);
        L.root = GLOBAL_A && W > 1 && window_root_lds_doubles(c) != 0
                     ? lds + (window_table_bytes(c) + 7) / 8 + (window_index_in_lds(c) ? window_index_doubles(c) : 0) : nullptr;
        if (!GLOBAL_A) p += (window_table_bytes(c) + 7) / 8;
    }
    L.nlev = 0;
    if (GLOBAL_A) {
        L.r_val = a.r_val + (size_t)inst * c.nr_max * 5;
        L.p_val = a.p_val + (size_t)inst * c.np_max * 18;
        L.s_val = a.s_val + (size_t)inst * c.ns_max * 48;
    }
    const double* gpose_in = a.poses_in + (size_t)inst * c.nv_max * 12;
    double* gpose = a.poses + (size_t)inst * c.nv_max * 12;
    if (!SOLO || lane < 64) {
    // edge tables: the values are staged into LDS for small windows and read in place for large ones; the index tables
    // always get a private writable copy (the SPARSE path relabels them)
    for (int i = lane; i < nr * 2; i += 64) L.r_idx[i] = a.r_idx[(size_t)inst * c.nr_max * 2 + i];
    for (int i = lane; i < np; i += 64) L.p_idx[i] = a.p_idx[(size_t)inst * c.np_max + i];
    for (int i = lane; i < ns * 4; i += 64) L.s_idx[i] = a.s_idx[(size_t)inst * c.ns_max * 4 + i];
    if (!GLOBAL_A) {
        double* st_rval = p; p += c.nr_max * 5;
        double* st_pval = p; p += c.np_max * 18;
        double* st_sval = p; p += c.ns_max * 48;
        L.r_val = st_rval; L.p_val = st_pval; L.s_val = st_sval;
        if (SMALL) {
            p += (p - lds) & 1;   // 16-byte records
            L.recA_cap = (int)small_recA_capacity(c);
            L.recA = reinterpret_cast<RecA*>(p); p += 2 * L.recA_cap;
            L.recB = reinterpret_cast<RecB*>(p); p += nnz_max / 36;
            L.rec_count = reinterpret_cast<int*>(p);
        }
        for (int i = lane; i < nr * 5; i += 64) st_rval[i] = a.r_val[(size_t)inst * c.nr_max * 5 + i];
        for (int i = lane; i < np * 18; i += 64) st_pval[i] = a.p_val[(size_t)inst * c.np_max * 18 + i];
        for (int i = lane; i < ns * 48; i += 64) st_sval[i] = a.s_val[(size_t)inst * c.ns_max * 48 + i];
    }
    sync_<SOLO>();
    if (SP) {
        const int nb_max = (int)(nnz_max / 36);
        int nb;
        if (W == 1) {
            nb = compute_sparse<GLOBAL_A>(L, lane, nv, c.nv_max, nb_max, nr, ns, a.natural_order != 0, plan.level);
            if (nb > nb_max) { __syncthreads(); nb = compute_sparse<GLOBAL_A>(L, lane, nv, c.nv_max, nb_max, nr, ns, true, plan.level); }  // (the natural order cannot exceed the envelope capacity)
        } else {
            nb = compute_sparse_mw<W, GLOBAL_A, SOLO>(L, lane, nv, c.nv_max, nb_max, nr, ns, a.natural_order != 0);
            if (nb > nb_max) { sync_<SOLO>(); nb = compute_sparse_mw<W, GLOBAL_A, SOLO>(L, lane, nv, c.nv_max, nb_max, nr, ns, true); }
        }
        // relabel: pose slots -> elimination positions
        for (int e = lane; e < nr; e += 64) {
            L.r_idx[2 * e] = L.perm[L.r_idx[2 * e]];
            const int v1 = L.r_idx[2 * e + 1];
            if (v1 >= 0) L.r_idx[2 * e + 1] = L.perm[v1];
        }
        for (int e = lane; e < np; e += 64) L.p_idx[e] = L.perm[L.p_idx[e]];
        for (int e = lane; e < ns; e += 64) { L.s_idx[4 * e] = L.perm[L.s_idx[4 * e]]; L.s_idx[4 * e + 1] = L.perm[L.s_idx[4 * e + 1]]; }
        for (int i = lane; i < nv * 12; i += 64) L.pose[L.perm[i / 12] * 12 + i % 12] = gpose_in[i];
    } else {
        compute_skyline(L, lane, nv, nr, ns);
        for (int i = lane; i < nv * 12; i += 64) L.pose[i] = gpose_in[i];
    }
    sync_<SOLO>();
    if (SMALL) plan = make_small_plan(L, lane, nv, plan.level);
    if (!SOLO) {
        const int nnz = L.boff[nv];
        for (int i = lane; i < nnz; i += 64) L.Hs[i] = 0.0;   // (once per launch: see the LM loop)
        __syncthreads();
        compute_incidence<SP, SOLO>(L, lane, nv, nr, np, ns);
    }
    }
    if (SOLO) {   // the other waves join: the structure is in LDS / the workspace, the level count in wave 0's registers
        if (lane == 0) { s_meta[0] = L.nlev; s_meta[1] = L.dense_ok; s_meta[2] = L.root_level; }
        __syncthreads();
        L.nlev = s_meta[0]; L.dense_ok = s_meta[1]; L.root_level = s_meta[2];
        L.colmode = 0; L.pushmask = 0;
        {
            const int nnz = L.boff[nv];
            for (int i = lane; i < nnz; i += NT) L.Hs[i] = 0.0;   // (once per launch: see the LM loop)
            __syncthreads();
        }
        compute_incidence_wide<SP, NW>(L, lane, nv, nr, np, ns);
    }

    constexpr double tau = 1e-5, good_lo = 1.0 / 3.0, good_hi = 2.0 / 3.0;
    constexpr int max_trials = 10;
    double lambda = 0.0, ni = 2.0, cur_chi = 0.0, last_plain = 0.0;
#ifdef LOCAMD_WINDOW_TIMING
    if (lane == 0 && LOCAMD_WINDOW_TIMING < 2) L.tim[0] += clock64() - t_start;
    if (lane == 0 && LOCAMD_WINDOW_TIMING == 5) L.tim[6] += clock64() - t_start - L.tim[0] - L.tim[1] - L.tim[2];
#endif
    int it = 0, trials = 0, terminated = 0;
    const bool empty = (nv <= 0) || (nr + np + ns <= 0);

    for (int i = lane; i < n; i += NT) L.x[i] = 0.0;  // the solver's x of a fresh optimize() call
    __syncthreads();
    bool ok = !empty;
    for (it = 0; it < a.iterations && ok; ++it) {
        double plain;
        // (H is zeroed ONCE per launch, below the structure set-up: every iteration writes all 36 entries of every block that has
        //  an edge and the lower triangle of every diagonal block — the only entries of those blocks that are ever read — and the
        //  fill blocks of the factor's structure are never written: they stay zero.  Zeroing it per iteration was 8 % of the bytes
        //  BASELINE config 5 moved.)
        {
            LOCAMD_TIC();
            evaluate_edges<true, JAC, SP, NW>(a, L, inst, lane, nr, np, ns, cur_chi, plain);  // computeActiveErrors + linearize
            last_plain = plain;
            __syncthreads();
            LOCAMD_TOC(1);
        }
        {
            LOCAMD_TIC();
            build_system<SP, NW>(L, lane, n, nr, ns);
            LOCAMD_TOC(2);
        }
        if (it == 0) {  // computeLambdaInit
            double md = 0.0;
            for (int j = lane; j < n; j += NT) md = fmax(md, fabs(L.Hs[sky<SP>(L, j, j)]));
            lambda = tau * block_max<NW>(md, L.red, lane);
            ni = 2.0;
        }
        double rho = 0.0;
        int q = 0;
        do {
            // (push / pop without copies: the trial state is written to the other pose buffer and the two pointers swap)
#ifdef LOCAMD_WINDOW_TIMING
            const long long locamd_tf = clock64();
#endif
            const bool ok2 = !SP ? factor_and_solve(L, lane, n, lambda)
                             : (SMALL && plan.tables ? factor_and_solve_small(L, lane, nv, lambda, plan)
                                                     : factor_and_solve_sparse<GLOBAL_A, W, NW>(L, lane, lambda));
#ifdef LOCAMD_WINDOW_TIMING
            if (!SP && lane == 0) L.tim[3] += clock64() - locamd_tf;  // SKYLINE: the whole sweep incl. back-substitution (slot 5)
#endif
            LOCAMD_TIC();
            // A failed Cholesky (H + lambda I not positive definite): g2o's solver returns false WITHOUT touching x, and LM
            // applies that stale x all the same (zero at the first solve of an optimize() call: x is cleared when the solve starts), scores
            // the trial tempChi = max double and pops the step.  Mirrored as is: L.x is only written by a successful
            // back-substitution.
            // update: X <- X * fromVectorMQT(dx), one pose per lane
            for (int v = lane; v < nv; v += NT) {
                const double* dx = L.x + v * 6;
                const double* X = L.pose + v * 12;
                double* Xn = L.bak + v * 12;
                double Rd[9];
                const double ww = 1.0 - (dx[3] * dx[3] + dx[4] * dx[4] + dx[5] * dx[5]);
                if (ww < 0) { Rd[0] = 1; Rd[1] = 0; Rd[2] = 0; Rd[3] = 0; Rd[4] = 1; Rd[5] = 0; Rd[6] = 0; Rd[7] = 0; Rd[8] = 1; }
                else { const double qd[4] = {sqrt(ww), dx[3], dx[4], dx[5]}; quat_to_mat(qd, Rd); }
                double Rn[9], tn[3];
                mat_mul(X, Rd, Rn);
                mat_vec(X, dx, tn);
                Xn[9] = X[9] + tn[0]; Xn[10] = X[10] + tn[1]; Xn[11] = X[11] + tn[2];
#pragma unroll
                for (int i = 0; i < 9; ++i) Xn[i] = Rn[i];
            }
            { double* t = L.pose; L.pose = L.bak; L.bak = t; }   // L.pose: the trial state; L.bak: the state it came from
            __syncthreads();
            ++trials;
            double temp_chi, plain2;
            evaluate_edges<false, JAC, SP, NW>(a, L, inst, lane, nr, np, ns, temp_chi, plain2);
            LOCAMD_TOC(6);
            last_plain = plain2;
            if (!ok2) temp_chi = DBL_MAX;
            double sc = 0.0;
            for (int j = lane; j < n; j += NT) sc += L.x[j] * (lambda * L.x[j] + L.b[j]);  // computeScale
            const double scale = block_sum<NW>(sc, L.red, lane) + 1e-3;
            rho = (cur_chi - temp_chi) / scale;
            if (rho > 0.0 && fabs(temp_chi) <= DBL_MAX) {  // g2o: rho > 0 && isfinite(tempChi)
                const double r21 = 2.0 * rho - 1.0;
                double alpha = 1.0 - r21 * r21 * r21;
                alpha = fmin(alpha, good_hi);
                lambda *= fmax(good_lo, alpha);
                ni = 2.0;
                cur_chi = temp_chi;
            } else {
                lambda *= ni;
                ni *= 2.0;
                { double* t = L.pose; L.pose = L.bak; L.bak = t; }   // pop
            }
            __syncthreads();
            ++q;
        } while (rho < 0.0 && q < max_trials);
        if (q == max_trials || rho == 0.0) { ok = false; terminated = 1; }
    }

    if (SP) { for (int i = lane; i < nv * 12; i += NT) gpose[i] = L.pose[L.perm[i / 12] * 12 + i % 12]; }
    else { for (int i = lane; i < nv * 12; i += NT) gpose[i] = L.pose[i]; }
    if (lane == 0) {
        double* res = a.result + (size_t)inst * 8;
        res[0] = last_plain; res[1] = cur_chi; res[2] = lambda; res[3] = (double)it; res[4] = (double)trials;
        res[5] = (double)terminated; res[6] = (double)L.shared[0]; res[7] = SP ? (double)(L.nlev * 65536 + L.boff[nv] / 36) + ((W > 1 && GLOBAL_A && L.root_level < L.nlev) ? (nv - L.lvl_col[L.root_level]) / 16.0 : 0.0) : 0.0;   // (+ poses in the root supernode / 16)
#ifdef LOCAMD_WINDOW_TIMING
        L.tim[7] = clock64() - t_start;
        for (int i = 0; i < 8; ++i) res[i] = (double)L.tim[i];
#endif
    }
}

}  // namespace

size_t window_lds_bytes(const WindowCaps& c, bool global_a) {
    // (+ the static 6x6 exchange block)
    if (global_a) return (window_table_bytes(c) + 7) / 8 * 8 + (window_index_in_lds(c) ? window_index_doubles(c) * 8 : 0) + window_root_lds_doubles(c) * 8;
    const size_t staged = (size_t)c.nr_max * 5 + (size_t)c.np_max * 18 + (size_t)c.ns_max * 48;
    return (window_instance_doubles(c) + (window_table_bytes(c) + 7) / 8 + staged + small_table_doubles(c)) * sizeof(double);
}
size_t window_workspace_doubles(const WindowCaps& c) { return window_instance_doubles(c) + window_push_doubles(c); }
constexpr int WIDE_WAVES = 8;   // waves per window of 65 .. 512 poses (two per SIMD)
constexpr size_t WIDE_STATIC_LDS = 4096;   // >= the static LDS of the several-waves kernel (reduction scratch: 2.4 KB)

template <bool GLOBAL_A, int JAC, bool SP, int W, int NW = 1>
static hipError_t launch_window_t(const WindowArgs& a, size_t lds, hipStream_t stream) {
    // the opt-in to more than 64 KiB of dynamic LDS is per device (and per kernel instantiation)
    static std::atomic<uint64_t> attr_set{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_set.load(std::memory_order_acquire) & bit)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&window_lm_kernel<GLOBAL_A, JAC, SP, W, NW>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (NW > 1 ? WIDE_STATIC_LDS : 512));  // static LDS on top
        if (e != hipSuccess) return e;
        attr_set.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((window_lm_kernel<GLOBAL_A, JAC, SP, W, NW>), dim3((unsigned)a.B), dim3(64 * NW), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_window(const WindowArgs& a, hipStream_t stream) {
    if (a.B <= 0 || a.caps.bw_max < 0 || a.caps.bw_max >= a.caps.nv_max + (a.caps.nv_max == 1)) return hipErrorInvalidValue;
    const bool global_a = a.workspace != nullptr;
    const size_t lds = window_lds_bytes(a.caps, global_a);
    if (lds > 160 * 1024 - 512) return hipErrorInvalidValue;
    const bool sp = window_sparse_path(a.caps);
    if (sp && window_mask_words(a.caps) > 1) {   // 65 .. 512 poses: 8-word masks, always in the workspace
        if (!global_a) return hipErrorInvalidValue;
        // Their structure tables fill a CU's LDS, so one such window runs per CU: it gets the CU's four SIMDs (WIDE_WAVES
        // waves); LOCAMD_WINDOW_WAVES=1 in the environment keeps the one-wave kernel (for comparison).
        static const bool one_wave = [] { const char* v = getenv("LOCAMD_WINDOW_WAVES"); return v && v[0] == '1' && v[1] == 0; }();
        if (one_wave || lds > 160 * 1024 - WIDE_STATIC_LDS) return a.jacobian ? launch_window_t<true, 1, true, 8>(a, lds, stream) : launch_window_t<true, 0, true, 8>(a, lds, stream);
        return a.jacobian ? launch_window_t<true, 1, true, 8, WIDE_WAVES>(a, lds, stream) : launch_window_t<true, 0, true, 8, WIDE_WAVES>(a, lds, stream);
    }
    const int sel = (global_a ? 4 : 0) | (a.jacobian ? 2 : 0) | (sp ? 1 : 0);
    switch (sel) {
        case 0: return launch_window_t<false, 0, false, 1>(a, lds, stream);
        case 1: return launch_window_t<false, 0, true, 1>(a, lds, stream);
        case 2: return launch_window_t<false, 1, false, 1>(a, lds, stream);
        case 3: return launch_window_t<false, 1, true, 1>(a, lds, stream);
        case 4: return launch_window_t<true, 0, false, 1>(a, lds, stream);
        case 5: return launch_window_t<true, 0, true, 1>(a, lds, stream);
        case 6: return launch_window_t<true, 1, false, 1>(a, lds, stream);
        default: return launch_window_t<true, 1, true, 1>(a, lds, stream);
    }
}


}  // namespace locamd
