// gfx950 (MI355X / CDNA4): TRANSLATION-ONLY windows that are a CHAIN with a small dense BORDER — BASELINE config 4, anchor
// self-calibration: one tag trajectory (a chain of poses tied by the zero-range smoothness edges of Robot::new_vertex,
// robot.cpp:75-110, localization.cpp:338-340) whose every pose ranges to a handful of nodes that are unknowns themselves
// ("every node moves" when topic/relative_range exists, localization.cpp:94-98; addRLRangeEdge, :378-436).  FOUR waves (one workgroup) per instance.
//
// Same problem, same LM as every other kernel here (g2o's Levenberg-Marquardt; Localization::solve(),
// localization.cpp:164-170; EdgeSE3Range, types_edge_se3range.cpp:105-114; Cauchy kernels, :608-627).  Translation-only means what
// it means in chain3_kernel.hip: no lever arm, identity rotations, no rotation information anywhere — the 6-DoF system then reduces
// EXACTLY to 3x3 blocks (every dropped term is an exact zero; SURVEY §8(a) note).
//
// Structure.  The last nb pose slots are the border (unknown anchors), the first n = nv - nb the chain.  H is
//        [ T   B^T ]   T  block-tridiagonal (3x3 blocks; one rank-1 coupling (w J_p) J_{p-1}^T per consecutive pair)
//        [ B   C   ]   B  dense (3 nb) x (3 n), C dense (3 nb) x (3 nb)
// and is solved the way its shape asks, not by a general sparse factorisation (window_lm_kernel runs this very graph through
// ordering + symbolic + 19 elimination levels of scattered 6x6 blocks: 425x the algorithmic bytes, one instance per CU):
//   forward    the chain's factor with LANE = ROW: the couplings are rank-1, so Sherman-Morrison turns G_p = chol(H_pp + lambda I -
//              W_p W_p^T), y_p into two scalars (alpha, beta) that run from row to row through DPP, all rows factoring AT ONCE
//              before and after (wave3_kernel.hip); then ONE sequential loop carries the border's rows of the factor, lane = border
//              row: F_p = (B_p - (F_{p-1} . g_p) u_p^T) G_p^-T (three numbers per lane, in registers), r_b -= F_p y_p;
//   Schur      S = C + lambda I - sum_p F_p F_p^T: a (3 nb) x (3 n) x (3 nb) SYRK — the one GEMM-shaped piece, on the f64 matrix
//              cores (v_mfma_f64_16x16x4_f64), four poses per step, accumulators in registers for the whole sweep;
//   border     dense Cholesky of S in LDS (<= 45 x 45, lower triangle packed), the right-hand side riding along as one more row;
//   chain      z = L^-1 (b_c - B^T x_b), x_c = L^-T z: two more sweeps with lane = row — recurrences z_p = c_p + a_p (b_p . z_{p-1}) whose
//              linear part is rank one, run as prefix scans of such maps (map3_scan: six DPP steps per 64 rows; round 4).
// F is never stored (it would be 184 KB per instance and trial); B is (dense 3x3 blocks in an HBM workspace, streamed: 3 D doubles
// per chain row, read twice per trial).  What the sequential sweeps touch lives in LDS (26 doubles per chain row + the border's
// dense arrays), what only thread-parallel phases touch (translations, the stale step, edge records) in HBM with coalesced access.
//
// Four waves.  A single wave issues one f64 instruction per ~8 cycles, and a 256-pose sweep is ~130 instructions per pose, so the
// sequential sweeps, not bytes, bound one instance.  The host therefore cuts the chain into up to four SEGMENTS at separator poses
// that JOIN THE BORDER (one level of nested dissection: 256 + 10 poses become 4 x ~63 chain rows + 13 border poses): every wave
// sweeps its own segment with its own MFMA accumulators, the partial Schur complements are subtracted in wave order
// (bit-reproducible), the dense factorisation and all thread-parallel phases (linearisation, trial scoring, B^T x, the update) run
// on all 256 threads.  Rows (chain rows segment by segment, then border rows) walk host-packed edge records
// [chunk of 64 rows][slot][lane] — coalesced, no index chasing; sums over a border pose's many edges are reduced in one fixed
// order (no atomics; 32 loads in flight per wave).  The border's dense factorisation runs on the whole block, one border pose
// (three columns) per step.  Measured on config 4 (256 + 10 poses, 16 LM trials per solve, 128 hypotheses = one GPU's share):
// window_lm_kernel 15.1 ms; this kernel with one wave per instance 6.9 ms; four waves + separators 3.0 ms; + blocked dense
// factorisation, g kept from the forward sweep, the reduction's loads in flight: 2.7 ms; + the latency round (the sequential parts were
// bound by LDS / HBM round trips per row, not by arithmetic: lane = row sweeps with DPP hand-over, LDS reads issued together, LDS-only
// synchronisation next to loads in flight, deeper prefetch of B rows and edge records, the border's back-substitution in registers):
// 1.9 ms (all 1 024 hypotheses on one GPU: 7.8 ms, 1.3e5 solves/s against 8.5e3).  Cycle shares now (tools/dev/probe_arrow3.py,
// make arrowtiming): linearisation 29 %, forward sweep + Schur 28 %, dense factorisation 17 %, chain z / x sweeps 11 %, trial scoring
// 10 %, B^T x 5 %.  Tried and dropped: the border's dense Cholesky on one wave — rows in registers with v_readlane hand-outs (2.5x
// slower), or in LDS with one row of the trailing update per lane (1.7x slower: 36 dependent LDS round trips per step);
// B transposed to [border row][component][chain row] (coalesced for the thread-parallel phases, but the forward sweep's 39 lanes
// then touch 39 cache lines per load: the sweep +20 %, the whole solve +7 %).
#include "window_kernel.h"
#include "device_math.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>

#include <atomic>

namespace locamd {

constexpr int ARROW_NW = 4;   // waves per instance = segments the chain is cut into at most

// doubles of HBM workspace per instance: chain translations (two buffers) + the stale step, B rows, per-(row, border pose) shares
static __host__ __device__ inline size_t window_arrow3_workspace_doubles_dev(const WindowCaps& c, int nb_max) {
    return (size_t)c.nv_max * 9 + (size_t)c.nv_max * 3 * nb_max * 3 + (size_t)c.nv_max * nb_max * 9;
}
size_t window_arrow3_workspace_doubles(const WindowCaps& c, int nb_max) { return window_arrow3_workspace_doubles_dev(c, nb_max); }
// doubles of LDS per instance
static __host__ __device__ inline size_t arrow3_lds_doubles(int nv_max, int nb_max) {
    const size_t D = 3 * (size_t)nb_max, D16 = 16 * ((D + 15) / 16);
    return (size_t)nv_max * 28 + D * (D + 1) / 2 + (D + 1) * (D + 2) / 2 + 3 * D + ARROW_NW * D + ARROW_NW * 4 * D16 * 3 + 6 * (size_t)nb_max + 16 + 3 * (size_t)kArrowMaxAnchors;
}
size_t window_arrow3_lds_bytes(const WindowCaps& c, int nb_max) { return arrow3_lds_doubles(c.nv_max, nb_max) * sizeof(double); }

namespace {

extern __shared__ double ldsA[];
#ifdef LOCAMD_ARROW_TIMING   // diagnostic build: cycle stamps of the phases go to result[1..7] (never benchmarked, never shipped)
#define AT_DECL long long at_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long at_t = clock64()
#if LOCAMD_ARROW_TIMING == 2   // the phases of the linearisation instead of the solve's (3: the forward sweep split, see arrow_solve)
#define AT(slot) do { at_t = clock64(); } while (0)
#define AT2(slot) do { const long long t_ = clock64(); at_[slot] += t_ - at_t; at_t = t_; } while (0)
#else
#define AT(slot) do { const long long t_ = clock64(); at_[slot] += t_ - at_t; at_t = t_; } while (0)
#define AT2(slot) do {} while (0)
#endif
#else
#define AT2(slot) do {} while (0)
#define AT_DECL do {} while (0)
#define AT(slot) do {} while (0)
#endif
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_or_zero(double v, double identity) {
    const int ilo = __double2loint(identity), ihi = __double2hiint(identity);
    const int lo = __builtin_amdgcn_update_dpp(ilo, __double2loint(v), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(ihi, __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
// the value of lane - 1 / lane + 1 (wave_shr:1 / wave_shl:1 cross the 16-lane rows on gfx9; no neighbour: 0), of lane l (wave-uniform l)
__device__ __forceinline__ double lane_from_prev(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_from_next(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double read_lane_dyn(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double read_lane63(double v) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {   // DPP row shifts + row broadcasts: every lane gets the same bits
    v += dpp_or_zero<0x111, 0xF>(v, 0.0);
    v += dpp_or_zero<0x112, 0xF>(v, 0.0);
    v += dpp_or_zero<0x114, 0xF>(v, 0.0);
    v += dpp_or_zero<0x118, 0xF>(v, 0.0);
    v += dpp_or_zero<0x142, 0xA>(v, 0.0);
    v += dpp_or_zero<0x143, 0xC>(v, 0.0);
    return read_lane63(v);
}
__device__ __forceinline__ double wave_max(double v) {   // non-negative inputs
    v = fmax(v, dpp_or_zero<0x111, 0xF>(v, 0.0));
    v = fmax(v, dpp_or_zero<0x112, 0xF>(v, 0.0));
    v = fmax(v, dpp_or_zero<0x114, 0xF>(v, 0.0));
    v = fmax(v, dpp_or_zero<0x118, 0xF>(v, 0.0));
    v = fmax(v, dpp_or_zero<0x142, 0xA>(v, 0.0));
    v = fmax(v, dpp_or_zero<0x143, 0xC>(v, 0.0));
    return read_lane63(v);
}
// one wave per workgroup: LDS operations of a wave execute in order; the fence keeps the compiler from moving them
__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// the same for data that only THIS wave exchanges through LDS while global loads are in flight: the workgroup-scope fences of wsync()
// also wait for every outstanding global load (vmcnt(0)) — in the forward sweep that serialised the prefetch of the next B rows
// (an HBM round trip per four rows).  LDS operations of one wave complete in order; this waits for them only.
__device__ __forceinline__ void wsync_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
// Inclusive scan, over the lanes of a wave, of maps F(z) = c + a (b . z) on 3-vectors under composition, earlier lanes first: afterwards
// lane l holds the composite of the maps of lanes 0 .. l.  The chain's two substitution sweeps are such recurrences — z_p = L_p^-1 (b_p -
// u_p (g_p . z_{p-1})) — and their linear part is RANK ONE, which composition preserves: (a_q b_q^T)(a_p b_p^T) = a_q (b_q . a_p) b_p^T.  Six
// DPP steps (row shifts by 1, 2, 4, 8, then the row broadcasts: the scan the compiler itself emits for wave-wide atomics) instead of 64
// dependent hand-overs from lane to lane; lanes without a source in a step keep their map.
struct Map3 { double a0, a1, a2, b0, b1, b2, c0, c1, c2; };
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void map3_scan_step(Map3& m, const bool valid) {
    const double pa0 = dpp_or_zero<CTRL, ROW_MASK>(m.a0, 0.0), pa1 = dpp_or_zero<CTRL, ROW_MASK>(m.a1, 0.0), pa2 = dpp_or_zero<CTRL, ROW_MASK>(m.a2, 0.0);
    const double pb0 = dpp_or_zero<CTRL, ROW_MASK>(m.b0, 0.0), pb1 = dpp_or_zero<CTRL, ROW_MASK>(m.b1, 0.0), pb2 = dpp_or_zero<CTRL, ROW_MASK>(m.b2, 0.0);
    const double pc0 = dpp_or_zero<CTRL, ROW_MASK>(m.c0, 0.0), pc1 = dpp_or_zero<CTRL, ROW_MASK>(m.c1, 0.0), pc2 = dpp_or_zero<CTRL, ROW_MASK>(m.c2, 0.0);
    const double s1 = m.b0 * pc0 + m.b1 * pc1 + m.b2 * pc2;   // (this map after the earlier one: c' = c + a (b . c_p), a' = a (b . a_p), b' = b_p)
    const double s2 = m.b0 * pa0 + m.b1 * pa1 + m.b2 * pa2;
    m.c0 = __builtin_fma(m.a0, s1, m.c0); m.c1 = __builtin_fma(m.a1, s1, m.c1); m.c2 = __builtin_fma(m.a2, s1, m.c2);   // (no source: c_p = 0)
    m.a0 = valid ? m.a0 * s2 : m.a0; m.a1 = valid ? m.a1 * s2 : m.a1; m.a2 = valid ? m.a2 * s2 : m.a2;
    m.b0 = valid ? pb0 : m.b0; m.b1 = valid ? pb1 : m.b1; m.b2 = valid ? pb2 : m.b2;
}
__device__ __forceinline__ void map3_scan(Map3& m, const int lane) {
    map3_scan_step<0x111, 0xF>(m, (lane & 15) >= 1);
    map3_scan_step<0x112, 0xF>(m, (lane & 15) >= 2);
    map3_scan_step<0x114, 0xF>(m, (lane & 15) >= 4);
    map3_scan_step<0x118, 0xF>(m, (lane & 15) >= 8);
    map3_scan_step<0x142, 0xA>(m, ((lane >> 4) & 1) != 0);
    map3_scan_step<0x143, 0xC>(m, lane >= 32);
}
__device__ __forceinline__ double pivot_rsqrtA(double d) {   // window_kernel.hip: pivot_rsqrt
    const double y = __builtin_amdgcn_rsq(d);
    const double t = d * y;
    const double e = __builtin_fma(-t, y, 1.0);
    const double pq = __builtin_fma(0.375, e, 0.5);
    const double ye = y * e;
    return __builtin_fma(ye, pq, y);
}

#pragma clang fp contract(off)
__device__ __forceinline__ double sq3_plainA(double dx, double dy, double dz) { return dx * dx + dy * dy + dz * dz; }
// g2o's central difference along axis D of endpoint `which` (chain3_kernel.hip: range_jac_numeric3); NEAR: the perturbed norms
// from the central one n0 (device_math.h: sqrt_ieee_near_c — the same correctly rounded numbers)
template <int D, bool NEAR>
__device__ __forceinline__ double range_jac_numericA(const double* p0, const double* p1, int which, double meas, double n0, double h0) {
    constexpr double delta = 1e-9;
    constexpr double scalar = 1.0 / (2 * delta);
    double a[3] = {p0[0], p0[1], p0[2]}, b[3] = {p1[0], p1[1], p1[2]}, am[3] = {p0[0], p0[1], p0[2]}, bm[3] = {p1[0], p1[1], p1[2]};
    if (which == 0) { a[D] = delta + p0[D]; am[D] = -delta + p0[D]; }
    else { b[D] = delta + p1[D]; bm[D] = -delta + p1[D]; }
    const double xp = sq3_plainA(a[0] - b[0], a[1] - b[1], a[2] - b[2]), xm = sq3_plainA(am[0] - bm[0], am[1] - bm[1], am[2] - bm[2]);
    const double ep = meas - (NEAR ? sqrt_ieee_near_c(xp, n0, h0, (h0 * h0) * (h0 + h0)) : sqrt_ieee_unscaled(xp));
    const double em = meas - (NEAR ? sqrt_ieee_near_c(xm, n0, h0, (h0 * h0) * (h0 + h0)) : sqrt_ieee_unscaled(xm));
    double bak = ep;
    bak -= em;
    return scalar * bak;
}
#pragma clang fp contract(fast)

// where one instance's arrays live
struct ArrowCtx {
    // LDS (doubles)
    double *PK;                            // per chain row: H_qq (6), b_q (3), coupling u (3), v (3), pad: 16
    double *GZ;                            // per chain row: factor (3 strict-lower + 3 reciprocal pivots), y / z / x (3), g = G_{q-1}^-1 v_q (3): 12
    double *C0, *S;                        // border block, lower triangle packed: C0 D rows; S D + 1 rows (row D: the right-hand side)
    double *bB, *xB, *xsB;                 // D each
    double *RA;                            // [NW][D] the segments' shares of the border's right-hand side
    double *FX;                            // [NW][4][D16][3]
    double *TB;                            // [2][nb][3] border translations (state / trial state)
    double *AN;                            // [kArrowMaxAnchors][3] the fixed anchors an edge may name (the host checks the indices)
    double *red;                           // block reductions
    // HBM
    double *TT;                            // [2][n][3] chain translations by row
    double *XS;                            // [n][3] the solver's x (stale when a factorisation fails)
    double *BB;                            // [n][D][3]  B_q rows: (border row r, chain component k) — the layout the forward sweep streams (lane = r)
    double *CS;                            // [nb][9][npad] per (border pose, entry, row): that row's share of H_bb (6) and b_b (3)
    size_t npad;
    const double *rec, *prec;              // host-packed edge / prior records of this instance
    const int32_t* seg;                    // seg[0 .. nseg]: chain rows of segment s = [seg[s], seg[s + 1])
    int n, nb, D, D16, rows, nseg, jmax, jpmax, tid, lane, wv;
    unsigned long long jpk0, jpk1, ppk0, ppk1;   // per chunk of 64 rows, one byte each: edge / prior slots really in use (chunks 0 .. 7, 8 .. 15)
};
__device__ __forceinline__ int tri(int R, int C) { return R * (R + 1) / 2 + C; }

// reductions over the block's four waves: every thread gets the same bits (wave order)
__device__ __forceinline__ double block_sum(const ArrowCtx& c, double v) {
    const double w = wave_sum(v);
    __syncthreads();
    if (c.lane == 0) c.red[c.wv] = w;
    __syncthreads();
    return ((c.red[0] + c.red[1]) + c.red[2]) + c.red[3];
}
__device__ __forceinline__ double block_max(const ArrowCtx& c, double v) {
    const double w = wave_max(v);
    __syncthreads();
    if (c.lane == 0) c.red[c.wv] = w;
    __syncthreads();
    return fmax(fmax(c.red[0], c.red[1]), fmax(c.red[2], c.red[3]));
}

struct EdgeTerms { double J0[3], J1[3], wr, wre, chi, rho; };

// one EdgeSE3Range between translation p0 (endpoint 0) and p1 (endpoint 1 or a fixed anchor)
template <bool FULL, int JAC>
__device__ __forceinline__ EdgeTerms range_terms(const double* p0, const double* p1, bool moving1, double meas, double info) {
    EdgeTerms t;
    double u[3] = {p0[0] - p1[0], p0[1] - p1[1], p0[2] - p1[2]};
    double n = 0.0, x0 = 0.0, h0 = 0.0, inv_n = 0.0;
    if (JAC == 0) {   // (norm and reciprocal norm from one v_rsq_f64 seed: 1e-16 / 4e-15 relative)
        const double x = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
        sqrt_and_rsqrt(x, n, inv_n);
        if (!(x > 0.0)) { n = 0.0; inv_n = 0.0; }   // coincident endpoints: J = 0, what the central difference gives
    } else { x0 = sq3_plainA(u[0], u[1], u[2]); n = sqrt_ieee_unscaled_h(x0, h0); }
    const double err = meas - n;
    t.chi = err * (info * err);
    const double aux = 1.0 + t.chi;
    t.rho = fast_log_ge1(aux);
    if (FULL) {
        if (JAC == 0) {
            u[0] *= inv_n; u[1] *= inv_n; u[2] *= inv_n;
            t.J0[0] = -u[0]; t.J0[1] = -u[1]; t.J0[2] = -u[2];
            t.J1[0] = moving1 ? u[0] : 0.0; t.J1[1] = moving1 ? u[1] : 0.0; t.J1[2] = moving1 ? u[2] : 0.0;
        } else {
            t.J1[0] = 0.0; t.J1[1] = 0.0; t.J1[2] = 0.0;
            if (x0 >= 1e-5 && x0 < 1e300) {   // endpoints more than ~3 mm apart
                t.J0[0] = range_jac_numericA<0, true>(p0, p1, 0, meas, n, h0);
                t.J0[1] = range_jac_numericA<1, true>(p0, p1, 0, meas, n, h0);
                t.J0[2] = range_jac_numericA<2, true>(p0, p1, 0, meas, n, h0);
                if (moving1) {
                    t.J1[0] = range_jac_numericA<0, true>(p0, p1, 1, meas, n, h0);
                    t.J1[1] = range_jac_numericA<1, true>(p0, p1, 1, meas, n, h0);
                    t.J1[2] = range_jac_numericA<2, true>(p0, p1, 1, meas, n, h0);
                }
            } else {
                t.J0[0] = range_jac_numericA<0, false>(p0, p1, 0, meas, n, h0);
                t.J0[1] = range_jac_numericA<1, false>(p0, p1, 0, meas, n, h0);
                t.J0[2] = range_jac_numericA<2, false>(p0, p1, 0, meas, n, h0);
                if (moving1) {
                    t.J1[0] = range_jac_numericA<0, false>(p0, p1, 1, meas, n, h0);
                    t.J1[1] = range_jac_numericA<1, false>(p0, p1, 1, meas, n, h0);
                    t.J1[2] = range_jac_numericA<2, false>(p0, p1, 1, meas, n, h0);
                }
            }
        }
        t.wr = info * fast_rcp(aux);
        t.wre = -t.wr * err;
    }
    return t;
}

// Every edge evaluated at translation buffer `buf`, one thread per ROW (chain rows, then border rows), a wave per chunk of 64 rows:
// chi sums always; FULL: H, b, B, C as well.  A row walks its host-packed records [chunk][slot][lane] (coalesced).
template <bool FULL, int JAC>
__device__ __forceinline__ void arrow_edges(const WindowArgs& a, const ArrowCtx& c, int buf, double& robust_chi, double& plain_chi,
                                            double& max_diag, int& shared_edges
#ifdef LOCAMD_ARROW_TIMING
                                            , long long* at_, long long& at_t
#endif
                                            ) {
    const int lane = c.lane, n = c.n, nb = c.nb, D = c.D, rows = c.rows, tid = c.tid;
    const double* T = c.TT + (size_t)buf * n * 3;
    const double* TB = c.TB + buf * nb * 3;
    double rsum = 0.0, csum = 0.0;
    int nshared = 0;
    if (FULL) {
        for (int i = tid; i < D * (D + 1) / 2; i += 64 * ARROW_NW) c.C0[i] = 0.0;
        __syncthreads();
    }
    for (int ch = c.wv; ch * 64 < rows; ch += ARROW_NW) {
        const int r = ch * 64 + lane;
        const bool valid = r < rows, is_chain = r < n;
        const int ob = r - n;   // border index of a border row
        // own translation, the previous chain row's — SCALARS assigned without a loop: a small array filled in a `for k` loop is not promoted
        // to registers before InstCombine first runs (the loop is unrolled later), and a select between its elements and a loaded value
        // then becomes a load through a selected pointer, which keeps the array in scratch memory for good
        double tp0 = 0.0, tp1 = 0.0, tp2 = 0.0, tq0 = 0.0, tq1 = 0.0, tq2 = 0.0;
        if (valid) {
            if (is_chain) {
                tp0 = T[3 * r]; tp1 = T[3 * r + 1]; tp2 = T[3 * r + 2];
                if (r > 0) { tq0 = T[3 * r - 3]; tq1 = T[3 * r - 2]; tq2 = T[3 * r - 1]; }
            } else {
                tp0 = TB[3 * ob]; tp1 = TB[3 * ob + 1]; tp2 = TB[3 * ob + 2];
            }
        }
        double hd[6] = {0, 0, 0, 0, 0, 0}, hb[3] = {0, 0, 0}, cu[3] = {0, 0, 0}, cv[3] = {0, 0, 0}, cp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        unsigned mask = 0, dup = 0;
        int ndup = 0;
        // (slots in use in THIS chunk: the loop below is as long for ten border rows with three records each as for 64 chain rows with
        //  fourteen — with rows = 266 the fifth chunk doubled the first wave's share of every sweep)
        const int jm = ch < 16 ? (int)(((ch < 8 ? c.jpk0 : c.jpk1) >> (8 * (ch & 7))) & 255) : c.jmax;
        const int jpm = ch < 16 ? (int)(((ch < 8 ? c.ppk0 : c.ppk1) >> (8 * (ch & 7))) & 255) : c.jpmax;
        const double* rec = c.rec + ((size_t)ch * c.jmax * 64 + lane) * 3;
        // The row's records, NS slots per block: the next block's 3 NS loads are issued before this block's first slot and awaited after its
        // last one.  vmcnt counts loads AND stores on gfx9 and the two return out of order, so the wait for ANY load in a stream of stores is
        // `s_waitcnt vmcnt(0)` — with a load (a record one slot ahead, an anchor) inside the slot every slot drained the stores of the slot
        // before it (~1.5 k cycles of ~5 k per slot).  One rolled slot body: the block's records sit in a register array that shifts down by
        // one entry per slot (constant indices only: no scratch), a quarter of the code of four unrolled slots.
        constexpr int NS = 8;
        double rq[NS][3], rn[NS][3];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            const double* rp = rec + (size_t)(q < c.jmax ? q : c.jmax - 1) * 64 * 3;   // (slots jm .. jmax - 1 exist and hold code -1)
#pragma unroll
            for (int k = 0; k < 3; ++k) rn[q][k] = rp[k];
        }
        for (int j0 = 0; j0 < jm; j0 += NS) {
#pragma unroll
            for (int q = 0; q < NS; ++q)
#pragma unroll
                for (int k = 0; k < 3; ++k) rq[q][k] = rn[q][k];
            if (j0 + NS < jm) {
#pragma unroll
                for (int q = 0; q < NS; ++q) {
                    const int jn = j0 + NS + q;
                    const double* rp = rec + (size_t)(jn < c.jmax ? jn : c.jmax - 1) * 64 * 3;
#pragma unroll
                    for (int k = 0; k < 3; ++k) rn[q][k] = rp[k];
                }
            }
            const int jend = jm - j0 < NS ? jm - j0 : NS;
#pragma nounroll
            for (int q = 0; q < jend; ++q) {
                const double code_d = rq[0][0], meas = rq[0][1], info = rq[0][2];
#pragma unroll
                for (int t = 0; t + 1 < NS; ++t)
#pragma unroll
                    for (int k = 0; k < 3; ++k) rq[t][k] = rq[t + 1][k];
                if (valid && code_d >= 0.0) {
                    const int code = (int)code_d;
                    const bool own0 = code & 1;
                    const int kind = (code >> 1) & 3, idx = code >> 3;
                    // The other endpoint by VALUE, never through a pointer chosen among the anchor table (global), the previous row's translation
                    // (registers) and the border's (LDS): such a pointer is a flat one, the register array behind it is pinned in scratch memory and
                    // every edge pays a flat load.  Anchors and border translations are both read from LDS, unconditionally (index 0 when unused):
                    // a global load here would put an `s_waitcnt vmcnt(0)` in front of every slot — on gfx9 that also drains the slot before's stores.
                    const double* tb = kind == 0 ? c.AN + 3 * idx : TB + 3 * (kind == 2 ? idx : 0);   // (both LDS)
                    double po0 = kind == 1 ? tq0 : tb[0], po1 = kind == 1 ? tq1 : tb[1], po2 = kind == 1 ? tq2 : tb[2];
                    const double e0[3] = {own0 ? tp0 : po0, own0 ? tp1 : po1, own0 ? tp2 : po2};
                    const double e1[3] = {own0 ? po0 : tp0, own0 ? po1 : tp1, own0 ? po2 : tp2};
                    const EdgeTerms t = range_terms<FULL, JAC>(e0, e1, own0 ? kind != 0 : true, meas, info);
                    rsum += t.rho;
                    csum += t.chi;
                    if (FULL) {
                        double Jo[3], Jx[3];   // own / other endpoint
        #pragma unroll
                        for (int k = 0; k < 3; ++k) { Jo[k] = own0 ? t.J0[k] : t.J1[k]; Jx[k] = own0 ? t.J1[k] : t.J0[k]; }
        #pragma unroll
                        for (int rr = 0; rr < 3; ++rr) {
        #pragma unroll
                            for (int cc = 0; cc <= rr; ++cc) hd[rr * (rr + 1) / 2 + cc] += t.wr * Jo[rr] * Jo[cc];
                            hb[rr] += Jo[rr] * t.wre;
                        }
                        if (kind == 2) {
                            double* cs = c.CS + (size_t)idx * 9 * c.npad + r;
                            const bool again = (mask >> idx) & 1u;
                            double blk[9], own9[9];
        #pragma unroll
                            for (int i = 0; i < 3; ++i)
        #pragma unroll
                                for (int k = 0; k < 3; ++k) blk[3 * i + k] = t.wr * Jx[i] * Jo[k];   // row = the OTHER (border) pose's component, column = own component
        #pragma unroll
                            for (int rr = 0; rr < 3; ++rr) {
        #pragma unroll
                                for (int cc = 0; cc <= rr; ++cc) own9[rr * (rr + 1) / 2 + cc] = t.wr * Jx[rr] * Jx[cc];
                                own9[6 + rr] = Jx[rr] * t.wre;
                            }
                            if (is_chain) {
                                double* bb = c.BB + ((size_t)r * D + 3 * idx) * 3;
                                if (again) {
        #pragma unroll
                                    for (int k = 0; k < 9; ++k) bb[k] += blk[k];
                                } else {
        #pragma unroll
                                    for (int k = 0; k < 9; ++k) bb[k] = blk[k];
                                }
                            } else {   // block (own border pose ob, other border pose idx < ob) of C: rows = own components, columns = the other's
        #pragma unroll
                                for (int i = 0; i < 3; ++i)
        #pragma unroll
                                    for (int k = 0; k < 3; ++k) c.C0[tri(3 * ob + k, 3 * idx + i)] += blk[3 * i + k];
                            }
                            if (again) {
        #pragma unroll
                                for (int k = 0; k < 9; ++k) cs[k * c.npad] += own9[k];
                                ndup += 1;
                                dup |= 1u << idx;
                            } else {
        #pragma unroll
                                for (int k = 0; k < 9; ++k) cs[k * c.npad] = own9[k];
                            }
                            mask |= 1u << idx;
                        } else if (kind == 1) {   // the edge to the previous chain row (one per pair: checked on the host)
        #pragma unroll
                            for (int k = 0; k < 3; ++k) { cu[k] = t.wr * Jo[k]; cv[k] = Jx[k]; }
        #pragma unroll
                            for (int rr = 0; rr < 3; ++rr) {
        #pragma unroll
                                for (int cc = 0; cc <= rr; ++cc) cp[rr * (rr + 1) / 2 + cc] = t.wr * Jx[rr] * Jx[cc];
                                cp[6 + rr] = Jx[rr] * t.wre;
                            }
                        }
                    }
                }
            }
        }
        if (FULL) { AT2(7); }
        const double* prec = c.prec + ((size_t)ch * c.jpmax * 64 + lane) * 7;
        for (int j = 0; j < jpm; ++j) {   // priors: e = t + Z^-1.t, diagonal translation information, no robust kernel
            const double* pr = prec + (size_t)j * 64 * 7;
            if (!valid || pr[0] <= 0.0) continue;
            double chi = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double er = (k == 0 ? tp0 : k == 1 ? tp1 : tp2) + pr[1 + k], wd = pr[4 + k];
                chi += er * (wd * er);
                if (FULL) { hd[k * (k + 1) / 2 + k] += wd; hb[k] += -wd * er; }
            }
            rsum += chi;
            csum += chi;
        }
        if (FULL && valid) {
            for (int xb = 0; xb < nb; ++xb) {
                double* cs = c.CS + (size_t)xb * 9 * c.npad + r;
                if (!is_chain && xb == ob) {   // a border row's own share of its own diagonal block and b
#pragma unroll
                    for (int k = 0; k < 6; ++k) cs[k * c.npad] = hd[k];
#pragma unroll
                    for (int k = 0; k < 3; ++k) cs[(6 + k) * c.npad] = hb[k];
                } else if (!((mask >> xb) & 1u)) {
#pragma unroll
                    for (int k = 0; k < 9; ++k) cs[k * c.npad] = 0.0;
                    if (is_chain) {
                        double* bb = c.BB + ((size_t)r * D + 3 * xb) * 3;
#pragma unroll
                        for (int k = 0; k < 9; ++k) bb[k] = 0.0;
                    }
                }
            }
            nshared += ndup + __popc(dup);
            if (is_chain) {
                double* pk = c.PK + 16 * r;
                double* gz = c.GZ + 12 * r;
#pragma unroll
                for (int k = 0; k < 6; ++k) pk[k] = hd[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) { pk[6 + k] = hb[k]; pk[9 + k] = cu[k]; pk[12 + k] = cv[k]; }
#pragma unroll
                for (int k = 0; k < 9; ++k) gz[k] = cp[k];
            }
        }
    }
    if (FULL) {
        __syncthreads();
        AT2(1);
        // the edge (q, q + 1) also belongs to row q: its share was left in row q + 1's (G, z) slots (zero across a segment boundary:
        // there the neighbour is a separator, i.e. a border pose)
        for (int q = tid; q + 1 < n; q += 64 * ARROW_NW) {
#pragma unroll
            for (int k = 0; k < 9; ++k) c.PK[16 * q + k] += c.GZ[12 * (q + 1) + k];
        }
        AT2(2);
        // a border pose's diagonal block and b: the sum over all rows of the shares left in CS — one wave per (border pose, entry),
        // lane partial sums over its rows then the DPP tree: one fixed order
        // (a sum's chunks are added in groups of four — (v0 + v1) + (v2 + v3) — in chunk order: one fixed order.  SIXTEEN sums per wave and
        //  round with the loads of FIVE chunks (cfg4's 266 rows) issued before the first addition: a round is one HBM round trip, and with
        //  eight sums and a second pass for the fifth chunk a linearisation paid eight of them here instead of two.)
        constexpr int CSU = 16;
        for (int t0 = CSU * c.wv; t0 < nb * 9; t0 += CSU * ARROW_NW) {
            double sp[CSU];
#pragma unroll
            for (int u = 0; u < CSU; ++u) sp[u] = 0.0;
            for (int ck0 = 0; ck0 * 64 < rows; ck0 += 5) {
                double v[CSU][5];
#pragma unroll
                for (int u = 0; u < CSU; ++u)
#pragma unroll
                    for (int c5 = 0; c5 < 5; ++c5) {
                        const int p = (ck0 + c5) * 64 + lane;
                        v[u][c5] = (t0 + u < nb * 9 && p < rows) ? c.CS[(size_t)(t0 + u) * c.npad + p] : 0.0;
                    }
#pragma unroll
                for (int u = 0; u < CSU; ++u) {
                    sp[u] += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
                    sp[u] += v[u][4];
                }
            }
            AT2(5);
            // the CSU DPP trees step by step side by side (independent chains: no dependent-issue stalls); the totals end up in lane 63
#define LOCAMD_CS_STEP(CTRL, MASK) _Pragma("unroll") for (int u = 0; u < CSU; ++u) sp[u] += dpp_or_zero<CTRL, MASK>(sp[u], 0.0)
            LOCAMD_CS_STEP(0x111, 0xF); LOCAMD_CS_STEP(0x112, 0xF); LOCAMD_CS_STEP(0x114, 0xF); LOCAMD_CS_STEP(0x118, 0xF);
            LOCAMD_CS_STEP(0x142, 0xA); LOCAMD_CS_STEP(0x143, 0xC);
#undef LOCAMD_CS_STEP
            if (lane == 63) {
#pragma unroll
                for (int u = 0; u < CSU; ++u) {
                    const int t = t0 + u;
                    if (t < nb * 9) {
                        const int xb = t / 9, k = t % 9;
                        if (k < 6) {
                            const int rr = k < 1 ? 0 : (k < 3 ? 1 : 2), cc = k - rr * (rr + 1) / 2;
                            c.C0[tri(3 * xb + rr, 3 * xb + cc)] = sp[u];
                        } else {
                            c.bB[3 * xb + k - 6] = sp[u];
                        }
                    }
                }
            }
            AT2(6);
        }
        __syncthreads();
        AT2(3);
    }
    robust_chi = block_sum(c, rsum);
    plain_chi = block_sum(c, csum);
    if (FULL) {
        double md = 0.0;
        for (int q = tid; q < n; q += 64 * ARROW_NW) md = fmax(md, fmax(fabs(c.PK[16 * q]), fmax(fabs(c.PK[16 * q + 2]), fabs(c.PK[16 * q + 5]))));
        for (int r = tid; r < D; r += 64 * ARROW_NW) md = fmax(md, fabs(c.C0[tri(r, r)]));
        max_diag = block_max(c, md);
        shared_edges = (int)block_sum(c, (double)nshared);
        AT2(4);
    }
}

// (H + lambda I) x = b, the step applied to the other translation buffer.  Returns false when a pivot fails (then the stale x is
// applied: SURVEY A.6); scale_sum = g2o's computeScale sum.  All threads of the block call it.
template <int NTI>
__device__ __forceinline__ bool arrow_solve(const ArrowCtx& c, double lambda, int buf, double& scale_sum
#ifdef LOCAMD_ARROW_TIMING
                                            , long long* at_, long long& at_t
#endif
                                            ) {
    const int lane = c.lane, n = c.n, nb = c.nb, D = c.D, D16 = c.D16, tid = c.tid, wv = c.wv;
    constexpr int NACC = NTI * (NTI + 1) / 2;
    v4f64 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = v4f64{0.0, 0.0, 0.0, 0.0};
    const int li = lane & 15, lk = lane >> 4;
    bool ok = true;
    const bool brow = lane < D;
    const bool sweeps = wv < c.nseg;
    const int s0 = sweeps ? c.seg[wv] : 0, s1 = sweeps ? c.seg[wv + 1] : 0;
    // ---- forward sweep over this wave's segment ----------------------------------------------------------------------------------------
    // (1) the chain's factor with LANE = ROW (chunks of 64 rows).  The couplings are rank-1, H_p,p-1 = u_p v_p^T, so the Schur complement
    //     of row p is S_p = A_p - alpha_p u_p u_p^T (A_p = H_pp + lambda I) with the SCALAR alpha_p = v_p^T S_{p-1}^-1 v_p and right-hand
    //     side b_p - beta_p u_p; Sherman-Morrison gives S^-1 from A^-1 (wave3_kernel.hip), so all rows factor their A_p and solve for
    //     A^-1 u, A^-1 v, A^-1 b AT ONCE, two scalars run from row to row through DPP (every repetition every lane redoes its own step),
    //     and then all rows AT ONCE factor S_p = G_p G_p^T and form y_p = G_p^-1 (b_p - beta_p u_p), g_{p+1} = G_p^-1 v_{p+1}.
    //     (One row per iteration — every lane redoing the row's 3x3 recurrence from LDS — cost ~1 700 cycles per row.)
    if (sweeps) {
        double cal = 0.0, cbe = 0.0, cg0 = 0.0, cg1 = 0.0, cg2 = 0.0;   // into the chunk's first row: alpha, beta, g
        for (int c0 = s0; c0 < s1; c0 += 64) {
            const int p = c0 + lane;
            const bool live = p < s1;
            const int rows = s1 - c0 < 64 ? s1 - c0 : 64;
            double pk[12], vn0 = 0.0, vn1 = 0.0, vn2 = 0.0;   // H_pp (6), b (3), u (3); the v of row p + 1
#pragma unroll
            for (int k = 0; k < 12; ++k) pk[k] = live ? c.PK[16 * p + k] : 0.0;
            if (p + 1 < s1) { vn0 = c.PK[16 * (p + 1) + 12]; vn1 = c.PK[16 * (p + 1) + 13]; vn2 = c.PK[16 * (p + 1) + 14]; }
            const double u0 = pk[9], u1 = pk[10], u2 = pk[11];
            // A = G_A G_A^T; A^-1 u, A^-1 v_next, A^-1 b and their dot products
            double uu, uv, vvq, ub, vb;
            {
                double a00 = pk[0] + lambda, a10 = pk[1], a11 = pk[2] + lambda, a20 = pk[3], a21 = pk[4], a22 = pk[5] + lambda;
                const double i0 = pivot_rsqrtA(a00);
                a10 *= i0; a20 *= i0;
                a11 = __builtin_fma(-a10, a10, a11); a21 = __builtin_fma(-a20, a10, a21); a22 = __builtin_fma(-a20, a20, a22);
                const double i1 = pivot_rsqrtA(a11);
                a21 *= i1;
                a22 = __builtin_fma(-a21, a21, a22);
                const double i2 = pivot_rsqrtA(a22);
                double P0 = u0 * i0, Q0 = vn0 * i0, R0 = pk[6] * i0;
                double P1 = __builtin_fma(-P0, a10, u1) * i1, Q1 = __builtin_fma(-Q0, a10, vn1) * i1, R1 = __builtin_fma(-R0, a10, pk[7]) * i1;
                double P2 = __builtin_fma(-P1, a21, __builtin_fma(-P0, a20, u2)) * i2, Q2 = __builtin_fma(-Q1, a21, __builtin_fma(-Q0, a20, vn2)) * i2;
                double R2 = __builtin_fma(-R1, a21, __builtin_fma(-R0, a20, pk[8])) * i2;
                // (G_A^-1 applied: u^T A^-1 v = (G_A^-1 u) . (G_A^-1 v), no back-substitution needed for the dot products)
                uu = P0 * P0 + P1 * P1 + P2 * P2; uv = P0 * Q0 + P1 * Q1 + P2 * Q2; vvq = Q0 * Q0 + Q1 * Q1 + Q2 * Q2;
                ub = P0 * R0 + P1 * R1 + P2 * R2; vb = Q0 * R0 + Q1 * R1 + Q2 * R2;
            }
            double ain = 0.0, bin = 0.0, al = 0.0, be = 0.0;
            for (int r = 0; r < rows; ++r) {
                double pal = lane_from_prev(al), pbe = lane_from_prev(be);
                if (lane == 0) { pal = cal; pbe = cbe; }
                ain = pal; bin = pbe;
                const double den = __builtin_fma(-pal, uu, 1.0);
                const double kuv = pal * fast_rcp(den) * uv;
                al = __builtin_fma(kuv, uv, vvq);
                be = __builtin_fma(kuv, __builtin_fma(-pbe, uu, ub), __builtin_fma(-pbe, uv, vb));
            }
            cal = read_lane_dyn(al, rows - 1); cbe = read_lane_dyn(be, rows - 1);
            // S_p = A_p - alpha_p u u^T = G_p G_p^T; y_p; g of the next row
            const double su0 = ain * u0, su1 = ain * u1, su2 = ain * u2;
            double a00 = __builtin_fma(-su0, u0, pk[0] + lambda), a10 = __builtin_fma(-su1, u0, pk[1]), a11 = __builtin_fma(-su1, u1, pk[2] + lambda);
            double a20 = __builtin_fma(-su2, u0, pk[3]), a21 = __builtin_fma(-su2, u1, pk[4]), a22 = __builtin_fma(-su2, u2, pk[5] + lambda);
            const double r0 = __builtin_fma(-u0, bin, pk[6]), r1 = __builtin_fma(-u1, bin, pk[7]), r2 = __builtin_fma(-u2, bin, pk[8]);
            const double ig0 = pivot_rsqrtA(a00);
            a10 *= ig0; a20 *= ig0;
            a11 = __builtin_fma(-a10, a10, a11); a21 = __builtin_fma(-a20, a10, a21); a22 = __builtin_fma(-a20, a20, a22);
            const double ig1 = pivot_rsqrtA(a11);
            a21 *= ig1;
            a22 = __builtin_fma(-a21, a21, a22);
            const double ig2 = pivot_rsqrtA(a22);
            if (__ballot(live && !((ig0 + ig1) + ig2 < DBL_MAX))) ok = false;
            const double y0 = r0 * ig0;
            const double y1 = __builtin_fma(-y0, a10, r1) * ig1;
            const double y2 = __builtin_fma(-y1, a21, __builtin_fma(-y0, a20, r2)) * ig2;
            const double z0 = vn0 * ig0;
            const double z1 = __builtin_fma(-z0, a10, vn1) * ig1;
            const double z2 = __builtin_fma(-z1, a21, __builtin_fma(-z0, a20, vn2)) * ig2;
            double g0 = lane_from_prev(z0), g1 = lane_from_prev(z1), g2 = lane_from_prev(z2);
            if (lane == 0) { g0 = cg0; g1 = cg1; g2 = cg2; }
            cg0 = read_lane_dyn(z0, rows - 1); cg1 = read_lane_dyn(z1, rows - 1); cg2 = read_lane_dyn(z2, rows - 1);
            if (live) {
                double* gz = c.GZ + 12 * p;
                gz[0] = a10; gz[1] = a20; gz[2] = a21; gz[3] = ig0; gz[4] = ig1; gz[5] = ig2; gz[6] = y0; gz[7] = y1; gz[8] = y2;
                gz[9] = g0; gz[10] = g1; gz[11] = g2;
            }
        }
    }
    wsync();
#if defined(LOCAMD_ARROW_TIMING) && LOCAMD_ARROW_TIMING == 3
    AT(2);   // (level 3: slot 2 = the chain's factor alone, slot 6 += the border rows + matrix cores)
#endif
    // (2) the border's rows of the factor, lane r < D = border row r: F_p = (B_p - (F_{p-1} . g_p) u_p^T) G_p^-T, four rows per step
    //     into the f64 matrix cores (P += F F^T)
    {
        double F0 = 0.0, F1 = 0.0, F2 = 0.0, racc = 0.0;
        const double* Brow = c.BB + (size_t)lane * 3;
        const size_t bst = (size_t)D * 3;
        double* FX = c.FX + (size_t)wv * 4 * D16 * 3;
        // B streams from the HBM workspace: the rows of the next PD groups of four are kept in flight (with one group ahead the loop waited
        // an HBM round trip per group: ~5 700 cycles for ~1 300 cycles of work)
        constexpr int PD = 3;
        double nbq[PD][4][3];
        // (unconditional loads from a clamped row — a lane beyond the border reads the neighbouring entries of the workspace — so that
        //  the compiler can count them: with a branch around every load it waited for ALL loads in flight, vmcnt(0), at the first use)
        const int plast = s1 > s0 ? s1 - 1 : s0;
#pragma unroll
        for (int q = 0; q < PD; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pr = s0 + 4 * q + j < s1 ? s0 + 4 * q + j : plast;
#pragma unroll
                for (int k = 0; k < 3; ++k) nbq[q][j][k] = sweeps ? Brow[(size_t)pr * bst + k] : 0.0;
            }
        for (int pq = s0; pq < s1; pq += 4 * PD) {
#pragma unroll
          for (int q = 0; q < PD; ++q) {
            const int p0 = pq + 4 * q;
            if (p0 >= s1) break;
            double cb[4][3];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < 3; ++k) cb[j][k] = (brow && p0 + j < s1) ? nbq[q][j][k] : 0.0;
            // the four rows' factor entries first, all LDS reads in flight together (left to itself the compiler issues each read where
            // the recurrence needs it: ~24 LDS round trips in sequence per group)
            double gzr[4][12], ur[4][3];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pr = p0 + j < s1 ? p0 + j : plast;
#pragma unroll
                for (int k = 0; k < 12; ++k) gzr[j][k] = c.GZ[12 * pr + k];
#pragma unroll
                for (int k = 0; k < 3; ++k) ur[j][k] = c.PK[16 * pr + 9 + k];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = p0 + j;
                double o0 = 0.0, o1 = 0.0, o2 = 0.0;
                if (p < s1) {
                    const double* gz = gzr[j];
                    const double fg = F0 * gz[9] + F1 * gz[10] + F2 * gz[11];
                    const double f0 = __builtin_fma(-fg, ur[j][0], cb[j][0]), f1 = __builtin_fma(-fg, ur[j][1], cb[j][1]), f2 = __builtin_fma(-fg, ur[j][2], cb[j][2]);
                    F0 = f0 * gz[3];
                    F1 = __builtin_fma(-F0, gz[0], f1) * gz[4];
                    F2 = __builtin_fma(-F1, gz[2], __builtin_fma(-F0, gz[1], f2)) * gz[5];
                    racc = __builtin_fma(F0, gz[6], __builtin_fma(F1, gz[7], __builtin_fma(F2, gz[8], racc)));
                    o0 = brow ? F0 : 0.0; o1 = brow ? F1 : 0.0; o2 = brow ? F2 : 0.0;
                }
                if (lane < D16) { double* fx = FX + ((size_t)j * D16 + lane) * 3; fx[0] = o0; fx[1] = o1; fx[2] = o2; }
            }
            // the B rows of the group PD steps ahead, requested HERE: the compiler waits for every load in flight (vmcnt(0)) at the next
            // group's first use of its own rows, so the youngest loads should have the matrix-core phase behind them by then
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pr = p0 + 4 * PD + j < s1 ? p0 + 4 * PD + j : plast;
#pragma unroll
                for (int k = 0; k < 3; ++k) nbq[q][j][k] = Brow[(size_t)pr * bst + k];
            }
            wsync_lds();
            // P += F F^T over these four rows' twelve columns: three k-steps (one per component), lane (row li of a tile, chain row lk)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                double fr[NTI];
#pragma unroll
                for (int ti = 0; ti < NTI; ++ti) fr[ti] = FX[((size_t)lk * D16 + 16 * ti + li) * 3 + i];
                int t = 0;
#pragma unroll
                for (int ti = 0; ti < NTI; ++ti)
#pragma unroll
                    for (int tj = 0; tj <= ti; ++tj) { acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[ti], fr[tj], acc[t], 0, 0, 0); ++t; }
            }
            wsync_lds();
                  }
        }
        if (brow) c.RA[wv * D + lane] = sweeps ? racc : 0.0;
    }
    __syncthreads();
#if defined(LOCAMD_ARROW_TIMING) && LOCAMD_ARROW_TIMING == 3
    AT(6);
#else
    AT(2);
#endif
    // ---- S = C + lambda I - sum of the segments' F F^T (lower triangle, packed); row D = b_b - sum of the segments' F y ---------------
    for (int i = tid; i < D * (D + 1) / 2; i += 64 * ARROW_NW) c.S[i] = c.C0[i];
    for (int r = tid; r < D; r += 64 * ARROW_NW) c.S[tri(D, r)] = c.bB[r] - (((c.RA[r] + c.RA[D + r]) + c.RA[2 * D + r]) + c.RA[3 * D + r]);
    __syncthreads();
    for (int r = tid; r < D; r += 64 * ARROW_NW) c.S[tri(r, r)] += lambda;
    for (int turn = 0; turn < c.nseg; ++turn) {   // (wave order: bit-reproducible)
        if (wv == turn) {
            int t = 0;
#pragma unroll
            for (int ti = 0; ti < NTI; ++ti)
#pragma unroll
                for (int tj = 0; tj <= ti; ++tj) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int R = 16 * ti + lk + 4 * v, C = 16 * tj + li;
                        if (C <= R && R < D) c.S[tri(R, C)] -= acc[t][v];
                    }
                    ++t;
                }
        }
        __syncthreads();
    }
    // ---- dense Cholesky of S by the whole block, one border pose (three columns) per step: every thread factors the step's 3x3
    //      diagonal block itself, one thread per row below solves its row against it, then the trailing update (rank 3) — two
    //      barriers per border pose.  Row D, the right-hand side, leaves the loop as y_b; reciprocal pivots in RA[0 .. D). ----------
    {
        const int ti = tid >> 4, tc = tid & 15;
        for (int J = 0; J < nb; ++J) {
            const int j0 = 3 * J;
            double d00 = c.S[tri(j0, j0)], d10 = c.S[tri(j0 + 1, j0)], d11 = c.S[tri(j0 + 1, j0 + 1)];
            double d20 = c.S[tri(j0 + 2, j0)], d21 = c.S[tri(j0 + 2, j0 + 1)], d22 = c.S[tri(j0 + 2, j0 + 2)];
            const double g0 = pivot_rsqrtA(d00);
            d10 *= g0; d20 *= g0;
            d11 = __builtin_fma(-d10, d10, d11); d21 = __builtin_fma(-d20, d10, d21); d22 = __builtin_fma(-d20, d20, d22);
            const double g1 = pivot_rsqrtA(d11);
            d21 *= g1;
            d22 = __builtin_fma(-d21, d21, d22);
            const double g2 = pivot_rsqrtA(d22);
            ok = ok && ((g0 + g1) + g2 < DBL_MAX);
            __syncthreads();   // (everybody has read the diagonal block)
            if (tid == 0) { c.S[tri(j0 + 1, j0)] = d10; c.S[tri(j0 + 2, j0)] = d20; c.S[tri(j0 + 2, j0 + 1)] = d21; c.RA[j0] = g0; c.RA[j0 + 1] = g1; c.RA[j0 + 2] = g2; }
            if (tid > j0 + 2 && tid <= D) {   // row tid of the panel: w <- w G^-T
                double w0 = c.S[tri(tid, j0)], w1 = c.S[tri(tid, j0 + 1)], w2 = c.S[tri(tid, j0 + 2)];
                w0 *= g0;
                w1 = __builtin_fma(-w0, d10, w1) * g1;
                w2 = __builtin_fma(-w1, d21, __builtin_fma(-w0, d20, w2)) * g2;
                c.S[tri(tid, j0)] = w0; c.S[tri(tid, j0 + 1)] = w1; c.S[tri(tid, j0 + 2)] = w2;
            }
            __syncthreads();
            for (int i = j0 + 3 + ti; i <= D; i += 16) {
                const double a0 = c.S[tri(i, j0)], a1 = c.S[tri(i, j0 + 1)], a2 = c.S[tri(i, j0 + 2)];
                const int cend = i < D ? i : D - 1;
                for (int cc = j0 + 3 + tc; cc <= cend; cc += 16)
                    c.S[tri(i, cc)] = __builtin_fma(-a2, c.S[tri(cc, j0 + 2)], __builtin_fma(-a1, c.S[tri(cc, j0 + 1)], __builtin_fma(-a0, c.S[tri(cc, j0)], c.S[tri(i, cc)])));
            }
            __syncthreads();
        }
    }
    // any segment's or the border's pivot failed?
    {
        __syncthreads();
        if (lane == 0) c.red[8 + wv] = ok ? 0.0 : 1.0;
        __syncthreads();
        ok = (c.red[8] + c.red[9] + c.red[10] + c.red[11]) == 0.0;
    }
    if (!ok) {
        // g2o leaves x alone when the factorisation fails and LM applies that stale x all the same
        double sc = 0.0;
        double* Tn = c.TT + (size_t)(1 - buf) * n * 3;
        const double* To = c.TT + (size_t)buf * n * 3;
        for (int i = tid; i < 3 * n; i += 64 * ARROW_NW) { const double dx = c.XS[i]; sc += dx * (lambda * dx + c.PK[16 * (i / 3) + 6 + i % 3]); Tn[i] = To[i] + dx; }
        for (int r = tid; r < D; r += 64 * ARROW_NW) { const double dx = c.xsB[r]; sc += dx * (lambda * dx + c.bB[r]); c.TB[(1 - buf) * nb * 3 + r] = c.TB[buf * nb * 3 + r] + dx; }
        __threadfence_block();
        scale_sum = block_sum(c, sc);
        return false;
    }
    AT(3);
    // x_b = L^-T y_b (one wave)
    if (wv == 0) {
        // lane = border row, its column of L in registers (all LDS reads in flight together), x_j handed out by v_readlane: one LDS
        // round trip instead of three per column (39 columns: ~23 k cycles before)
        constexpr int DM = 16 * NTI;
        double col[DM];
#pragma unroll
        for (int j = 0; j < DM; ++j) col[j] = (brow && j < D && j > lane) ? c.S[tri(j, lane)] : 0.0;
        double y = brow ? c.S[tri(D, lane)] : 0.0;
        const double rinv = brow ? c.RA[lane] : 0.0;
        double xv = 0.0;
#pragma unroll
        for (int j = DM - 1; j >= 0; --j) {
            if (j < D) {
                const double xj = read_lane_dyn(y * rinv, j);
                if (lane == j) xv = xj;
                y = __builtin_fma(-col[j], xj, y);
            }
        }
        if (brow) c.xB[lane] = xv;
    }
    __syncthreads();
    AT(3);
    // ---- chain: rhs'_q = b_q - B_q^T x_b (one thread per row), then per segment z = L^-1 rhs', x = L^-T z -------------------------------
    for (int q = tid; q < n; q += 64 * ARROW_NW) {
        double t0 = c.PK[16 * q + 6], t1 = c.PK[16 * q + 7], t2 = c.PK[16 * q + 8];
        const double* bp = c.BB + (size_t)q * D * 3;
        // (a thread streams its own row of B: twelve border components per step, their 36 loads in flight together — one component per
        //  step waited an HBM round trip per component)
        int r = 0;
        for (; r + 12 <= D; r += 12) {
            double bv[36], xv[12];
#pragma unroll
            for (int k = 0; k < 36; ++k) bv[k] = bp[3 * r + k];
#pragma unroll
            for (int k = 0; k < 12; ++k) xv[k] = c.xB[r + k];
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                t0 = __builtin_fma(-bv[3 * k], xv[k], t0); t1 = __builtin_fma(-bv[3 * k + 1], xv[k], t1); t2 = __builtin_fma(-bv[3 * k + 2], xv[k], t2);
            }
        }
        for (; r < D; ++r) {
            const double xr = c.xB[r];
            t0 = __builtin_fma(-bp[3 * r], xr, t0); t1 = __builtin_fma(-bp[3 * r + 1], xr, t1); t2 = __builtin_fma(-bp[3 * r + 2], xr, t2);
        }
        c.GZ[12 * q + 6] = t0; c.GZ[12 * q + 7] = t1; c.GZ[12 * q + 8] = t2;
    }
    __syncthreads();
    AT(4);
    if (sweeps) {
        // The two sweeps of a segment with LANE = ROW (chunks of 64 rows): a row's factor, g, u and right-hand side are loaded once
        // into its lane's registers; every repetition every lane takes its neighbour's 3-vector through DPP and redoes its own step —
        // after repetition r the rows 0 .. r of the chunk hold final values (wave3_kernel.hip).  One row per iteration with the row's
        // data read from LDS inside the loop was bound by the LDS round trip of every row (~480 cycles per row and sweep).
        double cz0 = 0.0, cz1 = 0.0, cz2 = 0.0;   // z of the row before this chunk
        for (int c0 = s0; c0 < s1; c0 += 64) {
            const int p = c0 + lane;
            const bool live = p < s1;
            const int rows = s1 - c0 < 64 ? s1 - c0 : 64;
            double G[12], u0 = 0.0, u1 = 0.0, u2 = 0.0;
#pragma unroll
            for (int k = 0; k < 12; ++k) G[k] = live ? c.GZ[12 * p + k] : 0.0;
            if (live) { u0 = c.PK[16 * p + 9]; u1 = c.PK[16 * p + 10]; u2 = c.PK[16 * p + 11]; }
            // z_p = L_p^-1 b_p - (L_p^-1 u_p) (g_p . z_{p-1}): the map (a, b, c) = (-L^-1 u, g, L^-1 b) of every row at once, then the scan
            Map3 m;
            m.c0 = G[6] * G[3];
            m.c1 = __builtin_fma(-m.c0, G[0], G[7]) * G[4];
            m.c2 = __builtin_fma(-m.c1, G[2], __builtin_fma(-m.c0, G[1], G[8])) * G[5];
            const double lu0 = u0 * G[3];
            const double lu1 = __builtin_fma(-lu0, G[0], u1) * G[4];
            const double lu2 = __builtin_fma(-lu1, G[2], __builtin_fma(-lu0, G[1], u2)) * G[5];
            m.a0 = -lu0; m.a1 = -lu1; m.a2 = -lu2;
            m.b0 = G[9]; m.b1 = G[10]; m.b2 = G[11];
            map3_scan(m, lane);
            const double bz = m.b0 * cz0 + m.b1 * cz1 + m.b2 * cz2;   // (every lane's composite starts at the chunk's first row: b = that row's g)
            const double z0 = __builtin_fma(m.a0, bz, m.c0), z1 = __builtin_fma(m.a1, bz, m.c1), z2 = __builtin_fma(m.a2, bz, m.c2);
            if (live) { c.GZ[12 * p + 6] = z0; c.GZ[12 * p + 7] = z1; c.GZ[12 * p + 8] = z2; }
            cz0 = read_lane_dyn(z0, rows - 1); cz1 = read_lane_dyn(z1, rows - 1); cz2 = read_lane_dyn(z2, rows - 1);
        }
        wsync();
        // x_p = G_p^-T (z_p - g_{p+1} (u_{p+1} . x_{p+1})), chunks from the segment's end
        double cx0 = 0.0, cx1 = 0.0, cx2 = 0.0, cu0 = 0.0, cu1 = 0.0, cu2 = 0.0, cg0 = 0.0, cg1 = 0.0, cg2 = 0.0;   // x, u, g of the row after this chunk
        const int nchunks = (s1 - s0 + 63) / 64;
        for (int ch = nchunks - 1; ch >= 0; --ch) {
            const int c0 = s0 + 64 * ch;
            const int rows = s1 - c0 < 64 ? s1 - c0 : 64;
            // lanes in REVERSE row order (lane 0 = the chunk's last row): the backward recurrence is then the same forward scan of maps
            // x_p = U_p^-1 z_p - (U_p^-1 g_{p+1}) (u_{p+1} . x_{p+1}), the row after p sitting in the lane before
            const bool live = lane < rows;
            const int p = c0 + (live ? rows - 1 - lane : 0);
            double G[12], u0 = 0.0, u1 = 0.0, u2 = 0.0;
#pragma unroll
            for (int k = 0; k < 12; ++k) G[k] = live ? c.GZ[12 * p + k] : 0.0;
            if (live) { u0 = c.PK[16 * p + 9]; u1 = c.PK[16 * p + 10]; u2 = c.PK[16 * p + 11]; }
            double nu0 = lane_from_prev(u0), nu1 = lane_from_prev(u1), nu2 = lane_from_prev(u2);
            double ng0 = lane_from_prev(G[9]), ng1 = lane_from_prev(G[10]), ng2 = lane_from_prev(G[11]);
            if (lane == 0) { nu0 = cu0; nu1 = cu1; nu2 = cu2; ng0 = cg0; ng1 = cg1; ng2 = cg2; }
            Map3 m;
            m.c2 = G[8] * G[5];
            m.c1 = __builtin_fma(-G[2], m.c2, G[7]) * G[4];
            m.c0 = __builtin_fma(-G[0], m.c1, __builtin_fma(-G[1], m.c2, G[6])) * G[3];
            const double w2 = ng2 * G[5];
            const double w1 = __builtin_fma(-G[2], w2, ng1) * G[4];
            const double w0 = __builtin_fma(-G[0], w1, __builtin_fma(-G[1], w2, ng0)) * G[3];
            m.a0 = -w0; m.a1 = -w1; m.a2 = -w2;
            m.b0 = nu0; m.b1 = nu1; m.b2 = nu2;
            map3_scan(m, lane);
            const double bx = m.b0 * cx0 + m.b1 * cx1 + m.b2 * cx2;   // (b of every lane's composite: u of the row after the chunk)
            const double x0 = __builtin_fma(m.a0, bx, m.c0), x1 = __builtin_fma(m.a1, bx, m.c1), x2 = __builtin_fma(m.a2, bx, m.c2);
            if (live) { c.GZ[12 * p + 6] = x0; c.GZ[12 * p + 7] = x1; c.GZ[12 * p + 8] = x2; }
            cx0 = read_lane_dyn(x0, rows - 1); cx1 = read_lane_dyn(x1, rows - 1); cx2 = read_lane_dyn(x2, rows - 1);
            cu0 = read_lane_dyn(u0, rows - 1); cu1 = read_lane_dyn(u1, rows - 1); cu2 = read_lane_dyn(u2, rows - 1);
            cg0 = read_lane_dyn(G[9], rows - 1); cg1 = read_lane_dyn(G[10], rows - 1); cg2 = read_lane_dyn(G[11], rows - 1);
        }
    }
    __syncthreads();
    AT(5);
    // ---- the step: t' = t + x (VertexSE3::oplus with R = I), x kept for a later failed solve, computeScale -------------------------
    {
        double sc = 0.0;
        double* Tn = c.TT + (size_t)(1 - buf) * n * 3;
        const double* To = c.TT + (size_t)buf * n * 3;
        for (int i = tid; i < 3 * n; i += 64 * ARROW_NW) {
            const double dx = c.GZ[12 * (i / 3) + 6 + i % 3];
            c.XS[i] = dx; sc += dx * (lambda * dx + c.PK[16 * (i / 3) + 6 + i % 3]); Tn[i] = To[i] + dx;
        }
        for (int r = tid; r < D; r += 64 * ARROW_NW) { const double dx = c.xB[r]; c.xsB[r] = dx; sc += dx * (lambda * dx + c.bB[r]); c.TB[(1 - buf) * nb * 3 + r] = c.TB[buf * nb * 3 + r] + dx; }
        __threadfence_block();
        scale_sum = block_sum(c, sc);
    }
    AT(6);
    return true;
}

template <int JAC, int NTI>
__global__ void __launch_bounds__(64 * ARROW_NW, 1) arrow3_lm_kernel(const WindowArgs a, const ArrowAux x) {
    const int tid = threadIdx.x;
    const long long inst = blockIdx.x;
    const WindowCaps& cp = a.caps;
    ArrowCtx c;
    c.tid = tid; c.lane = tid & 63; c.wv = tid >> 6;
    const int32_t* hdr = x.hdr + inst * 8;
    const int nv = a.counts[inst * 4 + 0];
    c.nb = hdr[0]; c.nseg = hdr[1]; c.n = hdr[2]; c.seg = hdr + 3; c.D = 3 * c.nb; c.rows = c.n + c.nb;
    const int Dm = 3 * x.nb_max, D16m = 16 * ((Dm + 15) / 16);
    c.D16 = D16m;   // (FX rows: the batch's tile count)
    c.jmax = x.jmax; c.jpmax = x.jpmax;
    c.jpk0 = 0; c.jpk1 = 0; c.ppk0 = 0; c.ppk1 = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        c.jpk0 |= (unsigned long long)(x.jch[k] & 255) << (8 * k); c.jpk1 |= (unsigned long long)(x.jch[8 + k] & 255) << (8 * k);
        c.ppk0 |= (unsigned long long)(x.jpch[k] & 255) << (8 * k); c.ppk1 |= (unsigned long long)(x.jpch[8 + k] & 255) << (8 * k);
    }
    const int n = c.n, nb = c.nb, D = c.D;
    {   // LDS carve-up (sized by the batch's capacities: arrow3_lds_doubles)
        double* q = ldsA;
        c.PK = q; q += (size_t)cp.nv_max * 16; c.GZ = q; q += (size_t)cp.nv_max * 12;
        c.C0 = q; q += (size_t)Dm * (Dm + 1) / 2; c.S = q; q += (size_t)(Dm + 1) * (Dm + 2) / 2;
        c.bB = q; q += Dm; c.xB = q; q += Dm; c.xsB = q; q += Dm;
        c.RA = q; q += (size_t)ARROW_NW * Dm;
        c.FX = q; q += (size_t)ARROW_NW * 4 * D16m * 3;
        c.TB = q; q += (size_t)6 * x.nb_max;
        c.red = q; q += 16;
        c.AN = q;
    }
    {
        double* w = x.ws + (size_t)inst * window_arrow3_workspace_doubles_dev(cp, x.nb_max);
        c.TT = w; w += (size_t)cp.nv_max * 6;
        c.XS = w; w += (size_t)cp.nv_max * 3;
        c.BB = w; w += (size_t)cp.nv_max * 3 * x.nb_max * 3;
        c.CS = w;
        c.npad = (size_t)cp.nv_max;
    }
    c.rec = x.rec + (size_t)inst * x.nchunk * x.jmax * 64 * 3;
    c.prec = x.prec + (size_t)inst * x.nchunk * x.jpmax * 64 * 7;
    const int32_t* rslot = x.rslot + (size_t)inst * cp.nv_max;
    const double* gin = a.poses_in + (size_t)inst * cp.nv_max * 12;
    double* gout = a.poses + (size_t)inst * cp.nv_max * 12;
    for (int i = tid; i < 3 * n; i += 64 * ARROW_NW) { c.TT[i] = gin[rslot[i / 3] * 12 + 9 + i % 3]; c.XS[i] = 0.0; }
    for (int r = tid; r < D; r += 64 * ARROW_NW) { c.TB[r] = gin[rslot[n + r / 3] * 12 + 9 + r % 3]; c.xsB[r] = 0.0; }
    for (int i = tid; i < ARROW_NW * Dm; i += 64 * ARROW_NW) c.RA[i] = 0.0;
    for (int i = tid; i < 3 * (a.n_anchors < kArrowMaxAnchors ? a.n_anchors : kArrowMaxAnchors); i += 64 * ARROW_NW) c.AN[i] = a.anchors[i];
    if (tid < 16) c.red[tid] = 0.0;
    __threadfence_block();
    __syncthreads();
    constexpr double tau = 1e-5, good_lo = 1.0 / 3.0, good_hi = 2.0 / 3.0;
    constexpr int max_trials = 10;
    double lambda = 0.0, ni = 2.0, cur_chi = 0.0, last_plain = 0.0;
    int it = 0, trials = 0, terminated = 0, buf = 0, shared_edges = 0;
    AT_DECL;
    const bool active = nv > 0 && a.counts[inst * 4 + 1] + a.counts[inst * 4 + 2] > 0 && a.iterations > 0;
    if (active) {
        for (; it < a.iterations;) {
            double plain, md, chi_lin;
            AT(0);
#ifdef LOCAMD_ARROW_TIMING
            arrow_edges<true, JAC>(a, c, buf, chi_lin, plain, md, shared_edges, at_, at_t);
#else
            arrow_edges<true, JAC>(a, c, buf, chi_lin, plain, md, shared_edges);
#endif
            AT(1);
            cur_chi = chi_lin;
            last_plain = plain;
            if (it == 0) { lambda = tau * md; ni = 2.0; }
            int q = 0;
            double rho = 0.0;
            do {
                double sc;
#ifdef LOCAMD_ARROW_TIMING
                const bool ok = arrow_solve<NTI>(c, lambda, buf, sc, at_, at_t);
#else
                const bool ok = arrow_solve<NTI>(c, lambda, buf, sc);
#endif
                ++trials;
                double temp_chi, plain2, md2;
                int us;
#ifdef LOCAMD_ARROW_TIMING
                arrow_edges<false, JAC>(a, c, 1 - buf, temp_chi, plain2, md2, us, at_, at_t);
#else
                arrow_edges<false, JAC>(a, c, 1 - buf, temp_chi, plain2, md2, us);
#endif
                AT(7);
                last_plain = plain2;
                if (!ok) temp_chi = DBL_MAX;
                rho = (cur_chi - temp_chi) / (sc + 1e-3);
                if (rho > 0.0 && fabs(temp_chi) <= DBL_MAX) {
                    const double r21 = 2.0 * rho - 1.0;
                    double alpha = 1.0 - r21 * r21 * r21;
                    alpha = fmin(alpha, good_hi);
                    lambda *= fmax(good_lo, alpha);
                    ni = 2.0;
                    cur_chi = temp_chi;
                    buf = 1 - buf;   // the trial state is the state
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
    __syncthreads();
    {
        const double* T = c.TT + (size_t)buf * n * 3;
        for (int i = tid; i < 12 * c.rows; i += 64 * ARROW_NW) {
            const int r = i / 12, k = i % 12, slot = rslot[r];
            gout[slot * 12 + k] = k < 9 ? gin[slot * 12 + k] : (r < n ? T[3 * r + k - 9] : c.TB[buf * nb * 3 + 3 * (r - n) + k - 9]);
        }
        if (tid == 0) {
            double* res = a.result + (size_t)inst * 8;
            res[0] = last_plain; res[1] = cur_chi; res[2] = lambda; res[3] = (double)it; res[4] = (double)trials;
            res[5] = (double)terminated; res[6] = (double)shared_edges; res[7] = (double)(2 * 65536 + c.rows) + nb / 16.0;
#ifdef LOCAMD_ARROW_TIMING
            for (int k = 1; k < 8; ++k) res[k] = (double)at_[k];
#endif
        }
    }
}

template <int JAC, int NTI>
hipError_t launch_arrow3_t(const WindowArgs& a, const ArrowAux& x, size_t lds, hipStream_t stream) {
    static std::atomic<uint64_t> attr_set{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_set.load(std::memory_order_acquire) & bit)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&arrow3_lm_kernel<JAC, NTI>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
        if (e != hipSuccess) return e;
        attr_set.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((arrow3_lm_kernel<JAC, NTI>), dim3((unsigned)a.B), dim3(64 * ARROW_NW), lds, stream, a, x);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_window_arrow3(const WindowArgs& a, const ArrowAux& x, hipStream_t stream) {
    if (a.B <= 0 || !x.ws || x.nb_max < 1 || x.nb_max > 15 || x.jmax < 1 || x.jpmax < 1) return hipErrorInvalidValue;
    const size_t lds = window_arrow3_lds_bytes(a.caps, x.nb_max);
    if (lds > 160 * 1024 - 512) return hipErrorInvalidValue;
    const int nti = (3 * x.nb_max + 15) / 16;
    const int sel = (a.jacobian ? 3 : 0) + nti - 1;
    switch (sel) {
        case 0: return launch_arrow3_t<0, 1>(a, x, lds, stream);
        case 1: return launch_arrow3_t<0, 2>(a, x, lds, stream);
        case 2: return launch_arrow3_t<0, 3>(a, x, lds, stream);
        case 3: return launch_arrow3_t<1, 1>(a, x, lds, stream);
        case 4: return launch_arrow3_t<1, 2>(a, x, lds, stream);
        default: return launch_arrow3_t<1, 3>(a, x, lds, stream);
    }
}

}  // namespace locamd
