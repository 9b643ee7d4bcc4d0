// gfx950 (MI355X / CDNA4): TRANSLATION-ONLY windows that are a CHAIN with a small dense BORDER — BASELINE config 4, anchor
// self-calibration: one tag trajectory (a chain of poses tied by the zero-range smoothness edges of Robot::new_vertex,
// robot.cpp:75-110, localization.cpp:338-340) whose every pose ranges to a handful of nodes that are unknowns themselves
// ("every node moves" when topic/relative_range exists, localization.cpp:94-98; addRLRangeEdge, :378-436).  One WAVE per instance.
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
//   forward    one sequential sweep over the chain: G_p = chol(H_pp + lambda I - W_p W_p^T), y_p; the border's rows of the factor ride
//              along, lane = border row: F_p = (B_p - F_{p-1} W_p^T) G_p^-T (three numbers per lane, in registers), r_b -= F_p y_p;
//   Schur      S = C + lambda I - sum_p F_p F_p^T: a (3 nb) x (3 n) x (3 nb) SYRK — the one GEMM-shaped piece, on the f64 matrix
//              cores (v_mfma_f64_16x16x4_f64), four poses per step, accumulators in registers for the whole sweep;
//   border     dense Cholesky of S in LDS (<= 48 x 48), the right-hand side riding along as one more row; back-substitution;
//   chain      z = L^-1 (b_c - B^T x_b), x_c = L^-T z: two more sweeps over the chain that touch only 3x3 data.
// F is never stored (it would be 184 KB per instance and trial); B is (dense 3x3 blocks in an HBM workspace, streamed: 720 bytes
// per chain pose, read twice per trial).  What the sequential sweeps touch lives in LDS (24 doubles per chain pose + the border's
// dense arrays: 68 KB for 256 + 10 poses), what only lane-parallel phases touch (translations, the stale step, edge tables) in HBM
// with coalesced access.  Linearisation and trial scoring are lane-parallel over chain poses through per-pose edge lists the
// host builds once per upload (capi_window.cpp: build_arrow_aux); sums over a border pose's many edges are reduced in one fixed
// order (bit-reproducible, no atomics).
#include "window_kernel.h"
#include "device_math.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>

#include <atomic>

namespace locamd {

// doubles of HBM workspace per instance: translations (two buffers) + stale step, B rows, per-(pose, border) contributions
static __host__ __device__ inline size_t window_arrow3_workspace_doubles_dev(const WindowCaps& c, int nb_max) {
    return (size_t)c.nv_max * 9 + (size_t)c.nv_max * 3 * nb_max * 3 + (size_t)c.nv_max * nb_max * 9;
}
size_t window_arrow3_workspace_doubles(const WindowCaps& c, int nb_max) { return window_arrow3_workspace_doubles_dev(c, nb_max); }
// doubles of LDS per instance
static __host__ __device__ inline size_t arrow3_lds_doubles(int nv_max, int nb_max) {
    const int D = 3 * nb_max, D16 = 16 * ((D + 15) / 16);
    return (size_t)nv_max * 24 + 2 * (size_t)(D + 1) * D + 4 * (size_t)D + 4 * (size_t)D16 * 3 + 6 * (size_t)nb_max + (size_t)nb_max * nb_max / 2 + 8;
}
size_t window_arrow3_lds_bytes(const WindowCaps& c, int nb_max) { return arrow3_lds_doubles(c.nv_max, nb_max) * sizeof(double); }

namespace {

extern __shared__ double ldsA[];
#ifdef LOCAMD_ARROW_TIMING   // diagnostic build: cycle stamps of the phases go to result[1..7] (never benchmarked, never shipped)
#define AT_DECL long long at_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long at_t = clock64()
#define AT(slot) do { const long long t_ = clock64(); at_[slot] += t_ - at_t; at_t = t_; } while (0)
#else
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
__device__ __forceinline__ double pivot_rsqrtA(double d) {   // window_kernel.hip: pivot_rsqrt
    const double y = __builtin_amdgcn_rsq(d);
    const double t = d * y;
    const double e = __builtin_fma(-t, y, 1.0);
    const double pq = __builtin_fma(0.375, e, 0.5);
    const double ye = y * e;
    return __builtin_fma(ye, pq, y);
}

#pragma clang fp contract(off)
__device__ __forceinline__ double norm3_plainA(double dx, double dy, double dz) { return sqrt(dx * dx + dy * dy + dz * dz); }
// g2o's central difference along axis D of endpoint `which` (chain3_kernel.hip: range_jac_numeric3)
template <int D>
__device__ __forceinline__ double range_jac_numericA(const double* p0, const double* p1, int which, double meas) {
    constexpr double delta = 1e-9;
    constexpr double scalar = 1.0 / (2 * delta);
    double a[3] = {p0[0], p0[1], p0[2]}, b[3] = {p1[0], p1[1], p1[2]}, am[3] = {p0[0], p0[1], p0[2]}, bm[3] = {p1[0], p1[1], p1[2]};
    if (which == 0) { a[D] = delta + p0[D]; am[D] = -delta + p0[D]; }
    else { b[D] = delta + p1[D]; bm[D] = -delta + p1[D]; }
    const double ep = meas - norm3_plainA(a[0] - b[0], a[1] - b[1], a[2] - b[2]);
    const double em = meas - norm3_plainA(am[0] - bm[0], am[1] - bm[1], am[2] - bm[2]);
    double bak = ep;
    bak -= em;
    return scalar * bak;
}
#pragma clang fp contract(fast)

// where one instance's arrays live
struct ArrowCtx {
    // LDS (doubles)
    double *HD, *HB, *CU, *CV, *GG, *ZZ;   // per chain pose: 6, 3, 3, 3, 6, 3
    double *C0, *S;                        // (D + 1) x D each (S: row D carries the right-hand side)
    double *bB, *xB, *xsB, *scr;           // D each
    double *FX;                            // [4][D16][3]
    double *TB;                            // [2][nb][3] border translations (state / trial state)
    int* paircnt;                          // [nb][nb] edges per border pair (result[6] bookkeeping)
    // HBM
    double *TT;                            // [2][nv][3] chain + border translations (border entries unused)
    double *XS;                            // [nv][3] the solver's x (stale when a factorisation fails)
    double *BB;                            // [n][D][3]  B_p rows: (border row r, chain component k)
    double *CS;                            // [n][nb][9] per (chain pose, border pose): its share of H_aa (6) and b_a (3)
    const int32_t *e_off, *e_perm, *p_off, *p_perm, *r_idx, *p_idx;
    const double *r_val, *p_val;
    int nv, n, nb, D, D16, nr, np, lane;
};

struct EdgeTerms { double J0[3], J1[3], wr, wre, chi, rho; };

// one EdgeSE3Range between translation p0 (endpoint 0) and p1 (endpoint 1 or a fixed anchor)
template <bool FULL, int JAC>
__device__ __forceinline__ EdgeTerms range_terms(const double* p0, const double* p1, bool moving1, double meas, double info) {
    EdgeTerms t;
    double u[3] = {p0[0] - p1[0], p0[1] - p1[1], p0[2] - p1[2]};
    const double n = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    const double err = JAC == 0 ? meas - n : meas - norm3_plainA(u[0], u[1], u[2]);
    t.chi = err * (info * err);
    const double aux = 1.0 + t.chi;
    t.rho = fast_log_ge1(aux);
    if (FULL) {
        if (JAC == 0) {
            const double inv = n > 0.0 ? 1.0 / n : 0.0;   // coincident endpoints: J = 0, what the central difference gives
            u[0] *= inv; u[1] *= inv; u[2] *= inv;
            t.J0[0] = -u[0]; t.J0[1] = -u[1]; t.J0[2] = -u[2];
            t.J1[0] = moving1 ? u[0] : 0.0; t.J1[1] = moving1 ? u[1] : 0.0; t.J1[2] = moving1 ? u[2] : 0.0;
        } else {
            t.J0[0] = range_jac_numericA<0>(p0, p1, 0, meas);
            t.J0[1] = range_jac_numericA<1>(p0, p1, 0, meas);
            t.J0[2] = range_jac_numericA<2>(p0, p1, 0, meas);
            if (moving1) {
                t.J1[0] = range_jac_numericA<0>(p0, p1, 1, meas);
                t.J1[1] = range_jac_numericA<1>(p0, p1, 1, meas);
                t.J1[2] = range_jac_numericA<2>(p0, p1, 1, meas);
            } else { t.J1[0] = 0.0; t.J1[1] = 0.0; t.J1[2] = 0.0; }
        }
        t.wr = info / aux;
        t.wre = -t.wr * err;
    }
    return t;
}

// Every edge evaluated at translation buffer `buf`: chi sums always; FULL: H, b, B, C as well.  Returns through references the
// robust cost, chi2 over all edges, the largest diagonal entry of H and (FULL) the edges sharing their pair with another edge.
template <bool FULL, int JAC>
__device__ __forceinline__ void arrow_edges(const WindowArgs& a, const ArrowCtx& c, int buf, double& robust_chi, double& plain_chi,
                                            double& max_diag, int& shared_edges) {
    const int lane = c.lane, n = c.n, nb = c.nb, D = c.D;
    const double* T = c.TT + (size_t)buf * c.nv * 3;
    const double* TB = c.TB + buf * nb * 3;
    double rsum = 0.0, csum = 0.0;
    int nshared = 0;
    if (FULL) {
        for (int i = lane; i < (D + 1) * D; i += 64) c.C0[i] = 0.0;
        for (int i = lane; i < D; i += 64) c.bB[i] = 0.0;
        for (int i = lane; i < nb * nb; i += 64) c.paircnt[i] = 0;
        wsync();
    }
    // ---- chain poses, lane-parallel: own edges = to anchors, to border poses, and the edge to the previous chain pose -----------
    for (int p0 = 0; p0 < n; p0 += 64) {
        const int p = p0 + lane;
        if (p < n) {
            double tp[3] = {T[3 * p], T[3 * p + 1], T[3 * p + 2]};
            double hd[6] = {0, 0, 0, 0, 0, 0}, hb[3] = {0, 0, 0}, cu[3] = {0, 0, 0}, cv[3] = {0, 0, 0}, cp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            unsigned mask = 0, dup = 0;
            int ndup = 0;
            const int e0 = c.e_off[p], e1 = c.e_off[p + 1];
            for (int idx = e0; idx < e1; ++idx) {
                const int e = c.e_perm[idx];
                const int v0 = c.r_idx[2 * e], v1 = c.r_idx[2 * e + 1];
                const double meas = c.r_val[5 * e], info = c.r_val[5 * e + 1];
                const bool own0 = v0 == p;
                const int other = own0 ? v1 : v0;
                double po[3];
                if (other < 0) { const double* an = a.anchors + (size_t)(-1 - other) * 3; po[0] = an[0]; po[1] = an[1]; po[2] = an[2]; }
                else if (other < n) { po[0] = T[3 * other]; po[1] = T[3 * other + 1]; po[2] = T[3 * other + 2]; }
                else { const double* tb = TB + 3 * (other - n); po[0] = tb[0]; po[1] = tb[1]; po[2] = tb[2]; }
                const EdgeTerms t = own0 ? range_terms<FULL, JAC>(tp, po, other >= 0, meas, info) : range_terms<FULL, JAC>(po, tp, true, meas, info);
                rsum += t.rho;
                csum += t.chi;
                if (FULL) {
                    double Jo[3], Jx[3];   // own / other endpoint
#pragma unroll
                    for (int k = 0; k < 3; ++k) { Jo[k] = own0 ? t.J0[k] : t.J1[k]; Jx[k] = own0 ? t.J1[k] : t.J0[k]; }
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
#pragma unroll
                        for (int cc = 0; cc <= r; ++cc) hd[r * (r + 1) / 2 + cc] += t.wr * Jo[r] * Jo[cc];
                        hb[r] += Jo[r] * t.wre;
                    }
                    if (other >= n) {
                        const int ab = other - n;
                        double* bb = c.BB + ((size_t)p * D + 3 * ab) * 3;
                        double* cs = c.CS + ((size_t)p * nb + ab) * 9;
                        const bool again = (mask >> ab) & 1u;
                        double blk[9], own9[9];
#pragma unroll
                        for (int i = 0; i < 3; ++i)
#pragma unroll
                            for (int k = 0; k < 3; ++k) blk[3 * i + k] = t.wr * Jx[i] * Jo[k];   // row = border component, column = chain component
#pragma unroll
                        for (int r = 0; r < 3; ++r) {
#pragma unroll
                            for (int cc = 0; cc <= r; ++cc) own9[r * (r + 1) / 2 + cc] = t.wr * Jx[r] * Jx[cc];
                            own9[6 + r] = Jx[r] * t.wre;
                        }
                        if (again) {
#pragma unroll
                            for (int k = 0; k < 9; ++k) { bb[k] += blk[k]; cs[k] += own9[k]; }
                            ndup += 1;
                            dup |= 1u << ab;
                        } else {
#pragma unroll
                            for (int k = 0; k < 9; ++k) { bb[k] = blk[k]; cs[k] = own9[k]; }
                        }
                        mask |= 1u << ab;
                    } else if (other >= 0) {   // the edge to the previous chain pose (one per pair: checked on the host)
#pragma unroll
                        for (int k = 0; k < 3; ++k) { cu[k] = t.wr * Jo[k]; cv[k] = Jx[k]; }
#pragma unroll
                        for (int r = 0; r < 3; ++r) {
#pragma unroll
                            for (int cc = 0; cc <= r; ++cc) cp[r * (r + 1) / 2 + cc] = t.wr * Jx[r] * Jx[cc];
                            cp[6 + r] = Jx[r] * t.wre;
                        }
                    }
                }
            }
            const int q0 = c.p_off[p], q1 = c.p_off[p + 1];
            for (int idx = q0; idx < q1; ++idx) {   // priors: e = t + Z^-1.t, diagonal translation information, no robust kernel
                const int e = c.p_perm[idx];
                double chi = 0.0;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double er = tp[k] + c.p_val[18 * e + 9 + k], wd = c.p_val[18 * e + 12 + k];
                    chi += er * (wd * er);
                    if (FULL) { hd[k * (k + 1) / 2 + k] += wd; hb[k] += -wd * er; }
                }
                rsum += chi;
                csum += chi;
            }
            if (FULL) {
                for (int ab = 0; ab < nb; ++ab)
                    if (!((mask >> ab) & 1u)) {
                        double* bb = c.BB + ((size_t)p * D + 3 * ab) * 3;
                        double* cs = c.CS + ((size_t)p * nb + ab) * 9;
#pragma unroll
                        for (int k = 0; k < 9; ++k) { bb[k] = 0.0; cs[k] = 0.0; }
                    }
                nshared += ndup + __popc(dup);
#pragma unroll
                for (int k = 0; k < 6; ++k) { c.HD[6 * p + k] = hd[k]; c.GG[6 * p + k] = cp[k]; }
#pragma unroll
                for (int k = 0; k < 3; ++k) { c.HB[3 * p + k] = hb[k]; c.CU[3 * p + k] = cu[k]; c.CV[3 * p + k] = cv[k]; c.ZZ[3 * p + k] = cp[6 + k]; }
            }
        }
    }
    if (FULL) {
        __threadfence_block();
        wsync();
        // the edge (p, p + 1) also belongs to pose p: its share was left in pose p + 1's (G, z) slots
        for (int p0 = 0; p0 < n; p0 += 64) {
            const int p = p0 + lane;
            if (p + 1 < n) {
#pragma unroll
                for (int k = 0; k < 6; ++k) c.HD[6 * p + k] += c.GG[6 * (p + 1) + k];
#pragma unroll
                for (int k = 0; k < 3; ++k) c.HB[3 * p + k] += c.ZZ[3 * (p + 1) + k];
            }
        }
        // a border pose's diagonal block and b: the sum over the chain of the shares left in CS, one fixed order (four partial sums)
        for (int t0 = 0; t0 < nb * 9; t0 += 64) {
            const int t = t0 + lane;
            if (t < nb * 9) {
                double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
                const double* col = c.CS + t;
                const size_t st = (size_t)nb * 9;
                int p = 0;
                for (; p + 3 < n; p += 4) { s0 += col[p * st]; s1 += col[(p + 1) * st]; s2 += col[(p + 2) * st]; s3 += col[(p + 3) * st]; }
                for (; p < n; ++p) s0 += col[p * st];
                const double s = (s0 + s1) + (s2 + s3);
                const int ab = t / 9, k = t % 9;
                if (k < 6) {
                    const int r = k < 1 ? 0 : (k < 3 ? 1 : 2), cc = k - r * (r + 1) / 2;
                    c.C0[(3 * ab + r) * D + 3 * ab + cc] = s;
                } else {
                    c.bB[3 * ab + k - 6] = s;
                }
            }
        }
        wsync();
    }
    // ---- edges and priors owned by border poses (border-border ranges, border-anchor ranges, priors): few; one lane, in order -----
    if (lane == 0) {
        for (int o = n; o < c.nv; ++o) {
            const int ob = o - n;
            const double to[3] = {TB[3 * ob], TB[3 * ob + 1], TB[3 * ob + 2]};
            for (int idx = c.e_off[o]; idx < c.e_off[o + 1]; ++idx) {
                const int e = c.e_perm[idx];
                const int v0 = c.r_idx[2 * e], v1 = c.r_idx[2 * e + 1];
                const double meas = c.r_val[5 * e], info = c.r_val[5 * e + 1];
                const bool own0 = v0 == o;
                const int other = own0 ? v1 : v0;
                double po[3];
                if (other < 0) { const double* an = a.anchors + (size_t)(-1 - other) * 3; po[0] = an[0]; po[1] = an[1]; po[2] = an[2]; }
                else { const double* tb = TB + 3 * (other - n); po[0] = tb[0]; po[1] = tb[1]; po[2] = tb[2]; }
                const EdgeTerms t = own0 ? range_terms<FULL, JAC>(to, po, other >= 0, meas, info) : range_terms<FULL, JAC>(po, to, true, meas, info);
                rsum += t.rho;
                csum += t.chi;
                if (FULL) {
                    double Jo[3], Jx[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) { Jo[k] = own0 ? t.J0[k] : t.J1[k]; Jx[k] = own0 ? t.J1[k] : t.J0[k]; }
                    for (int r = 0; r < 3; ++r) {
                        for (int cc = 0; cc <= r; ++cc) c.C0[(3 * ob + r) * D + 3 * ob + cc] += t.wr * Jo[r] * Jo[cc];
                        c.bB[3 * ob + r] += Jo[r] * t.wre;
                    }
                    if (other >= 0) {   // the other border pose has the smaller slot (the owner is the later one)
                        const int xb = other - n;
                        for (int r = 0; r < 3; ++r) {
                            for (int cc = 0; cc <= r; ++cc) c.C0[(3 * xb + r) * D + 3 * xb + cc] += t.wr * Jx[r] * Jx[cc];
                            c.bB[3 * xb + r] += Jx[r] * t.wre;
                            for (int cc = 0; cc < 3; ++cc) c.C0[(3 * ob + r) * D + 3 * xb + cc] += t.wr * Jo[r] * Jx[cc];
                        }
                        const int cnt = ++c.paircnt[ob * nb + xb];
                        if (cnt == 2) nshared += 2; else if (cnt > 2) nshared += 1;
                    }
                }
            }
            for (int idx = c.p_off[o]; idx < c.p_off[o + 1]; ++idx) {
                const int e = c.p_perm[idx];
                double chi = 0.0;
                for (int k = 0; k < 3; ++k) {
                    const double er = to[k] + c.p_val[18 * e + 9 + k], wd = c.p_val[18 * e + 12 + k];
                    chi += er * (wd * er);
                    if (FULL) { c.C0[(3 * ob + k) * D + 3 * ob + k] += wd; c.bB[3 * ob + k] += -wd * er; }
                }
                rsum += chi;
                csum += chi;
            }
        }
    }
    wsync();
    robust_chi = wave_sum(rsum);
    plain_chi = wave_sum(csum);
    if (FULL) {
        double md = 0.0;
        for (int p = lane; p < n; p += 64) md = fmax(md, fmax(fabs(c.HD[6 * p]), fmax(fabs(c.HD[6 * p + 2]), fabs(c.HD[6 * p + 5]))));
        for (int r = lane; r < D; r += 64) md = fmax(md, fabs(c.C0[r * D + r]));
        max_diag = wave_max(md);
        shared_edges = (int)wave_sum((double)nshared);
    }
}

// (H + lambda I) x = b, the step applied to the other translation buffer.  Returns false when a pivot fails (then the stale x is
// applied: SURVEY A.6); scale_sum = g2o's computeScale sum.
template <int NTI>
__device__ __forceinline__ bool arrow_solve(const ArrowCtx& c, double lambda, int buf, double& scale_sum
#ifdef LOCAMD_ARROW_TIMING
                                            , long long* at_, long long& at_t
#endif
                                            ) {
    const int lane = c.lane, n = c.n, nb = c.nb, D = c.D, D16 = c.D16;
    constexpr int NACC = NTI * (NTI + 1) / 2;
    v4f64 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = v4f64{0.0, 0.0, 0.0, 0.0};
    const int li = lane & 15, lk = lane >> 4;
    // ---- forward sweep over the chain (every lane runs the 3x3 recurrence; lane r < D carries border row r) -----------------------
    double l10 = 0.0, l20 = 0.0, l21 = 0.0, ig0 = 0.0, ig1 = 0.0, ig2 = 0.0, y0 = 0.0, y1 = 0.0, y2 = 0.0;
    double F0 = 0.0, F1 = 0.0, F2 = 0.0, racc = 0.0;
    bool ok = true;
    const bool brow = lane < D;
    const double* Brow = c.BB + (size_t)lane * 3;
    const size_t bst = (size_t)D * 3;
    double nb_[4][3];   // the next four poses' B rows of this lane, in flight
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) nb_[j][k] = (brow && j < n) ? Brow[(size_t)j * bst + k] : 0.0;
    for (int p0 = 0; p0 < n; p0 += 4) {
        double cb[4][3];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) cb[j][k] = nb_[j][k];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) nb_[j][k] = (brow && p0 + 4 + j < n) ? Brow[(size_t)(p0 + 4 + j) * bst + k] : 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = p0 + j;
            double o0 = 0.0, o1 = 0.0, o2 = 0.0;
            if (p < n) {
                const double* hd = c.HD + 6 * p;
                const double u0 = c.CU[3 * p], u1 = c.CU[3 * p + 1], u2 = c.CU[3 * p + 2];
                const double v0 = c.CV[3 * p], v1 = c.CV[3 * p + 1], v2 = c.CV[3 * p + 2];
                // g = G_{p-1}^-1 v: W_p = u g^T
                const double g0 = v0 * ig0;
                const double g1 = __builtin_fma(-g0, l10, v1) * ig1;
                const double g2 = __builtin_fma(-g1, l21, __builtin_fma(-g0, l20, v2)) * ig2;
                const double gg = g0 * g0 + g1 * g1 + g2 * g2, gy = g0 * y0 + g1 * y1 + g2 * y2;
                const double su0 = gg * u0, su1 = gg * u1, su2 = gg * u2;
                double a00 = __builtin_fma(-su0, u0, hd[0] + lambda), a10 = __builtin_fma(-su1, u0, hd[1]), a11 = __builtin_fma(-su1, u1, hd[2] + lambda);
                double a20 = __builtin_fma(-su2, u0, hd[3]), a21 = __builtin_fma(-su2, u1, hd[4]), a22 = __builtin_fma(-su2, u2, hd[5] + lambda);
                double r0 = __builtin_fma(-u0, gy, c.HB[3 * p]), r1 = __builtin_fma(-u1, gy, c.HB[3 * p + 1]), r2 = __builtin_fma(-u2, gy, c.HB[3 * p + 2]);
                // border row: f = B_p[r] - (F_{p-1}[r] . g) u   (with the PREVIOUS pose's F), then F = f G_p^-T
                const double fg = F0 * g0 + F1 * g1 + F2 * g2;
                const double f0 = __builtin_fma(-fg, u0, cb[j][0]), f1 = __builtin_fma(-fg, u1, cb[j][1]), f2 = __builtin_fma(-fg, u2, cb[j][2]);
                // 3x3 Cholesky (right-looking, reciprocal square roots of the pivots)
                ig0 = pivot_rsqrtA(a00);
                a10 *= ig0; a20 *= ig0;
                a11 = __builtin_fma(-a10, a10, a11); a21 = __builtin_fma(-a20, a10, a21); a22 = __builtin_fma(-a20, a20, a22);
                ig1 = pivot_rsqrtA(a11);
                a21 *= ig1;
                a22 = __builtin_fma(-a21, a21, a22);
                ig2 = pivot_rsqrtA(a22);
                ok = ok && ((ig0 + ig1) + ig2 < DBL_MAX);
                l10 = a10; l20 = a20; l21 = a21;
                y0 = r0 * ig0;
                y1 = __builtin_fma(-y0, l10, r1) * ig1;
                y2 = __builtin_fma(-y1, l21, __builtin_fma(-y0, l20, r2)) * ig2;
                F0 = f0 * ig0;
                F1 = __builtin_fma(-F0, l10, f1) * ig1;
                F2 = __builtin_fma(-F1, l21, __builtin_fma(-F0, l20, f2)) * ig2;
                racc = __builtin_fma(F0, y0, __builtin_fma(F1, y1, __builtin_fma(F2, y2, racc)));
                if (lane == 0) {
                    double* gg_ = c.GG + 6 * p;
                    gg_[0] = l10; gg_[1] = l20; gg_[2] = l21; gg_[3] = ig0; gg_[4] = ig1; gg_[5] = ig2;
                    c.ZZ[3 * p] = y0; c.ZZ[3 * p + 1] = y1; c.ZZ[3 * p + 2] = y2;
                }
                o0 = brow ? F0 : 0.0; o1 = brow ? F1 : 0.0; o2 = brow ? F2 : 0.0;
            }
            if (lane < D16) { double* fx = c.FX + ((size_t)j * D16 + lane) * 3; fx[0] = o0; fx[1] = o1; fx[2] = o2; }
        }
        wsync();
        // P += F F^T over these four poses' twelve columns: three k-steps (one per component), lane (row li of a tile, pose lk)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double fr[NTI];
#pragma unroll
            for (int ti = 0; ti < NTI; ++ti) fr[ti] = c.FX[((size_t)lk * D16 + 16 * ti + li) * 3 + i];
            int t = 0;
#pragma unroll
            for (int ti = 0; ti < NTI; ++ti)
#pragma unroll
                for (int tj = 0; tj <= ti; ++tj) { acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[ti], fr[tj], acc[t], 0, 0, 0); ++t; }
        }
        wsync();
    }
    AT(2);
    // ---- S = C + lambda I - P (lower triangle); row D = the border's right-hand side b_b - sum F y --------------------------------
    {
        int t = 0;
#pragma unroll
        for (int ti = 0; ti < NTI; ++ti)
#pragma unroll
            for (int tj = 0; tj <= ti; ++tj) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int R = 16 * ti + lk + 4 * v, C = 16 * tj + li;
                    if (C <= R && R < D) c.S[R * D + C] = c.C0[R * D + C] + (R == C ? lambda : 0.0) - acc[t][v];
                }
                ++t;
            }
        if (brow) c.S[D * D + lane] = c.bB[lane] - racc;
    }
    wsync();
    // ---- dense Cholesky of S, lane = row (row D: the right-hand side, which leaves the loop as y_b) ---------------------------------
    for (int j = 0; j < D; ++j) {
        const double g = pivot_rsqrtA(c.S[j * D + j]);
        ok = ok && (g < DBL_MAX);
        double lij = 0.0;
        if (lane > j && lane <= D) { lij = c.S[lane * D + j] * g; c.S[lane * D + j] = lij; }
        if (lane == j) c.S[j * D + j] = g;   // (the diagonal keeps the reciprocal pivot)
        wsync();
        if (lane > j && lane <= D) {
            const int cend = lane < D ? lane : D - 1;
            for (int cc = j + 1; cc <= cend; ++cc) c.S[lane * D + cc] = __builtin_fma(-lij, c.S[cc * D + j], c.S[lane * D + cc]);
        }
        wsync();
    }
    if (!ok) {
        // g2o leaves x alone when the factorisation fails and LM applies that stale x all the same
        double sc = 0.0;
        double* Tn = c.TT + (size_t)(1 - buf) * c.nv * 3;
        const double* To = c.TT + (size_t)buf * c.nv * 3;
        for (int i = lane; i < 3 * n; i += 64) { const double dx = c.XS[i]; sc += dx * (lambda * dx + c.HB[i]); Tn[i] = To[i] + dx; }
        for (int r = lane; r < D; r += 64) { const double dx = c.xsB[r]; sc += dx * (lambda * dx + c.bB[r]); c.TB[(1 - buf) * nb * 3 + r] = c.TB[buf * nb * 3 + r] + dx; }
        __threadfence_block();
        wsync();
        scale_sum = wave_sum(sc);
        return false;
    }
    // x_b = L^-T y_b
    AT(3);
    if (brow) c.xB[lane] = c.S[D * D + lane];
    wsync();
    for (int j = D - 1; j >= 0; --j) {
        const double xj = c.xB[j] * c.S[j * D + j];
        wsync();
        if (lane == j) c.xB[j] = xj;
        if (lane < j) c.xB[lane] = __builtin_fma(-c.S[j * D + lane], xj, c.xB[lane]);
        wsync();
    }
    AT(3);
    // ---- chain: rhs'_p = b_p - B_p^T x_b (lane-parallel), then z = L^-1 rhs', x = L^-T z (two sweeps) -----------------------------
    for (int p0 = 0; p0 < n; p0 += 64) {
        const int p = p0 + lane;
        if (p < n) {
            double s0 = c.HB[3 * p], s1 = c.HB[3 * p + 1], s2 = c.HB[3 * p + 2];
            const double* bp = c.BB + (size_t)p * bst;
            for (int r = 0; r < D; ++r) {
                const double xr = c.xB[r];
                s0 = __builtin_fma(-bp[3 * r], xr, s0); s1 = __builtin_fma(-bp[3 * r + 1], xr, s1); s2 = __builtin_fma(-bp[3 * r + 2], xr, s2);
            }
            c.ZZ[3 * p] = s0; c.ZZ[3 * p + 1] = s1; c.ZZ[3 * p + 2] = s2;
        }
    }
    wsync();
    AT(4);
    {
        double pl10 = 0.0, pl20 = 0.0, pl21 = 0.0, pi0 = 0.0, pi1 = 0.0, pi2 = 0.0, z0 = 0.0, z1 = 0.0, z2 = 0.0;
        for (int p = 0; p < n; ++p) {
            const double* gg_ = c.GG + 6 * p;
            const double v0 = c.CV[3 * p], v1 = c.CV[3 * p + 1], v2 = c.CV[3 * p + 2];
            const double g0 = v0 * pi0;
            const double g1 = __builtin_fma(-g0, pl10, v1) * pi1;
            const double g2 = __builtin_fma(-g1, pl21, __builtin_fma(-g0, pl20, v2)) * pi2;
            const double gz = g0 * z0 + g1 * z1 + g2 * z2;
            const double r0 = __builtin_fma(-c.CU[3 * p], gz, c.ZZ[3 * p]), r1 = __builtin_fma(-c.CU[3 * p + 1], gz, c.ZZ[3 * p + 1]), r2 = __builtin_fma(-c.CU[3 * p + 2], gz, c.ZZ[3 * p + 2]);
            pl10 = gg_[0]; pl20 = gg_[1]; pl21 = gg_[2]; pi0 = gg_[3]; pi1 = gg_[4]; pi2 = gg_[5];
            z0 = r0 * pi0;
            z1 = __builtin_fma(-z0, pl10, r1) * pi1;
            z2 = __builtin_fma(-z1, pl21, __builtin_fma(-z0, pl20, r2)) * pi2;
            if (lane == 0) { c.ZZ[3 * p] = z0; c.ZZ[3 * p + 1] = z1; c.ZZ[3 * p + 2] = z2; }
        }
        wsync();
        // x_p = G_p^-T (z_p - g_{p+1} (u_{p+1} . x_{p+1})), g_{p+1} = G_p^-1 v_{p+1}
        double x0 = 0.0, x1 = 0.0, x2 = 0.0, nu0 = 0.0, nu1 = 0.0, nu2 = 0.0, nv0 = 0.0, nv1 = 0.0, nv2 = 0.0;
        for (int p = n - 1; p >= 0; --p) {
            const double* gg_ = c.GG + 6 * p;
            const double a10 = gg_[0], a20 = gg_[1], a21 = gg_[2], i0 = gg_[3], i1 = gg_[4], i2 = gg_[5];
            const double g0 = nv0 * i0;
            const double g1 = __builtin_fma(-g0, a10, nv1) * i1;
            const double g2 = __builtin_fma(-g1, a21, __builtin_fma(-g0, a20, nv2)) * i2;
            const double ux = nu0 * x0 + nu1 * x1 + nu2 * x2;
            double t0 = __builtin_fma(-g0, ux, c.ZZ[3 * p]), t1 = __builtin_fma(-g1, ux, c.ZZ[3 * p + 1]), t2 = __builtin_fma(-g2, ux, c.ZZ[3 * p + 2]);
            x2 = t2 * i2;
            t1 = __builtin_fma(-a21, x2, t1); t0 = __builtin_fma(-a20, x2, t0);
            x1 = t1 * i1;
            t0 = __builtin_fma(-a10, x1, t0);
            x0 = t0 * i0;
            nu0 = c.CU[3 * p]; nu1 = c.CU[3 * p + 1]; nu2 = c.CU[3 * p + 2];
            nv0 = c.CV[3 * p]; nv1 = c.CV[3 * p + 1]; nv2 = c.CV[3 * p + 2];
            if (lane == 0) { c.ZZ[3 * p] = x0; c.ZZ[3 * p + 1] = x1; c.ZZ[3 * p + 2] = x2; }
        }
        wsync();
    }
    AT(5);
    // ---- the step: t' = t + x (VertexSE3::oplus with R = I), x kept for a later failed solve, computeScale -------------------------
    {
        double sc = 0.0;
        double* Tn = c.TT + (size_t)(1 - buf) * c.nv * 3;
        const double* To = c.TT + (size_t)buf * c.nv * 3;
        for (int i = lane; i < 3 * n; i += 64) { const double dx = c.ZZ[i]; c.XS[i] = dx; sc += dx * (lambda * dx + c.HB[i]); Tn[i] = To[i] + dx; }
        for (int r = lane; r < D; r += 64) { const double dx = c.xB[r]; c.xsB[r] = dx; sc += dx * (lambda * dx + c.bB[r]); c.TB[(1 - buf) * nb * 3 + r] = c.TB[buf * nb * 3 + r] + dx; }
        __threadfence_block();
        wsync();
        scale_sum = wave_sum(sc);
    }
    AT(6);
    return true;
}

template <int JAC, int NTI>
__global__ void __launch_bounds__(64, 1) arrow3_lm_kernel(const WindowArgs a, const ArrowAux x) {
    const int lane = threadIdx.x;
    const long long inst = blockIdx.x;
    const WindowCaps& cp = a.caps;
    ArrowCtx c;
    c.lane = lane;
    c.nv = a.counts[inst * 4 + 0]; c.nr = a.counts[inst * 4 + 1]; c.np = a.counts[inst * 4 + 2];
    c.nb = x.nb[inst]; c.n = c.nv - c.nb; c.D = 3 * c.nb; c.D16 = 16 * ((3 * x.nb_max + 15) / 16);   // (FX rows: the batch's tile count)
    const int nv = c.nv, n = c.n, nb = c.nb, D = c.D;
    {   // LDS carve-up (sized by the batch's capacities: arrow3_lds_doubles)
        double* q = ldsA;
        c.HD = q; q += (size_t)cp.nv_max * 6; c.HB = q; q += (size_t)cp.nv_max * 3; c.CU = q; q += (size_t)cp.nv_max * 3;
        c.CV = q; q += (size_t)cp.nv_max * 3; c.GG = q; q += (size_t)cp.nv_max * 6; c.ZZ = q; q += (size_t)cp.nv_max * 3;
        const int Dm = 3 * x.nb_max, D16m = 16 * ((Dm + 15) / 16);
        c.C0 = q; q += (size_t)(Dm + 1) * Dm; c.S = q; q += (size_t)(Dm + 1) * Dm;
        c.bB = q; q += Dm; c.xB = q; q += Dm; c.xsB = q; q += Dm; c.scr = q; q += Dm;
        c.FX = q; q += (size_t)4 * D16m * 3;
        c.TB = q; q += (size_t)6 * x.nb_max;
        c.paircnt = reinterpret_cast<int*>(q);
    }
    {
        double* w = x.ws + (size_t)inst * window_arrow3_workspace_doubles_dev(cp, x.nb_max);
        c.TT = w; w += (size_t)cp.nv_max * 6;
        c.XS = w; w += (size_t)cp.nv_max * 3;
        c.BB = w; w += (size_t)cp.nv_max * 3 * x.nb_max * 3;
        c.CS = w;
    }
    c.e_off = x.e_off + (size_t)inst * (cp.nv_max + 1); c.e_perm = x.e_perm + (size_t)inst * cp.nr_max;
    c.p_off = x.p_off + (size_t)inst * (cp.nv_max + 1); c.p_perm = x.p_perm + (size_t)inst * cp.np_max;
    c.r_idx = a.r_idx + (size_t)inst * cp.nr_max * 2; c.r_val = a.r_val + (size_t)inst * cp.nr_max * 5;
    c.p_idx = a.p_idx + (size_t)inst * cp.np_max; c.p_val = a.p_val + (size_t)inst * cp.np_max * 18;
    // TT is laid out [2][nv][3] with the INSTANCE's nv
    const double* gin = a.poses_in + (size_t)inst * cp.nv_max * 12;
    double* gout = a.poses + (size_t)inst * cp.nv_max * 12;
    for (int i = lane; i < 3 * nv; i += 64) { c.TT[i] = gin[(i / 3) * 12 + 9 + i % 3]; c.XS[i] = 0.0; }
    for (int r = lane; r < D; r += 64) { c.TB[r] = gin[(n + r / 3) * 12 + 9 + r % 3]; c.xsB[r] = 0.0; }
    __threadfence_block();
    wsync();
    constexpr double tau = 1e-5, good_lo = 1.0 / 3.0, good_hi = 2.0 / 3.0;
    constexpr int max_trials = 10;
    double lambda = 0.0, ni = 2.0, cur_chi = 0.0, last_plain = 0.0;
    int it = 0, trials = 0, terminated = 0, buf = 0, shared_edges = 0;
    AT_DECL;
    const bool active = nv > 0 && c.nr + c.np > 0 && a.iterations > 0;
    if (active) {
        for (; it < a.iterations;) {
            double plain, md, chi_lin;
            AT(0);
            arrow_edges<true, JAC>(a, c, buf, chi_lin, plain, md, shared_edges);
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
                arrow_edges<false, JAC>(a, c, 1 - buf, temp_chi, plain2, md2, us);
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
    wsync();
    {
        const double* T = c.TT + (size_t)buf * nv * 3;
        for (int i = lane; i < 12 * nv; i += 64) {
            const int p = i / 12, k = i % 12;
            gout[i] = k < 9 ? gin[i] : (p < n ? T[3 * p + k - 9] : c.TB[buf * nb * 3 + 3 * (p - n) + k - 9]);
        }
        if (lane == 0) {
            double* res = a.result + (size_t)inst * 8;
            res[0] = last_plain; res[1] = cur_chi; res[2] = lambda; res[3] = (double)it; res[4] = (double)trials;
            res[5] = (double)terminated; res[6] = (double)shared_edges; res[7] = (double)(2 * 65536 + n + nb) + nb / 16.0;
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
    hipLaunchKernelGGL((arrow3_lm_kernel<JAC, NTI>), dim3((unsigned)a.B), dim3(64), lds, stream, a, x);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_window_arrow3(const WindowArgs& a, const ArrowAux& x, hipStream_t stream) {
    if (a.B <= 0 || !x.ws || x.nb_max < 1 || x.nb_max > 16) return hipErrorInvalidValue;
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
