// gfx950 (MI355X / CDNA4): TRANSLATION-ONLY chain windows, one WAVE per window — the drop-in node's own solve (one window per
// range message, cfg/uwb_only.yaml) and batches of such windows of any size.
//
// What it solves (reference file:line): the graph Localization::addRangeEdge builds per range message
// (localization.cpp:297-376) — per pose one EdgeSE3Range to an anchor (:331) and the zero-range smoothness edge to the previous
// pose (:338-340), Cauchy kernels (:608-627) — solved by Localization::solve() = g2o Levenberg-Marquardt (:164-170), chi2() (:197).
// The 3-DoF form is exact for these graphs (chain3_kernel.hip's header: identity antenna offsets, identity rotations, no rotation
// information — every dropped term of the 6x6-block system is an exact zero); the host takes this kernel under the same
// conditions as chain3_lm_kernel, for windows of <= 64 poses (capi_window.cpp: pick_kernel).
//
// MI355X mapping.  A single window is a latency problem: one wave issues one f64 instruction per 4 .. 8 cycles whatever its 64
// lanes hold, so the general wave-per-window kernel's 6x6-block schedule (~9 k instructions per LM iteration on a ten-pose window,
// 0.36 ms per solve) is cut to what this graph needs — 0.056 ms:
//   * lane = EDGE for everything per edge (residual, robust weight, the twelve perturbed norms of g2o's numeric Jacobian): one
//     pass whatever the edge count (<= 64 per pass; the first 64 edges stay in registers); an edge writes one record per moving
//     endpoint (w, -w e, its J, the other endpoint's J when that is the previous pose) into the slot of that pose's list;
//   * lane = POSE for the normal equations: pose p sums its records (host order = g2o's accumulation order) into H_pp (6), b_p (3)
//     and the coupling with pose p - 1, all in that lane's registers for every trial of the iteration;
//   * the block-tridiagonal solve: with one edge per consecutive pair the couplings are rank-1, H_p,p-1 = u v^T, the Schur complement
//     of pose p is A_p - alpha_p u u^T, and Sherman-Morrison turns the sequential elimination into a recurrence on TWO SCALARS per
//     pose (see "The solve" in the kernel); every repetition every lane takes its neighbour's scalars through DPP (wave_shr:1 /
//     wave_shl:1) and redoes its own step — no exec masking, no LDS, no v_readlane round trips.  Windows with several edges on a pair
//     keep 3x3 blocks handed over the same way;
//   * LM's next trials are solved SPECULATIVELY in the idle lanes: up to four groups of 16 lanes solve the same H with the lambdas
//     of this trial and of the next ones (what LM would try after a rejection), and are scored in LM's order;
//   * nothing but the poses and the per-edge records ever leaves registers; no barriers (one wave per workgroup), no HBM
//     workspace, 6 KB of LDS for a ten-pose window; 168 registers: three waves per SIMD in batches (3.7e7 windows/s).
#include "window_kernel.h"
#include "device_math.h"

#include <float.h>
#include <math.h>

namespace locamd {

namespace {

extern __shared__ double w3lds[];

// -DLOCAMD_WAVE3_TIMING: cycle stamps per phase, reported INSTEAD of result[0 .. 7] (tools/dev/probe_wave3.py; never benchmarked)
#ifdef LOCAMD_WAVE3_TIMING
#define W3_T0() unsigned long long w3_tc = __builtin_readcyclecounter(), w3_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long w3_start = w3_tc
#define W3_T(k) do { const unsigned long long n_ = __builtin_readcyclecounter(); w3_ph[k] += n_ - w3_tc; w3_tc = n_; } while (0)
#else
#define W3_T0() do {} while (0)
#define W3_T(k) do {} while (0)
#endif

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double w3_dpp(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, CTRL == 0x138 || CTRL == 0x130);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, CTRL == 0x138 || CTRL == 0x130);
    return __hiloint2double(hi, lo);
}
// value of lane l (wave-uniform l) in every lane — through SGPRs
__device__ __forceinline__ double w3_bcast(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
// the value of lane - 1 / lane + 1 (wave_shr:1 / wave_shl:1 cross the 16-lane rows on gfx9; no neighbour: 0)
__device__ __forceinline__ double w3_from_prev(double v) { return w3_dpp<0x138, 0xF>(v); }
__device__ __forceinline__ double w3_from_next(double v) { return w3_dpp<0x130, 0xF>(v); }
__device__ __forceinline__ double w3_sum(double v) {   // DPP row shifts + row broadcasts: one fixed order, every lane gets the same bits
    v += w3_dpp<0x111, 0xF>(v);
    v += w3_dpp<0x112, 0xF>(v);
    v += w3_dpp<0x114, 0xF>(v);
    v += w3_dpp<0x118, 0xF>(v);
    v += w3_dpp<0x142, 0xA>(v);
    v += w3_dpp<0x143, 0xC>(v);
    return w3_bcast(v, 63);
}
__device__ __forceinline__ double w3_max(double v) {   // non-negative inputs
    v = fmax(v, w3_dpp<0x111, 0xF>(v));
    v = fmax(v, w3_dpp<0x112, 0xF>(v));
    v = fmax(v, w3_dpp<0x114, 0xF>(v));
    v = fmax(v, w3_dpp<0x118, 0xF>(v));
    v = fmax(v, w3_dpp<0x142, 0xA>(v));
    v = fmax(v, w3_dpp<0x143, 0xC>(v));
    return w3_bcast(v, 63);
}
// one wave per workgroup: the LDS operations of a wave execute in order; the fences keep the compiler from moving them
__device__ __forceinline__ void w3_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ double w3_pivot_rsqrt(double d) {   // window_kernel.hip: pivot_rsqrt
    const double y = __builtin_amdgcn_rsq(d);
    const double t = d * y;
    const double e = __builtin_fma(-t, y, 1.0);
    const double pq = __builtin_fma(0.375, e, 0.5);
    const double ye = y * e;
    return __builtin_fma(ye, pq, y);
}

#pragma clang fp contract(off)
// ||d|| the way a plain CPU build of computeError evaluates it (numeric_jacobian.h: range_error_plain with a zero lever arm)
__device__ __forceinline__ double w3_sq_plain(double dx, double dy, double dz) { return dx * dx + dy * dy + dz * dz; }
// g2o's central difference of e = meas - ||p0 - p1|| along axis D of endpoint `which`'s translation (R = I, zero lever arm:
// X * fromVectorMQT(+-delta e_D) = (I, t +- delta e_D); chain3_kernel.hip: range_jac_numeric3)
// NEAR: the perturbed norms from the central one n0 (device_math.h: sqrt_ieee_near_c — the same correctly rounded numbers)
template <int D, bool NEAR>
__device__ __forceinline__ double w3_jac_numeric(const double* p0, const double* p1, int which, double meas, double n0, double h0) {
    constexpr double delta = 1e-9;
    constexpr double scalar = 1.0 / (2 * delta);
    double a[3] = {p0[0], p0[1], p0[2]}, b[3] = {p1[0], p1[1], p1[2]}, am[3] = {p0[0], p0[1], p0[2]}, bm[3] = {p1[0], p1[1], p1[2]};
    if (which == 0) { a[D] = delta + p0[D]; am[D] = -delta + p0[D]; }
    else { b[D] = delta + p1[D]; bm[D] = -delta + p1[D]; }
    const double xp = w3_sq_plain(a[0] - b[0], a[1] - b[1], a[2] - b[2]), xm = w3_sq_plain(am[0] - bm[0], am[1] - bm[1], am[2] - bm[2]);
    const double ep = meas - (NEAR ? sqrt_ieee_near_c(xp, n0, h0, (h0 * h0) * (h0 + h0)) : sqrt_ieee_unscaled(xp));
    const double em = meas - (NEAR ? sqrt_ieee_near_c(xm, n0, h0, (h0 * h0) * (h0 + h0)) : sqrt_ieee_unscaled(xm));
    double bak = ep;
    bak -= em;
    return scalar * bak;
}
#pragma clang fp contract(fast)

// LDS of one window (doubles first, then ints)
struct W3Lds {
    double* T;      // [5][nv_max][3] translations: the state and up to four trial states
    double* rec;    // [2 nr_max][8]  the last linearisation, one record per (edge, moving endpoint), grouped by pose: w, -w e, the
                    //                pose's own J (3), the other endpoint's J (3) when that is the previous pose (else 0)
    double* ev;     // [nr_max][2]    measurement, information
    double* fix;    // [nr_max][3]    the fixed endpoint of an anchor edge
    double* pv;     // [np_max][6]    priors: Z^-1 t (3), information diagonal (3)
    int* eidx;      // [nr_max][2]
    int* epos;      // [nr_max][2]    where the edge's two records go (endpoint 1 of an anchor edge: -1)
    int* pidx;      // [np_max]
};
__device__ __forceinline__ W3Lds w3_carve(const WindowCaps& c) {
    W3Lds l;
    double* p = w3lds;
    l.T = p; p += 5 * c.nv_max * 3;
    l.rec = p; p += (size_t)c.nr_max * 16;
    l.ev = p; p += (size_t)c.nr_max * 2;
    l.fix = p; p += (size_t)c.nr_max * 3;
    l.pv = p; p += (size_t)c.np_max * 6;
    int* q = reinterpret_cast<int*>(p);
    l.eidx = q; q += (size_t)c.nr_max * 2;
    l.epos = q; q += (size_t)c.nr_max * 2;
    l.pidx = q;
    return l;
}

// one range edge as its lane holds it (the first 64 edges of a window stay in registers for the whole solve)
struct W3Edge {
    int v0, v1, s0, s1;
    double meas, info, fx, fy, fz;
};
__device__ __forceinline__ W3Edge w3_load_edge(const W3Lds& l, int e) {
    W3Edge E;
    E.v0 = l.eidx[2 * e]; E.v1 = l.eidx[2 * e + 1];
    E.s0 = l.epos[2 * e]; E.s1 = l.epos[2 * e + 1];
    E.meas = l.ev[2 * e]; E.info = l.ev[2 * e + 1];
    E.fx = l.fix[3 * e]; E.fy = l.fix[3 * e + 1]; E.fz = l.fix[3 * e + 2];
    return E;
}

// one edge at the translations T: its robust / plain chi2; FULL: its linearisation records as well
template <bool FULL, int JAC>
__device__ __forceinline__ void w3_edge(const W3Lds& l, const double* T, const W3Edge& E, double& rsum, double& csum) {
    const int v0 = E.v0, v1 = E.v1;
    const double meas = E.meas, info = E.info;
    const double* s0 = T + v0 * 3;
    const double* s1 = T + (v1 >= 0 ? v1 : 0) * 3;
    const double p0[3] = {s0[0], s0[1], s0[2]};
    const double m1[3] = {s1[0], s1[1], s1[2]};
    const double p1[3] = {v1 >= 0 ? m1[0] : E.fx, v1 >= 0 ? m1[1] : E.fy, v1 >= 0 ? m1[2] : E.fz};
    double u[3] = {p0[0] - p1[0], p0[1] - p1[1], p0[2] - p1[2]};
    double err, inv = 0.0, x0 = 0.0, n0 = 0.0, h0 = 0.0;
    if (JAC == 0) {
        const double x = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
        double n;
        sqrt_and_rsqrt(x, n, inv);
        if (!(x > 0.0)) { n = 0.0; inv = 0.0; }   // coincident endpoints: J = 0, what the central difference gives
        err = meas - n;
    } else {
        x0 = w3_sq_plain(u[0], u[1], u[2]);
        n0 = sqrt_ieee_unscaled_h(x0, h0);
        err = meas - n0;
    }
    const double chi = err * (info * err);
    const double aux = 1.0 + chi;
    rsum += fast_log_ge1(aux);
    csum += chi;
    if (FULL) {
        double J0[3], J1[3];
        if (JAC == 0) {
            u[0] *= inv; u[1] *= inv; u[2] *= inv;
            J0[0] = -u[0]; J0[1] = -u[1]; J0[2] = -u[2];
            J1[0] = u[0]; J1[1] = u[1]; J1[2] = u[2];
        } else {
            if (x0 >= 1e-5 && x0 < 1e300) {   // endpoints more than ~3 mm apart
                J0[0] = w3_jac_numeric<0, true>(p0, p1, 0, meas, n0, h0);
                J0[1] = w3_jac_numeric<1, true>(p0, p1, 0, meas, n0, h0);
                J0[2] = w3_jac_numeric<2, true>(p0, p1, 0, meas, n0, h0);
                J1[0] = w3_jac_numeric<0, true>(p0, p1, 1, meas, n0, h0);
                J1[1] = w3_jac_numeric<1, true>(p0, p1, 1, meas, n0, h0);
                J1[2] = w3_jac_numeric<2, true>(p0, p1, 1, meas, n0, h0);
            } else {
                J0[0] = w3_jac_numeric<0, false>(p0, p1, 0, meas, n0, h0);
                J0[1] = w3_jac_numeric<1, false>(p0, p1, 0, meas, n0, h0);
                J0[2] = w3_jac_numeric<2, false>(p0, p1, 0, meas, n0, h0);
                J1[0] = w3_jac_numeric<0, false>(p0, p1, 1, meas, n0, h0);
                J1[1] = w3_jac_numeric<1, false>(p0, p1, 1, meas, n0, h0);
                J1[2] = w3_jac_numeric<2, false>(p0, p1, 1, meas, n0, h0);
            }
        }
        const double wr = info * fast_rcp(aux), wre = -wr * err;
        const bool c0 = v1 >= 0 && v1 == v0 - 1, c1 = v0 == v1 - 1;   // the endpoint that is the later pose of a consecutive pair keeps the coupling block
        double* r = l.rec + (size_t)E.s0 * 8;
        r[0] = wr; r[1] = wre;
#pragma unroll
        for (int k = 0; k < 3; ++k) { r[2 + k] = J0[k]; r[5 + k] = c0 ? J1[k] : 0.0; }
        if (v1 >= 0) {
            double* r1 = l.rec + (size_t)E.s1 * 8;
            r1[0] = wr; r1[1] = wre;
#pragma unroll
            for (int k = 0; k < 3; ++k) { r1[2 + k] = J1[k]; r1[5 + k] = c1 ? J0[k] : 0.0; }
        }
    }
}

// every edge and prior of the window at the translations of buffer `buf`: the robust and plain chi2 sums; FULL: the edges'
// linearisation records as well
template <bool FULL, int JAC>
__device__ __forceinline__ void w3_edges(const W3Lds& l, const W3Edge& E0, int nvm, int nr, int np, int buf, int lane, double& robust_chi, double& plain_chi) {
    double rsum = 0.0, csum = 0.0;
    const double* T = l.T + (size_t)buf * nvm * 3;
    if (lane < nr) w3_edge<FULL, JAC>(l, T, E0, rsum, csum);
    for (int e = lane + 64; e < nr; e += 64) w3_edge<FULL, JAC>(l, T, w3_load_edge(l, e), rsum, csum);
    // unary priors: e = t + Z^-1.t (identity rotations), diagonal information on the translation, no robust kernel
    for (int q = lane; q < np; q += 64) {
        const double* s = T + l.pidx[q] * 3;
        const double* v = l.pv + (size_t)q * 6;
        double chi = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) { const double er = s[k] + v[k]; chi += er * (v[3 + k] * er); }
        rsum += chi;
        csum += chi;
    }
    robust_chi = w3_sum(rsum);
    plain_chi = w3_sum(csum);
}

// TWO trial states scored in one pass (windows of <= 32 edges and <= 32 priors): state A by lanes 0 .. 31, state B by lanes 32 .. 63, every
// lane with the edge of its position in its half.  The sums are the DPP tree of w3_sum read where a half has been summed — lanes 31 and 63
// after the row_bcast:15 step — which is bit for bit what the whole-wave sum of ONE state gives (its other rows add zeros).
template <int JAC>
__device__ __forceinline__ void w3_edges_dual(const W3Lds& l, const W3Edge& E0, int nvm, int nr, int np, int bufA, int bufB, int lane,
                                              double& chiA, double& plainA, double& chiB, double& plainB) {
    double rsum = 0.0, csum = 0.0;
    const int el = lane & 31;
    const double* T = l.T + (size_t)(lane < 32 ? bufA : bufB) * nvm * 3;
    if (el < nr) w3_edge<false, JAC>(l, T, E0, rsum, csum);
    if (el < np) {
        const double* s = T + l.pidx[el] * 3;
        const double* v = l.pv + (size_t)el * 6;
        double chi = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) { const double er = s[k] + v[k]; chi += er * (v[3 + k] * er); }
        rsum += chi;
        csum += chi;
    }
    rsum += w3_dpp<0x111, 0xF>(rsum); csum += w3_dpp<0x111, 0xF>(csum);
    rsum += w3_dpp<0x112, 0xF>(rsum); csum += w3_dpp<0x112, 0xF>(csum);
    rsum += w3_dpp<0x114, 0xF>(rsum); csum += w3_dpp<0x114, 0xF>(csum);
    rsum += w3_dpp<0x118, 0xF>(rsum); csum += w3_dpp<0x118, 0xF>(csum);
    rsum += w3_dpp<0x142, 0xA>(rsum); csum += w3_dpp<0x142, 0xA>(csum);
    chiA = w3_bcast(rsum, 31); plainA = w3_bcast(csum, 31);
    chiB = w3_bcast(rsum, 63); plainB = w3_bcast(csum, 63);
}

template <int JAC>
__global__ void __launch_bounds__(64, 3) wave3_lm_kernel(const WindowArgs a) {
    const int lane = threadIdx.x;
    const long long inst = blockIdx.x;
    const WindowCaps& cp = a.caps;
    const W3Lds l = w3_carve(cp);
    const int nvm = cp.nv_max;
    const double* gin = a.poses_in + (size_t)inst * nvm * 12;
    double* gout = a.poses + (size_t)inst * nvm * 12;
    const int32_t* ridx = a.r_idx + (size_t)inst * cp.nr_max * 2;
    const double* rval = a.r_val + (size_t)inst * cp.nr_max * 5;
    // the first 64 poses and edges are requested before the counts have arrived (inside the instance's slices whatever the counts
    // say): the node's window comes straight from page-locked host memory, a round trip of microseconds per dependent level
    double pin[12];
    int pv0 = 0, pv1 = 0;
    double pmeas = 0.0, pinfo = 0.0;
#pragma unroll
    for (int k = 9; k < 12; ++k) pin[k] = 0.0;
    if (lane < nvm) {
#pragma unroll
        for (int k = 9; k < 12; ++k) pin[k] = gin[lane * 12 + k];
    }
    if (lane < cp.nr_max) { pv0 = ridx[2 * lane]; pv1 = ridx[2 * lane + 1]; pmeas = rval[5 * lane]; pinfo = rval[5 * lane + 1]; }
    const int nv = a.counts[inst * 4 + 0], nr = a.counts[inst * 4 + 1], np = a.counts[inst * 4 + 2];
    // speculation: G groups of W lanes solve the SAME normal equations with the lambdas of this and the next G - 1 trials (what LM
    // would try next if it rejects); a group keeps one idle lane after its last pose (the hand-over between lanes reads zeros there)
    const int W = nv <= 15 ? 16 : (nv <= 31 ? 32 : 64);
    const int G = 64 / W;
    const int grp = lane / W, pp = lane & (W - 1);
    const bool pose = pp < nv;
    const unsigned long long group_mask = W == 64 ? ~0ull : ((1ull << W) - 1ull);
    W3_T0();
    // ---- set-up: translations, edges (with their fixed endpoints), priors into LDS; the poses' incidence lists ---------------
    if (lane < nv) {
#pragma unroll
        for (int k = 0; k < 3; ++k) l.T[lane * 3 + k] = pin[9 + k];
    }
    {
        for (int e = lane; e < nr; e += 64) {
            const bool first = e == lane;
            const int v0 = first ? pv0 : ridx[2 * e], v1 = first ? pv1 : ridx[2 * e + 1];
            l.eidx[2 * e] = v0; l.eidx[2 * e + 1] = v1;
            l.ev[2 * e] = first ? pmeas : rval[5 * e]; l.ev[2 * e + 1] = first ? pinfo : rval[5 * e + 1];
            l.epos[2 * e + 1] = -1;
            const double* an = a.anchors + (size_t)(v1 < 0 ? -1 - v1 : 0) * 3;
            l.fix[3 * e] = an[0]; l.fix[3 * e + 1] = an[1]; l.fix[3 * e + 2] = an[2];
        }
        const int32_t* pidx = a.p_idx + (size_t)inst * cp.np_max;
        const double* pval = a.p_val + (size_t)inst * cp.np_max * 18;
        for (int q = lane; q < np; q += 64) {
            l.pidx[q] = pidx[q];
#pragma unroll
            for (int k = 0; k < 3; ++k) { l.pv[6 * q + k] = pval[18 * q + 9 + k]; l.pv[6 * q + 3 + k] = pval[18 * q + 12 + k]; }
        }
    }
    w3_sync();
    int deg = 0, nbin = 0;   // (lanes 0 .. nv-1: pose = lane)
    for (int e = 0; e < nr; ++e) {
        const int v0 = l.eidx[2 * e], v1 = l.eidx[2 * e + 1];
        deg += (v0 == lane) + (v1 == lane);
        nbin += (v1 >= 0 && (v0 > v1 ? v0 : v1) == lane);
    }
    int lst = 0;   // first entry of this pose's list: exclusive prefix sum of deg over the lanes
    int kc = -1;   // the record of the (last) edge to the previous pose
    {
        int incl = deg;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        lst = incl - deg;
        int k = lst;
        for (int e = 0; e < nr; ++e) {
            const int v0 = l.eidx[2 * e], v1 = l.eidx[2 * e + 1];
            if (v0 == lane) { if (v1 >= 0 && v1 == lane - 1) kc = k; l.epos[2 * e] = k++; }
            if (v1 == lane) { if (v0 == lane - 1) kc = k; l.epos[2 * e + 1] = k++; }
        }
    }
    const int shared_edges = (int)w3_sum(lane < nv && nbin >= 2 ? (double)nbin : 0.0);
    // no pair of consecutive poses with more than one edge (the reference's own window: one smoothness edge per pair): every
    // coupling block is a rank-1 product (w J_p) J_{p-1}^T, and the sweeps below hand two numbers from pose to pose instead of nine
    const bool rank1 = __ballot(lane < nv && nbin >= 2) == 0ull;
    deg = __shfl(deg, pp, 64);   // every group's lane of pose pp
    lst = __shfl(lst, pp, 64);
    kc = __shfl(kc, pp, 64);
    w3_sync();
    W3Edge E0;
    E0.v0 = 0; E0.v1 = -1; E0.s0 = 0; E0.s1 = -1; E0.meas = 0.0; E0.info = 0.0; E0.fx = 0.0; E0.fy = 0.0; E0.fz = 0.0;
    // (speculative trials are scored two at a time when a window's edges fit half a wave: the upper half keeps the same edges)
    const bool dual = G >= 2 && nr <= 32 && np <= 32;
    if ((dual ? (lane & 31) : lane) < nr) E0 = w3_load_edge(l, dual ? (lane & 31) : lane);
    W3_T(0);

    // ---- Levenberg-Marquardt (g2o: OptimizationAlgorithmLevenberg::solve), wave-uniform control flow ---------------
    constexpr double tau = 1e-5, good_lo = 1.0 / 3.0, good_hi = 2.0 / 3.0;
    constexpr int max_trials = 10;
    constexpr int NSLOT = 5;   // translation buffers in LDS: the state + up to four trial states
    double lambda = 0.0, ni = 2.0, cur_chi = 0.0, last_plain = 0.0;
    int it = 0, q = 0, trials = 0, terminated = 0, cur = 0, jlast = 0;
    bool need_lin = true;
    bool done = nv <= 0 || nr + np <= 0 || a.iterations <= 0;
    double D[6], b[3], O[9], X[3] = {0.0, 0.0, 0.0};   // this pose's H_pp, b_p, H_p,p-1 (entry (r, c) at 3 c + r); x_p of this group's last solve
    double u[3] = {0.0, 0.0, 0.0}, UU[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, vnext[3] = {0.0, 0.0, 0.0};   // rank1: H_p,p-1 = u v^T; u u^T; the v of pose p + 1's block (this pose's own J)
#pragma unroll
    for (int k = 0; k < 6; ++k) D[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) b[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) O[k] = 0.0;
    while (!done) {
        if (need_lin) {
            double plain;
            w3_edges<true, JAC>(l, E0, nvm, nr, np, cur, lane, cur_chi, plain);
            last_plain = plain;
            w3_sync();
            W3_T(1);
            // pose pp: its edges' blocks in creation order (every group's lane of the pose holds the same numbers)
#pragma unroll
            for (int k = 0; k < 6; ++k) D[k] = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) b[k] = 0.0;
#pragma unroll
            for (int k = 0; k < 9; ++k) O[k] = 0.0;
            double uu_[3] = {0.0, 0.0, 0.0}, vv_[3] = {0.0, 0.0, 0.0};
            if (pose) {
                const int k1 = lst + deg;
                auto fold = [&](int k, const double* r) __attribute__((always_inline)) {
                    const double wr = r[0], wre = r[1];
                    const double Jm[3] = {r[2], r[3], r[4]}, Jo[3] = {r[5], r[6], r[7]};
                    const double wj[3] = {wr * Jm[0], wr * Jm[1], wr * Jm[2]};
#pragma unroll
                    for (int rr = 0; rr < 3; ++rr) {
#pragma unroll
                        for (int cc = 0; cc <= rr; ++cc) D[rr * (rr + 1) / 2 + cc] = __builtin_fma(wj[rr], Jm[cc], D[rr * (rr + 1) / 2 + cc]);
                        b[rr] = __builtin_fma(Jm[rr], wre, b[rr]);
#pragma unroll
                        for (int cc = 0; cc < 3; ++cc) O[3 * cc + rr] = __builtin_fma(wj[rr], Jo[cc], O[3 * cc + rr]);
                    }
                    if (k == kc) {
#pragma unroll
                        for (int rr = 0; rr < 3; ++rr) { uu_[rr] = wj[rr]; vv_[rr] = Jo[rr]; }
                    }
                };
                // the first three records together (their LDS reads in flight at once: a pose of the reference's window has three), then the rest
                double r3[3][8];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int c8 = 0; c8 < 8; ++c8) r3[i][c8] = lst + i < k1 ? l.rec[(size_t)(lst + i) * 8 + c8] : 0.0;
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    if (lst + i < k1) fold(lst + i, r3[i]);
                for (int k = lst + 3; k < k1; ++k) fold(k, l.rec + (size_t)k * 8);
                for (int pq = 0; pq < np; ++pq) {
                    if (l.pidx[pq] != pp) continue;
                    const double* v = l.pv + (size_t)pq * 6;
                    const double* s = l.T + ((size_t)cur * nvm + pp) * 3;
#pragma unroll
                    for (int k = 0; k < 3; ++k) { const double er = s[k] + v[k]; D[k * (k + 1) / 2 + k] += v[3 + k]; b[k] += -v[3 + k] * er; }
                }
            }
            if (rank1) {
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) {
                    u[rr] = uu_[rr];
                    vnext[rr] = w3_from_next(vv_[rr]);
#pragma unroll
                    for (int cc = 0; cc <= rr; ++cc) UU[rr * (rr + 1) / 2 + cc] = uu_[rr] * uu_[cc];
                }
            }
            if (it == 0) {
                const double md = fmax(fmax(fabs(D[0]), fabs(D[2])), fabs(D[5]));
                lambda = tau * w3_max(pose ? md : 0.0);
                ni = 2.0;
            }
            q = 0;
            need_lin = false;
            W3_T(2);
        }
        // ---- one round: (H + lambda_g I) x = b for the lambdas of the next G trials --------------------------------------------------
        double lamv[4], niv[4];
        lamv[0] = lambda; niv[0] = ni;
#pragma unroll
        for (int g = 1; g < 4; ++g) { lamv[g] = lamv[g - 1] * niv[g - 1]; niv[g] = 2.0 * niv[g - 1]; }   // (a rejected trial: lambda *= ni, ni *= 2)
        const double mylam = grp == 0 ? lamv[0] : (grp == 1 ? lamv[1] : (grp == 2 ? lamv[2] : lamv[3]));
        // The solve.  rank1 (every coupling H_p,p-1 = u_p v_p^T): the Schur complement of pose p is S_p = A_p - alpha_p u_p u_p^T with
        // A_p = H_pp + lambda I and the SCALAR alpha_p = v_p^T S_{p-1}^-1 v_p, its right-hand side b_p - beta_p u_p with
        // beta_p = v_p^T S_{p-1}^-1 (b_{p-1} - beta_{p-1} u_{p-1}).  Sherman-Morrison gives S_p^-1 from A_p^-1:
        //   S^-1 = A^-1 + k (A^-1 u)(A^-1 u)^T,  k = alpha / den,  den = 1 - alpha u^T A^-1 u   (S positive definite <=> A is and den > 0)
        // so everything that needs a 3x3 factorisation — A_p = G G^T, A^-1 u, A^-1 v, A^-1 b and their five dot products — is done by ALL
        // poses AT ONCE, and the sequential part of the block-tridiagonal elimination shrinks to a recurrence on two scalars per pose
        // (forward: alpha, beta; backward: gamma = u_{p+1} . x_{p+1}).  Each repetition every lane takes its neighbour's scalars through
        // DPP and redoes its own step: after repetition r the poses 0 .. r hold final values (a lane whose inputs are final recomputes
        // the same numbers); all G groups ride in the same instructions.
        double Xn[3] = {0.0, 0.0, 0.0};
        unsigned long long bad = 0ull;
        if (rank1) {
            double pA[3] = {0.0, 0.0, 0.0}, qA[3] = {0.0, 0.0, 0.0}, rA[3] = {0.0, 0.0, 0.0};   // A^-1 u, A^-1 v_next, A^-1 b
            double uu = 0.0, uv = 0.0, vvq = 0.0, ub = 0.0, vb = 0.0;
            bool piv_ok = true;
            if (pose) {
                double A[3][3], ig[3];
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) {
#pragma unroll
                    for (int c = 0; c <= rr; ++c) A[rr][c] = D[rr * (rr + 1) / 2 + c];
                    A[rr][rr] += mylam;
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const double g = w3_pivot_rsqrt(A[j][j]);
                    ig[j] = g;
#pragma unroll
                    for (int i2 = j + 1; i2 < 3; ++i2) A[i2][j] *= g;
#pragma unroll
                    for (int i2 = j + 1; i2 < 3; ++i2)
#pragma unroll
                        for (int cc = j + 1; cc <= i2; ++cc) A[i2][cc] = __builtin_fma(-A[i2][j], A[cc][j], A[i2][cc]);
                }
                piv_ok = (ig[0] + ig[1]) + ig[2] < DBL_MAX;
#pragma unroll
                for (int k = 0; k < 3; ++k) { pA[k] = u[k]; qA[k] = vnext[k]; rA[k] = b[k]; }
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    pA[cc] *= ig[cc]; qA[cc] *= ig[cc]; rA[cc] *= ig[cc];
#pragma unroll
                    for (int c2 = cc + 1; c2 < 3; ++c2) {
                        pA[c2] = __builtin_fma(-pA[cc], A[c2][cc], pA[c2]);
                        qA[c2] = __builtin_fma(-qA[cc], A[c2][cc], qA[c2]);
                        rA[c2] = __builtin_fma(-rA[cc], A[c2][cc], rA[c2]);
                    }
                }
#pragma unroll
                for (int cc = 2; cc >= 0; --cc) {
                    double a0 = pA[cc], a1 = qA[cc], a2 = rA[cc];
#pragma unroll
                    for (int c2 = cc + 1; c2 < 3; ++c2) {
                        a0 = __builtin_fma(-A[c2][cc], pA[c2], a0);
                        a1 = __builtin_fma(-A[c2][cc], qA[c2], a1);
                        a2 = __builtin_fma(-A[c2][cc], rA[c2], a2);
                    }
                    pA[cc] = a0 * ig[cc]; qA[cc] = a1 * ig[cc]; rA[cc] = a2 * ig[cc];
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    uu = __builtin_fma(u[k], pA[k], uu); uv = __builtin_fma(u[k], qA[k], uv); vvq = __builtin_fma(vnext[k], qA[k], vvq);
                    ub = __builtin_fma(u[k], rA[k], ub); vb = __builtin_fma(vnext[k], rA[k], vb);
                }
            }
            W3_T(3);
            double bin = 0.0, kk = 0.0, rden = 1.0, den = 1.0;
            {
                double al = 0.0, be = 0.0;
                for (int r = 0; r < nv; ++r) {
                    const double pal = w3_from_prev(al), pbe = w3_from_prev(be);
                    if (pose) {
                        bin = pbe;
                        den = __builtin_fma(-pal, uu, 1.0);
                        rden = fast_rcp(den);
                        kk = pal * rden;
                        const double kuv = kk * uv;
                        al = __builtin_fma(kuv, uv, vvq);
                        be = __builtin_fma(kuv, __builtin_fma(-pbe, uu, ub), __builtin_fma(-pbe, uv, vb));
                    }
                }
            }
            // a group in which some A_p or some Schur complement is not positive definite (or not finite) has failed
            bad = __ballot(pose && !(piv_ok && den > 0.0 && den < DBL_MAX));
            {
                double ga = 0.0, gin = 0.0, cfin = 0.0;
                for (int r = 0; r < nv; ++r) {
                    const double pga = w3_from_next(ga);
                    if (pose) {
                        gin = pga;
                        cfin = __builtin_fma(-pga, uv, __builtin_fma(-bin, uu, ub));   // u^T A^-1 (b - beta u - gamma v)
                        ga = cfin * rden;
                    }
                }
                if (pose) {
                    const double cb = __builtin_fma(-kk, cfin, bin);   // x = A^-1 b - (beta - k c) A^-1 u - gamma A^-1 v
#pragma unroll
                    for (int k = 0; k < 3; ++k) Xn[k] = __builtin_fma(-gin, qA[k], __builtin_fma(-cb, pA[k], rA[k]));
                }
            }
        } else {
            // windows with several edges on a pair of consecutive poses: 3x3 coupling blocks, the factor (G, y) of pose p - 1 handed to
            // pose p through DPP the same way
            double g10 = 0.0, g20 = 0.0, g21 = 0.0, ig[3] = {0.0, 0.0, 0.0}, y[3] = {0.0, 0.0, 0.0};   // this pose's factor (strict lower part, inverse pivots), y
            for (int r = 0; r < nv; ++r) {
                const double pg10 = w3_from_prev(g10), pg20 = w3_from_prev(g20), pg21 = w3_from_prev(g21);
                const double pig[3] = {w3_from_prev(ig[0]), w3_from_prev(ig[1]), w3_from_prev(ig[2])};
                const double py[3] = {w3_from_prev(y[0]), w3_from_prev(y[1]), w3_from_prev(y[2])};
                if (pose) {
                    double A[3][3], rhs[3];
    #pragma unroll
                    for (int rr = 0; rr < 3; ++rr) {
    #pragma unroll
                        for (int c = 0; c <= rr; ++c) A[rr][c] = D[rr * (rr + 1) / 2 + c];
                        A[rr][rr] += mylam;
                        rhs[rr] = b[rr];
                    }
                    // row by row: w = row r of W = H_p,p-1 G_{p-1}^-T; S -= w w^T; rhs_r -= w . y_{p-1}   (pose 0: O = 0 and a zero neighbour)
                    double Wm[9];
    #pragma unroll
                    for (int rr = 0; rr < 3; ++rr) {
                        double w0 = O[rr], w1 = O[3 + rr], w2 = O[6 + rr];
                        w0 *= pig[0];
                        w1 = __builtin_fma(-w0, pg10, w1);
                        w2 = __builtin_fma(-w0, pg20, w2);
                        w1 *= pig[1];
                        w2 = __builtin_fma(-w1, pg21, w2);
                        w2 *= pig[2];
                        Wm[rr] = w0; Wm[3 + rr] = w1; Wm[6 + rr] = w2;
                        double acc = rhs[rr];
                        acc = __builtin_fma(-w0, py[0], acc);
                        acc = __builtin_fma(-w1, py[1], acc);
                        acc = __builtin_fma(-w2, py[2], acc);
                        rhs[rr] = acc;
                    }
    #pragma unroll
                    for (int rr = 0; rr < 3; ++rr)
    #pragma unroll
                        for (int c2 = 0; c2 <= rr; ++c2) {
                            double s2 = A[rr][c2];
    #pragma unroll
                            for (int k = 0; k < 3; ++k) s2 = __builtin_fma(-Wm[3 * k + rr], Wm[3 * k + c2], s2);
                            A[rr][c2] = s2;
                        }
    #pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const double g = w3_pivot_rsqrt(A[j][j]);
                        ig[j] = g;
    #pragma unroll
                        for (int i2 = j + 1; i2 < 3; ++i2) A[i2][j] *= g;
    #pragma unroll
                        for (int i2 = j + 1; i2 < 3; ++i2)
    #pragma unroll
                            for (int cc = j + 1; cc <= i2; ++cc) A[i2][cc] = __builtin_fma(-A[i2][j], A[cc][j], A[i2][cc]);
                    }
    #pragma unroll
                    for (int cc = 0; cc < 3; ++cc) {
                        rhs[cc] *= ig[cc];
    #pragma unroll
                        for (int c2 = cc + 1; c2 < 3; ++c2) rhs[c2] = __builtin_fma(-rhs[cc], A[c2][cc], rhs[c2]);
                    }
                    g10 = A[1][0]; g20 = A[2][0]; g21 = A[2][1];
    #pragma unroll
                    for (int rr = 0; rr < 3; ++rr) y[rr] = rhs[rr];
                }
            }
                    // a group whose factorisation met a pivot that is not positive and finite has failed
            bad = __ballot(pose && !((ig[0] + ig[1]) + ig[2] < DBL_MAX));
            W3_T(3);
            // back-substitution the same way, from the right neighbour: v = H_{p+1,p}^T x_{p+1}
            double vo[3] = {0.0, 0.0, 0.0};
            for (int r = 0; r < nv; ++r) {
                const double v[3] = {w3_from_next(vo[0]), w3_from_next(vo[1]), w3_from_next(vo[2])};
                if (pose) {
                    double t[3] = {y[0], y[1], y[2]};
                    // z = G_p^-1 v (forward substitution), t -= z
                    double z0 = v[0] * ig[0];
                    double z1 = __builtin_fma(-z0, g10, v[1]);
                    double z2 = __builtin_fma(-z0, g20, v[2]);
                    t[0] -= z0;
                    z1 *= ig[1];
                    z2 = __builtin_fma(-z1, g21, z2);
                    t[1] -= z1;
                    z2 *= ig[2];
                    t[2] -= z2;
                    Xn[2] = t[2] * ig[2];
                    t[0] = __builtin_fma(-g20, Xn[2], t[0]);
                    t[1] = __builtin_fma(-g21, Xn[2], t[1]);
                    Xn[1] = t[1] * ig[1];
                    t[0] = __builtin_fma(-g10, Xn[1], t[0]);
                    Xn[0] = t[0] * ig[0];
    #pragma unroll
                    for (int cc = 0; cc < 3; ++cc) {
                        double acc = 0.0;
    #pragma unroll
                        for (int rr = 0; rr < 3; ++rr) acc = __builtin_fma(O[3 * cc + rr], Xn[rr], acc);
                        vo[cc] = acc;
                    }
                }
            }
                }
#ifdef LOCAMD_WAVE3_TIMING
        ++w3_ph[7];
#endif
        // x of a failed factorisation: g2o leaves its x alone, and LM applies that stale x all the same — the x of the
        // trial before, i.e. of the group before (group 0: of the trial consumed last)
        if (bad) {
            double st[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) st[k] = __shfl(X[k], jlast * W + pp, 64);
            for (int g = 0; g < G; ++g) {
                if (grp == g && ((bad >> (g * W)) & group_mask)) { Xn[0] = st[0]; Xn[1] = st[1]; Xn[2] = st[2]; }
#pragma unroll
                for (int k = 0; k < 3; ++k) st[k] = __shfl(Xn[k], g * W + pp, 64);
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) X[k] = Xn[k];
        W3_T(4);
        // the trial states t + x (VertexSE3::oplus with R = I) and g2o's computeScale sums, one per group
        double scv[4];
        {
            double sc = 0.0;
            if (pose) {
                const int slot = cur + 1 + grp - (cur + 1 + grp >= NSLOT ? NSLOT : 0);
                const double* s = l.T + ((size_t)cur * nvm + pp) * 3;
                double* d = l.T + ((size_t)slot * nvm + pp) * 3;
#pragma unroll
                for (int k = 0; k < 3; ++k) { sc += X[k] * (mylam * X[k] + b[k]); d[k] = s[k] + X[k]; }
            }
            // (the DPP tree of w3_sum, read where a group's lanes have been summed: the same bits as the whole-wave sum of one group)
            sc += w3_dpp<0x111, 0xF>(sc);
            sc += w3_dpp<0x112, 0xF>(sc);
            sc += w3_dpp<0x114, 0xF>(sc);
            sc += w3_dpp<0x118, 0xF>(sc);
            const double s16 = sc;
            sc += w3_dpp<0x142, 0xA>(sc);
            const double s32 = sc;
            sc += w3_dpp<0x143, 0xC>(sc);
#pragma unroll
            for (int g = 0; g < 4; ++g) scv[g] = W == 16 ? w3_bcast(s16, 16 * g + 15) : (W == 32 ? w3_bcast(s32, 32 * (g & 1) + 31) : w3_bcast(sc, 63));
        }
        w3_sync();
        // ---- consume the trials in LM's order until one is accepted (or the iteration ends) -------------------------------------------
        bool iteration_over = false;
        double rho = 0.0;
        double tchi[4] = {0.0, 0.0, 0.0, 0.0}, tplain[4] = {0.0, 0.0, 0.0, 0.0};
        bool scored[4] = {false, false, false, false};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < G && !iteration_over) {
                const int slot = cur + 1 + g - (cur + 1 + g >= NSLOT ? NSLOT : 0);
                if (!scored[g]) {
                    if (g < 3 && dual && g + 1 < G) {   // this trial and the next one in one pass
                        const int slot2 = cur + 2 + g - (cur + 2 + g >= NSLOT ? NSLOT : 0);
                        w3_edges_dual<JAC>(l, E0, nvm, nr, np, slot, slot2, lane, tchi[g], tplain[g], tchi[g < 3 ? g + 1 : 3], tplain[g < 3 ? g + 1 : 3]);
                        scored[g < 3 ? g + 1 : 3] = true;
                    } else {
                        w3_edges<false, JAC>(l, E0, nvm, nr, np, slot, lane, tchi[g], tplain[g]);
                    }
                }
                double temp_chi = tchi[g];
                const double plain2 = tplain[g];
                last_plain = plain2;
                ++trials;
                jlast = g;
                if ((bad >> (g * W)) & group_mask) temp_chi = DBL_MAX;
                const double scale = scv[g] + 1e-3;
                rho = (cur_chi - temp_chi) / scale;
                if (rho > 0.0 && fabs(temp_chi) <= DBL_MAX) {
                    const double r21 = 2.0 * rho - 1.0;
                    double alpha = 1.0 - r21 * r21 * r21;
                    alpha = fmin(alpha, good_hi);
                    lambda = lamv[g] * fmax(good_lo, alpha);
                    ni = 2.0;
                    cur_chi = temp_chi;
                    cur = slot;   // the trial state is the state
                    ++q;
                    iteration_over = true;
                } else {
                    lambda = lamv[g] * niv[g];
                    ni = 2.0 * niv[g];      // (pop: the state was never overwritten)
                    ++q;
                    iteration_over = !(rho < 0.0 && q < max_trials);
                }
            }
        }
        W3_T(5);
        if (iteration_over) {
            ++it;
            need_lin = true;
            if (q == max_trials || rho == 0.0) { terminated = 1; done = true; }
            if (it >= a.iterations) done = true;
        }
        W3_T(6);
    }
    if (lane < nv) {
        const double* s = l.T + ((size_t)cur * nvm + lane) * 3;
        if (gin != gout) {   // the rotations never move
#pragma unroll
            for (int k = 0; k < 9; ++k) gout[lane * 12 + k] = gin[lane * 12 + k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) gout[lane * 12 + 9 + k] = s[k];
    }
    if (lane == 0) {
        double* res = a.result + (size_t)inst * 8;
        res[0] = last_plain; res[1] = cur_chi; res[2] = lambda; res[3] = (double)it; res[4] = (double)trials;
        res[5] = (double)terminated; res[6] = (double)shared_edges; res[7] = nv > 0 ? (double)(nv * 65536 + 2 * nv - 1) : 0.0;
#ifdef LOCAMD_WAVE3_TIMING
        for (int k = 0; k < 7; ++k) res[k] = (double)w3_ph[k];
        res[6] = (double)w3_ph[7];   // rounds
        res[7] = (double)(__builtin_readcyclecounter() - w3_start) + 1e12 * trials;
#endif
    }
}

}  // namespace

size_t window_wave3_lds_bytes(const WindowCaps& c) {
    const size_t doubles = (size_t)5 * c.nv_max * 3 + (size_t)c.nr_max * 21 + (size_t)c.np_max * 6;
    const size_t ints = (size_t)c.nr_max * 4 + c.np_max;
    return doubles * sizeof(double) + ((ints + 1) & ~(size_t)1) * sizeof(int);
}

hipError_t launch_window_wave3(const WindowArgs& a, hipStream_t stream) {
    if (a.B <= 0 || a.caps.nv_max > 64 || a.caps.nv_max <= 0) return hipErrorInvalidValue;
    const size_t lds = window_wave3_lds_bytes(a.caps);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    if (a.jacobian) hipLaunchKernelGGL((wave3_lm_kernel<1>), dim3((unsigned)a.B), dim3(64), lds, stream, a);
    else hipLaunchKernelGGL((wave3_lm_kernel<0>), dim3((unsigned)a.B), dim3(64), lds, stream, a);
    return hipGetLastError();
}

}  // namespace locamd
