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
// (g2o semantics: SURVEY.md Appendix A.  Jacobians are analytic here; the reference's numeric range Jacobian is the
//  same derivative up to 1e-7 relative noise.)
//
// MI355X mapping: ONE WAVE PER INSTANCE, everything in LDS.
//   * edge evaluation: lane e evaluates edge e (error, Jacobian blocks, robust weight) into an LDS record;
//   * normal equations: edges are folded into H one after the other (fixed order => bit-reproducible, no atomics),
//     the 64 lanes covering the 6x6 block entries;  H (n = 6 * poses <= 96) lives in the UPPER triangle of one
//     n x ld LDS matrix (ld odd: conflict-free column walks), its Cholesky factor goes into the strict LOWER triangle,
//     so a rejected LM trial needs no copy of H;
//   * left-looking Cholesky with lanes over rows, column-oriented triangular solves, lane-per-pose oplus;
//   * all LM control flow is wave-uniform (sums are xor-butterflies: every lane holds the same bits).
// This is the latency-oriented first version of the window path (one 64-lane wave per 60..96-unknown system);
// DESIGN.md lists what the throughput version changes.
#include "window_kernel.h"
#include "device_math.h"

#include <float.h>
#include <math.h>

namespace locamd {

namespace {

constexpr int RREC = 16;   // J0[6] J1[6] wr omega_r chi rho0
constexpr int PREC = 56;   // J[36] W[6] omega_r[6] (+pad)
constexpr int SREC = 152;  // J0[36] J1[36] WJ0[36] WJ1[36] omega_r[6] (+pad)

// Wave-wide reductions on the VALU (DPP row shifts + row broadcasts, then one readlane): every lane gets the same
// bits, no LDS crossbar round trips (a __shfl_xor butterfly on doubles costs ~6 dependent ds_bpermute pairs).
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
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_or_zero<0x111, 0xF>(v, 0.0);  // row_shr:1
    v += dpp_or_zero<0x112, 0xF>(v, 0.0);  // row_shr:2
    v += dpp_or_zero<0x114, 0xF>(v, 0.0);  // row_shr:4
    v += dpp_or_zero<0x118, 0xF>(v, 0.0);  // row_shr:8  -> lane 15 of each row holds the row sum
    v += dpp_or_zero<0x142, 0xA>(v, 0.0);  // row_bcast:15 into rows 1 and 3
    v += dpp_or_zero<0x143, 0xC>(v, 0.0);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return read_lane63(v);
}
__device__ __forceinline__ double wave_max(double v) {  // for non-negative inputs (identity 0)
    v = fmax(v, dpp_or_zero<0x111, 0xF>(v, 0.0));
    v = fmax(v, dpp_or_zero<0x112, 0xF>(v, 0.0));
    v = fmax(v, dpp_or_zero<0x114, 0xF>(v, 0.0));
    v = fmax(v, dpp_or_zero<0x118, 0xF>(v, 0.0));
    v = fmax(v, dpp_or_zero<0x142, 0xA>(v, 0.0));
    v = fmax(v, dpp_or_zero<0x143, 0xC>(v, 0.0));
    return read_lane63(v);
}

// ---- small SE3 algebra (row-major 3x3) -------------------------------------------------------------------------
__device__ __forceinline__ void mat_mul(const double* A, const double* B, double* C) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j] + A[i * 3 + 2] * B[2 * 3 + j];
}
__device__ __forceinline__ void mat_tmul(const double* A, const double* B, double* C) {  // A^T B
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[0 * 3 + i] * B[0 * 3 + j] + A[1 * 3 + i] * B[1 * 3 + j] + A[2 * 3 + i] * B[2 * 3 + j];
}
__device__ __forceinline__ void mat_vec(const double* A, const double* v, double* o) {
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = A[i * 3 + 0] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2];
}
__device__ __forceinline__ void mat_tvec(const double* A, const double* v, double* o) {  // A^T v
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = A[0 * 3 + i] * v[0] + A[1 * 3 + i] * v[1] + A[2 * 3 + i] * v[2];
}
// Eigen::Quaternion(Matrix3) — q = (w, x, y, z).  Scalars, not an array: the optimiser otherwise merges the branches
// into a computed index and the array lands in scratch memory.
__device__ __forceinline__ void mat_to_quat(const double* R, double* q) {
    double qw, qx, qy, qz;
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        qw = 0.5 * t; t = 0.5 / t;
        qx = (R[7] - R[5]) * t; qy = (R[2] - R[6]) * t; qz = (R[3] - R[1]) * t;
    } else if (R[0] >= R[4] && R[0] >= R[8]) {  // i = 0, j = 1, k = 2
        t = sqrt(R[0] - R[4] - R[8] + 1.0);
        qx = 0.5 * t; t = 0.5 / t;
        qw = (R[7] - R[5]) * t; qy = (R[3] + R[1]) * t; qz = (R[6] + R[2]) * t;
    } else if (R[4] > R[0] && R[4] >= R[8]) {   // i = 1, j = 2, k = 0
        t = sqrt(R[4] - R[8] - R[0] + 1.0);
        qy = 0.5 * t; t = 0.5 / t;
        qw = (R[2] - R[6]) * t; qz = (R[7] + R[5]) * t; qx = (R[1] + R[3]) * t;
    } else {                                      // i = 2, j = 0, k = 1
        t = sqrt(R[8] - R[0] - R[4] + 1.0);
        qz = 0.5 * t; t = 0.5 / t;
        qw = (R[3] - R[1]) * t; qx = (R[2] + R[6]) * t; qy = (R[5] + R[7]) * t;
    }
    q[0] = qw; q[1] = qx; q[2] = qy; q[3] = qz;
}
__device__ __forceinline__ void quat_mul(const double* a, const double* b, double* o) {
    o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}
// g2o internal::normalize: unit norm, w >= 0; returns the sign applied
__device__ __forceinline__ double quat_normalize_sign(double* q) {
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    double s = 1.0 / n, sg = 1.0;
    if (q[0] < 0) { s = -s; sg = -1.0; }
    q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s;
    return sg;
}
// Eigen toRotationMatrix (no normalisation)
__device__ __forceinline__ void quat_to_mat(const double* q, double* R) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
// rows 3..5 x cols 3..5 of a Jacobian: d vec(q (x) (sqrt(1-|v|^2), v)) / dv at 0 = w I + [q_xyz]x
__device__ __forceinline__ void quat_right_jac(const double* q, double sgn, double* J, int ldj) {
    const double w = q[0] * sgn, x = q[1] * sgn, y = q[2] * sgn, z = q[3] * sgn;
    J[3 * ldj + 3] = w;  J[3 * ldj + 4] = -z; J[3 * ldj + 5] = y;
    J[4 * ldj + 3] = z;  J[4 * ldj + 4] = w;  J[4 * ldj + 5] = -x;
    J[5 * ldj + 3] = -y; J[5 * ldj + 4] = x;  J[5 * ldj + 5] = w;
}

struct Lds {
    double *Hs, *Ls;  // skyline H (lower) and its Cholesky factor (LDS, or an HBM workspace slice for large windows)
    double *diagL, *b, *x, *yrow, *pose, *bak, *rrec, *prec, *srec;
    int *fb, *last, *boff;  // per block row: first / last connected block, offset of the block row's storage
    int *ioff, *ilist;      // per pose: its incident edges in fold order (CSR), entry = kind << 28 | role << 27 | edge
    int *shared;            // [0] = count, then the binary edges (range e, or nr + SE3 e) whose pair of poses has another edge, in fold order
    double* blk; // 6x6 scratch: the diagonal block being factored
#ifdef LOCAMD_WINDOW_TIMING
    long long* tim;  // diagnostic build: cycles in (a) segments, (b) block exchange+factor, (c) row finish, back-substitution
#endif
    // this instance's edge tables, staged from HBM once per launch
    const int32_t *r_idx, *p_idx, *s_idx;
    const double *r_val, *p_val, *s_val;
};

// Evaluate every edge at the current poses: errors + chi sums always, Jacobian/weight records when FULL.
template <bool FULL>
__device__ __forceinline__ void evaluate_edges(const WindowArgs& a, const Lds& L, int inst, int lane, int nr, int np, int ns,
                               double& robust_chi, double& plain_chi) {
    (void)inst;
    double rsum = 0.0, csum = 0.0;
    // ---- range edges ------------------------------------------------------------------------------------------
    for (int e = lane; e < nr; e += 64) {
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
        double u[3] = {p0[0] - p1[0], p0[1] - p1[1], p0[2] - p1[2]};
        const double n = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        const double err = meas - n;
        const double chi = err * (info * err);
        const double aux = 1.0 + chi;
        rsum += log(aux);
        csum += chi;
        if (FULL) {
            double* rec = L.rrec + e * RREC;
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
            } else { rec[6] = 0; rec[7] = 0; rec[8] = 0; }
            rec[9] = 0; rec[10] = 0; rec[11] = 0;
            const double wr = info / aux;  // rho' * Omega
            rec[12] = wr;
            rec[13] = -wr * err;           // omega_r
        }
    }
    // ---- unary priors -------------------------------------------------------------------------------------------
    for (int e = lane; e < np; e += 64) {
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
#pragma unroll
            for (int i = 0; i < 36; ++i) rec[i] = 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) rec[i * 6 + j] = RE[i * 3 + j];
            quat_right_jac(q, 1.0, rec, 6);
#pragma unroll
            for (int i = 0; i < 6; ++i) { rec[36 + i] = val[12 + i]; rec[42 + i] = -val[12 + i] * err[i]; }
        }
    }
    // ---- binary SE3 edges ---------------------------------------------------------------------------------------
    for (int e = lane; e < ns; e += 64) {
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
        rsum += robust ? log(aux) : chi;
        csum += chi;
        if (FULL) {
            double* rec = L.srec + e * SREC;
            const double w = robust ? 1.0 / aux : 1.0;
            double* J0 = rec; double* J1 = rec + 36;
#pragma unroll
            for (int i = 0; i < 72; ++i) rec[i] = 0.0;
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
            // WJ = w * Omega * J, omega_r = -w * Omega * e
#pragma unroll
            for (int i = 0; i < 6; ++i) {
#pragma unroll
                for (int cc = 0; cc < 6; ++cc) {
                    double s0 = 0.0, s1 = 0.0;
#pragma unroll
                    for (int j = 0; j < 6; ++j) { s0 += Om[i * 6 + j] * J0[j * 6 + cc]; s1 += Om[i * 6 + j] * J1[j * 6 + cc]; }
                    rec[72 + i * 6 + cc] = w * s0;
                    rec[108 + i * 6 + cc] = w * s1;
                }
                rec[144 + i] = -w * Oe[i];
            }
        }
    }
    robust_chi = wave_sum(rsum);
    plain_chi = wave_sum(csum);
}

// ---- skyline storage ---------------------------------------------------------------------------------------------
// H (lower triangle) and its Cholesky factor live in SKYLINE form: the six rows of pose block v keep columns
// [6 fb[v], 6 v + 5], fb[v] = the leftmost block v is connected to (itself if none).  Cholesky fill-in never leaves that
// envelope, so storage and work are O(n * band^2) instead of O(n^2) / O(n^3): a 500-pose chain (cfg/uwb_pose.yaml) is
// 3000 rows of ~12 entries.  A block row is stored COLUMN-MAJOR (6 rows x its columns): entry (6 v + r, col) sits at
// boff[v] + 6 (col - 6 fb[v]) + r, so the six lanes working on one block row read one contiguous 48-byte piece per column
// and a 6-column step of the sweep touches 288 contiguous bytes per block row (it was six separate row segments);
// the back-substitution's per-column reads are contiguous across lanes.  boff[nv] = stored entries.
__host__ __device__ inline size_t sky_nnz_bound(int nv, int bw) {
    size_t s = 0;
    for (int v = 0; v < nv; ++v) s += 36 * ((size_t)(v < bw ? v : bw) + 1);
    return s;
}

// entries of the per-pose incidence lists: every edge once per moving endpoint
__host__ __device__ inline size_t window_incidences(const WindowCaps& c) {
    return 2 * (size_t)c.nr_max + (size_t)c.np_max + 2 * (size_t)c.ns_max;
}
// doubles of one instance's arrays (skyline pair, dense vectors, poses, edge records, index tables) — the layout the
// kernel carves, in LDS or in the HBM workspace
__host__ __device__ inline size_t window_instance_doubles(const WindowCaps& c) {
    const size_t n_max = 6 * (size_t)c.nv_max;
    return 2 * sky_nnz_bound(c.nv_max, c.bw_max) + 4 * n_max + 2 * (size_t)c.nv_max * 12 + (size_t)c.nr_max * RREC + (size_t)c.np_max * PREC +
           (size_t)c.ns_max * SREC + 2 * (((size_t)c.nv_max + 1) / 2) + (n_max + 2) / 2 + ((size_t)c.nv_max + 2) / 2 +
           (window_incidences(c) + 1) / 2 + ((size_t)c.nr_max + (size_t)c.ns_max + 2) / 2;
}

// address of H/L entry (row, col), col <= row, col inside row's envelope
__device__ __forceinline__ int sky(const Lds& L, int row, int col) {
    const int v = row / 6;
    return L.boff[v] + 6 * (col - 6 * L.fb[v]) + (row - 6 * v);
}

constexpr int INC_KIND_SHIFT = 28, INC_ROLE_SHIFT = 27, INC_EDGE_MASK = (1 << 27) - 1;

// Once per solve (the topology does not change between iterations): fb[], last[], boff[], the incidence lists, and
// the list of binary edges that share their pair of poses with another edge.
__device__ __forceinline__ void compute_skyline(const Lds& L, int lane, int n, int nr, int np, int ns) {
    const int nv = n / 6;
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
        // incidence lists in the order the fold visits the edges: ranges, priors, SE3 (stable counting sort by pose)
        for (int v = 0; v <= nv; ++v) L.ioff[v] = 0;
        for (int e = 0; e < nr; ++e) { ++L.ioff[L.r_idx[2 * e] + 1]; if (L.r_idx[2 * e + 1] >= 0) ++L.ioff[L.r_idx[2 * e + 1] + 1]; }
        for (int e = 0; e < np; ++e) ++L.ioff[L.p_idx[e] + 1];
        for (int e = 0; e < ns; ++e) { ++L.ioff[L.s_idx[4 * e] + 1]; ++L.ioff[L.s_idx[4 * e + 1] + 1]; }
        for (int v = 0; v < nv; ++v) L.ioff[v + 1] += L.ioff[v];
        // (last[] doubles as the write cursor while the lists are filled; it is rebuilt right after)
        for (int v = 0; v < nv; ++v) L.last[v] = L.ioff[v];
        for (int e = 0; e < nr; ++e) {
            const int v0 = L.r_idx[2 * e], v1 = L.r_idx[2 * e + 1];
            L.ilist[L.last[v0]++] = e;
            if (v1 >= 0) L.ilist[L.last[v1]++] = (1 << INC_ROLE_SHIFT) | e;
        }
        for (int e = 0; e < np; ++e) L.ilist[L.last[L.p_idx[e]]++] = (1 << INC_KIND_SHIFT) | e;
        for (int e = 0; e < ns; ++e) {
            L.ilist[L.last[L.s_idx[4 * e]]++] = (2 << INC_KIND_SHIFT) | e;
            L.ilist[L.last[L.s_idx[4 * e + 1]]++] = (2 << INC_KIND_SHIFT) | (1 << INC_ROLE_SHIFT) | e;
        }
        for (int v = 0; v < nv; ++v) L.last[v] = v;
        for (int v = 0; v < nv; ++v) for (int J = L.fb[v]; J <= v; ++J) L.last[J] = max(L.last[J], v);
    }
    __syncthreads();
    // Binary edges that share their pair of poses with another one: every binary edge stamps its number on the first
    // entry of its off-diagonal block (H is not in use yet); whoever does not read its own stamp back overwrites it with
    // -1, and after that everybody on a shared pair reads -1.  Lane 0 lists those edges in fold order.
    auto block_entry = [&](int t) {
        const bool is_r = t < nr;
        const int va = is_r ? L.r_idx[2 * t] : L.s_idx[4 * (t - nr)], vb = is_r ? L.r_idx[2 * t + 1] : L.s_idx[4 * (t - nr) + 1];
        return vb >= 0 ? sky(L, 6 * max(va, vb), 6 * min(va, vb)) : -1;
    };
    for (int t = lane; t < nr + ns; t += 64) { const int a0 = block_entry(t); if (a0 >= 0) L.Hs[a0] = (double)(t + 1); }
    __syncthreads();
    for (int t = lane; t < nr + ns; t += 64) { const int a0 = block_entry(t); if (a0 >= 0 && L.Hs[a0] != (double)(t + 1)) L.Hs[a0] = -1.0; }
    __syncthreads();
    if (lane == 0) {
        int cnt = 0;
        for (int t = 0; t < nr + ns; ++t) {
            const int a0 = block_entry(t);
            if (a0 < 0) continue;
            const double stamp = L.Hs[a0];
            if (stamp == -1.0) { L.shared[1 + cnt++] = (1 << INC_KIND_SHIFT) | t; L.Hs[a0] = -2.0; }  // first edge of its pair
            else if (stamp == -2.0) L.shared[1 + cnt++] = t;
        }
        L.shared[0] = cnt;
    }
    __syncthreads();
}
// Fold the edge records into H (skyline lower triangle) and b: H_vv = sum J_v^T (rho' Omega) J_v etc., every entry summed
// over its edges in one fixed order (ranges, priors, SE3; bit-reproducible, no atomics).  It is a gather: task
// (pose v, entry) walks v's incidence list and sums that entry of the diagonal block / of b in a register, one store at
// the end; an off-diagonal block that belongs to one edge is written by it; the few pairs of poses with several edges
// (a key-frame pose edge landing on the previous pose next to the smoothness edge) are accumulated edge by edge at the
// end.  No read-modify-write chains through memory for the bulk, all tasks independent.
__device__ __forceinline__ void build_system(const Lds& L, int lane, int n, int nr, int ns) {
    const int nv = n / 6;
    const int nnz = L.boff[nv];
    for (int i = lane; i < nnz; i += 64) L.Hs[i] = 0.0;
    __syncthreads();
    for (int task = lane; task < nv * 27; task += 64) {
        const int v = task / 27, k = task - 27 * v;  // k < 21: lower-triangle entry (r, cc) of the diagonal block; else b[k - 21]
        int r = 0;
        while ((r + 1) * (r + 2) / 2 <= k && r < 5) ++r;
        const int cc = k - r * (r + 1) / 2;
        const bool is_b = k >= 21;
        const int rb = k - 21;
        double acc = 0.0;
        const int p1 = L.ioff[v + 1];
        for (int p = L.ioff[v]; p < p1; ++p) {
            const int code = L.ilist[p];
            const int kind = code >> INC_KIND_SHIFT, role = (code >> INC_ROLE_SHIFT) & 1, e = code & INC_EDGE_MASK;
            if (kind == 0) {
                const double* rec = L.rrec + e * RREC;
                const double* J = rec + 6 * role;
                if (is_b) acc += J[rb] * rec[13];
                else acc += rec[12] * J[r] * J[cc];
            } else if (kind == 1) {
                const double* rec = L.prec + e * PREC;
                double t = 0.0;
                if (is_b) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) t += rec[i * 6 + rb] * rec[42 + i];
                } else {
#pragma unroll
                    for (int i = 0; i < 6; ++i) t += rec[i * 6 + r] * rec[36 + i] * rec[i * 6 + cc];
                }
                acc += t;
            } else {
                const double* rec = L.srec + e * SREC;
                const double* J = rec + 36 * role;
                double t = 0.0;
                if (is_b) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) t += J[i * 6 + rb] * rec[144 + i];
                } else {
                    const double* WJ = rec + 72 + 36 * role;
#pragma unroll
                    for (int i = 0; i < 6; ++i) t += J[i * 6 + r] * WJ[i * 6 + cc];
                }
                acc += t;
            }
        }
        if (is_b) L.b[v * 6 + rb] = acc;
        else L.Hs[sky(L, v * 6 + r, v * 6 + cc)] = acc;
    }
    // off-diagonal blocks: one task per (binary edge, entry); the value of entry (r, cc) of edge t
    auto offdiag = [&](int t, int r, int cc, int& addr) {
        double h = 0.0;
        if (t < nr) {
            const int v0 = L.r_idx[2 * t], v1 = L.r_idx[2 * t + 1];
            const double* rec = L.rrec + t * RREC;
            const double wr = rec[12];
            if (v0 > v1) { h = wr * rec[r] * rec[6 + cc]; addr = sky(L, v0 * 6 + r, v1 * 6 + cc); }
            else         { h = wr * rec[6 + r] * rec[cc]; addr = sky(L, v1 * 6 + r, v0 * 6 + cc); }
        } else {
            const int e = t - nr;
            const int vi = L.s_idx[4 * e], vj = L.s_idx[4 * e + 1];
            const double* rec = L.srec + e * SREC;
            const double *J0 = rec, *J1 = rec + 36, *WJ0 = rec + 72, *WJ1 = rec + 108;
#pragma unroll
            for (int i = 0; i < 6; ++i) h += (vi > vj) ? J0[i * 6 + r] * WJ1[i * 6 + cc] : J1[i * 6 + r] * WJ0[i * 6 + cc];  // rows of the later pose
            addr = (vi > vj) ? sky(L, vi * 6 + r, vj * 6 + cc) : sky(L, vj * 6 + r, vi * 6 + cc);
        }
        return h;
    };
    for (int task = lane; task < (nr + ns) * 36; task += 64) {
        const int t = task / 36, q = task - 36 * t;
        if (t < nr && L.r_idx[2 * t + 1] < 0) continue;  // a range to a fixed anchor has no off-diagonal block
        int addr;
        const double h = offdiag(t, q / 6, q - 6 * (q / 6), addr);
        L.Hs[addr] = h;   // (for a pair with several edges this is overwritten below)
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
            const double h = offdiag(t, lane / 6, lane % 6, addr);
            L.Hs[addr] = first ? h : L.Hs[addr] + h;
        }
    }
    __syncthreads();
}

// (H + lambda I) x = b in one sweep over 6x6 block columns (poses are 6-DoF blocks, n = 6 nv):
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
#ifdef LOCAMD_WINDOW_TIMING
            long long ts0 = clock64();
#endif
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
#ifdef LOCAMD_WINDOW_TIMING
            long long ts1 = clock64();
#endif
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
#ifdef LOCAMD_WINDOW_TIMING
            long long ts2 = clock64();
#endif
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
#ifdef LOCAMD_WINDOW_TIMING
            if (lane == 0) { const long long ts3 = clock64(); L.tim[0] += ts1 - ts0; L.tim[1] += ts2 - ts1; L.tim[2] += ts3 - ts2; }
#endif
        }
        __syncthreads();
    }
#ifdef LOCAMD_WINDOW_TIMING
    const long long tb0 = clock64();
#endif
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
#ifdef LOCAMD_WINDOW_TIMING
    if (lane == 0) L.tim[3] += clock64() - tb0;
#endif
    return true;
}

// GLOBAL_A: the skyline arrays (H and its factor) live in an HBM workspace slice instead of LDS (large windows);
// a workgroup is one wave on one CU, whose L1 is coherent for its own stores after the workgroup barrier.
template <bool GLOBAL_A>
__global__ void __launch_bounds__(64) window_lm_kernel(const WindowArgs a) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int inst = blockIdx.x;
    const int lane = threadIdx.x;
    const WindowCaps& c = a.caps;
    const int nv = a.counts[inst * 4 + 0], nr = a.counts[inst * 4 + 1], np = a.counts[inst * 4 + 2], ns = a.counts[inst * 4 + 3];
    const int n = 6 * nv;
    const int n_max = 6 * c.nv_max;
    const size_t nnz_max = sky_nnz_bound(c.nv_max, c.bw_max);
    Lds L;
    // Small windows: everything per-instance lives in LDS.  Large windows (GLOBAL_A): everything lives in this
    // instance's slice of the HBM workspace (L1/L2-cached), edge tables are read where the caller put them; only the
    // 6x6 exchange block stays in LDS.
    __shared__ double s_blk[36];
#ifdef LOCAMD_WINDOW_TIMING
    __shared__ long long s_tim[4];
    if (lane < 4) s_tim[lane] = 0;
    L.tim = s_tim;
#endif
    double* p = GLOBAL_A ? a.workspace + (size_t)inst * window_instance_doubles(c) : lds;
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
    if (GLOBAL_A) {
        // the three small index tables stay in LDS even when everything else is in the HBM workspace: every address in
        // the sweep and in the edge fold starts with a lookup in them, and an HBM round trip there is pure latency
        int* t = reinterpret_cast<int*>(lds);
        L.fb = t; L.last = t + c.nv_max; L.boff = t + 2 * c.nv_max; L.ioff = t + 3 * c.nv_max + 1;
        p += 2 * ((c.nv_max + 1) / 2) + (n_max + 2) / 2 + (c.nv_max + 2) / 2;  // (their workspace slots stay unused)
    } else {
        L.fb = reinterpret_cast<int*>(p); p += (c.nv_max + 1) / 2;
        L.last = reinterpret_cast<int*>(p); p += (c.nv_max + 1) / 2;
        L.boff = reinterpret_cast<int*>(p); p += (n_max + 2) / 2;
        L.ioff = reinterpret_cast<int*>(p); p += (c.nv_max + 2) / 2;
    }
    L.ilist = reinterpret_cast<int*>(p); p += (window_incidences(c) + 1) / 2;
    L.shared = reinterpret_cast<int*>(p); p += (c.nr_max + c.ns_max + 2) / 2;
    double* gpose = a.poses + (size_t)inst * c.nv_max * 12;
    for (int i = lane; i < nv * 12; i += 64) L.pose[i] = gpose[i];
    if (GLOBAL_A) {
        L.r_idx = a.r_idx + (size_t)inst * c.nr_max * 2; L.r_val = a.r_val + (size_t)inst * c.nr_max * 5;
        L.p_idx = a.p_idx + (size_t)inst * c.np_max;     L.p_val = a.p_val + (size_t)inst * c.np_max * 18;
        L.s_idx = a.s_idx + (size_t)inst * c.ns_max * 4; L.s_val = a.s_val + (size_t)inst * c.ns_max * 48;
    } else {
        double* st_rval = p; p += c.nr_max * 5;
        double* st_pval = p; p += c.np_max * 18;
        double* st_sval = p; p += c.ns_max * 48;
        int32_t* st_ridx = reinterpret_cast<int32_t*>(p); p += c.nr_max;        // nr_max * 2 ints
        int32_t* st_pidx = reinterpret_cast<int32_t*>(p); p += (c.np_max + 1) / 2;
        int32_t* st_sidx = reinterpret_cast<int32_t*>(p);                        // ns_max * 4 ints
        L.r_idx = st_ridx; L.p_idx = st_pidx; L.s_idx = st_sidx; L.r_val = st_rval; L.p_val = st_pval; L.s_val = st_sval;
        for (int i = lane; i < nr * 2; i += 64) st_ridx[i] = a.r_idx[(size_t)inst * c.nr_max * 2 + i];
        for (int i = lane; i < nr * 5; i += 64) st_rval[i] = a.r_val[(size_t)inst * c.nr_max * 5 + i];
        for (int i = lane; i < np; i += 64) st_pidx[i] = a.p_idx[(size_t)inst * c.np_max + i];
        for (int i = lane; i < np * 18; i += 64) st_pval[i] = a.p_val[(size_t)inst * c.np_max * 18 + i];
        for (int i = lane; i < ns * 4; i += 64) st_sidx[i] = a.s_idx[(size_t)inst * c.ns_max * 4 + i];
        for (int i = lane; i < ns * 48; i += 64) st_sval[i] = a.s_val[(size_t)inst * c.ns_max * 48 + i];
    }
    __syncthreads();
    compute_skyline(L, lane, n, nr, np, ns);

    constexpr double tau = 1e-5, good_lo = 1.0 / 3.0, good_hi = 2.0 / 3.0;
    constexpr int max_trials = 10;
    double lambda = 0.0, ni = 2.0, cur_chi = 0.0, last_plain = 0.0;
#ifdef LOCAMD_WINDOW_TIMING  // diagnostic build only: phase cycle counts into result[6], result[7]
    long long t_fs = 0, t_ev = 0, t_bd = 0, tt0 = 0; const long long t_start = clock64();
#define LOCAMD_T0() tt0 = clock64()
#define LOCAMD_T1(acc) acc += clock64() - tt0
#else
#define LOCAMD_T0()
#define LOCAMD_T1(acc)
#endif
    int it = 0, trials = 0, terminated = 0;
    const bool empty = (nv <= 0) || (nr + np + ns <= 0);

    bool ok = !empty;
    for (it = 0; it < a.iterations && ok; ++it) {
        double plain;
        LOCAMD_T0();
        evaluate_edges<true>(a, L, inst, lane, nr, np, ns, cur_chi, plain);  // computeActiveErrors + linearize
        last_plain = plain;
        __syncthreads();
        build_system(L, lane, n, nr, ns);
        LOCAMD_T1(t_bd);
        if (it == 0) {  // computeLambdaInit
            double md = 0.0;
            for (int j = lane; j < n; j += 64) md = fmax(md, fabs(L.Hs[sky(L, j, j)]));
            lambda = tau * wave_max(md);
            ni = 2.0;
        }
        double rho = 0.0;
        int q = 0;
        do {
            for (int i = lane; i < nv * 12; i += 64) L.bak[i] = L.pose[i];  // push
            LOCAMD_T0();
            const bool ok2 = factor_and_solve(L, lane, n, lambda);
            LOCAMD_T1(t_fs);
            if (!ok2) { for (int i = lane; i < n; i += 64) L.x[i] = 0.0; __syncthreads(); }
            // update: X <- X * fromVectorMQT(dx), one pose per lane
            for (int v = lane; v < nv; v += 64) {
                const double* dx = L.x + v * 6;
                double* X = L.pose + v * 12;
                double Rd[9];
                const double ww = 1.0 - (dx[3] * dx[3] + dx[4] * dx[4] + dx[5] * dx[5]);
                if (ww < 0) { Rd[0] = 1; Rd[1] = 0; Rd[2] = 0; Rd[3] = 0; Rd[4] = 1; Rd[5] = 0; Rd[6] = 0; Rd[7] = 0; Rd[8] = 1; }
                else { const double qd[4] = {sqrt(ww), dx[3], dx[4], dx[5]}; quat_to_mat(qd, Rd); }
                double Rn[9], tn[3];
                mat_mul(X, Rd, Rn);
                mat_vec(X, dx, tn);
                X[9] += tn[0]; X[10] += tn[1]; X[11] += tn[2];
#pragma unroll
                for (int i = 0; i < 9; ++i) X[i] = Rn[i];
            }
            __syncthreads();
            ++trials;
            double temp_chi, plain2;
            LOCAMD_T0();
            evaluate_edges<false>(a, L, inst, lane, nr, np, ns, temp_chi, plain2);
            LOCAMD_T1(t_ev);
            last_plain = plain2;
            if (!ok2) temp_chi = DBL_MAX;
            double sc = 0.0;
            for (int j = lane; j < n; j += 64) sc += L.x[j] * (lambda * L.x[j] + L.b[j]);  // computeScale
            const double scale = wave_sum(sc) + 1e-3;
            rho = (cur_chi - temp_chi) / scale;
            if (rho > 0.0 && fabs(temp_chi) < DBL_MAX && ok2) {
                const double r21 = 2.0 * rho - 1.0;
                double alpha = 1.0 - r21 * r21 * r21;
                alpha = fmin(alpha, good_hi);
                lambda *= fmax(good_lo, alpha);
                ni = 2.0;
                cur_chi = temp_chi;
            } else {
                lambda *= ni;
                ni *= 2.0;
                __syncthreads();
                for (int i = lane; i < nv * 12; i += 64) L.pose[i] = L.bak[i];  // pop
            }
            __syncthreads();
            ++q;
        } while (rho < 0.0 && q < max_trials);
        if (q == max_trials || rho == 0.0) { ok = false; terminated = 1; }
    }

    for (int i = lane; i < nv * 12; i += 64) gpose[i] = L.pose[i];
    if (lane == 0) {
        double* res = a.result + (size_t)inst * 8;
        res[0] = last_plain; res[1] = cur_chi; res[2] = lambda; res[3] = (double)it; res[4] = (double)trials;
        res[5] = (double)terminated; res[6] = (double)L.shared[0]; res[7] = 0.0;
#ifdef LOCAMD_WINDOW_TIMING
        res[6] = (double)t_fs * 1e6 + (double)t_ev * 1e-3; res[7] = (double)t_bd * 1e6 + (double)(clock64() - t_start) * 1e-3;
        res[0] = (double)L.tim[0]; res[1] = (double)L.tim[1]; res[2] = (double)L.tim[2]; res[5] = (double)L.tim[3];
#endif
    }
}

}  // namespace

size_t window_lds_bytes(const WindowCaps& c, bool global_a) {
    if (global_a) return (4 * (size_t)c.nv_max + 4) * sizeof(int);  // fb, last, boff, ioff (+ the static 6x6 exchange block)
    const size_t tables = (size_t)c.nr_max * 6 + (size_t)c.np_max * 18 + (c.np_max + 1) / 2 + (size_t)c.ns_max * 50;
    return (window_instance_doubles(c) + tables) * sizeof(double);
}
size_t window_workspace_doubles(const WindowCaps& c) { return window_instance_doubles(c); }

template <bool GLOBAL_A>
static hipError_t launch_window_t(const WindowArgs& a, size_t lds, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&window_lm_kernel<GLOBAL_A>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);  // 288 B of static LDS on top
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((window_lm_kernel<GLOBAL_A>), dim3((unsigned)a.B), dim3(64), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_window(const WindowArgs& a, hipStream_t stream) {
    if (a.B <= 0 || a.caps.bw_max < 0 || a.caps.bw_max >= a.caps.nv_max + (a.caps.nv_max == 1)) return hipErrorInvalidValue;
    const bool global_a = a.workspace != nullptr;
    const size_t lds = window_lds_bytes(a.caps, global_a);
    if (lds > 160 * 1024 - 512) return hipErrorInvalidValue;
    return global_a ? launch_window_t<true>(a, lds, stream) : launch_window_t<false>(a, lds, stream);
}

}  // namespace locamd
