// gfx950 (MI355X / CDNA4) fusion snapshot kernel — BASELINE config 3: UWB ranges + IMU orientation prior, 6-DoF state.
//
// Per tag and epoch (reference call sites in brackets):
//   1. the IMU quaternion overwrites the pose's rotation, translation kept        [localization.cpp:505-513]
//   2. unary EdgeSE3Prior, measurement = that pose, information diag(0,0,0, 1/c0, 1/c4, 1/c8), not robust
//      e = toVectorMQT(Z^-1 X) -> only the quaternion-vector rows count                     [localization.cpp:515-525]
//   3. M range factors e = d - ||R o + t - a_m|| with the antenna lever arm o on the tag side, information 1/err^2,
//      Cauchy; outlier gate on the vertex ORIGINS (no lever arm)   [localization.cpp:306-313, 331-336, 608-627;
//                                                                   types_edge_se3range.cpp:105-114]
//   4. Localization::solve(): g2o LM on the 6-DoF vertex (X <- X * fromVectorMQT(dx)), then optimizer.chi2()
//                                                                                   [localization.cpp:164-170, 197]
// One lane per tag; the 6x6 normal equations (21 + 6 doubles) live in registers; the LM / epoch-window machinery is
// the snapshot kernel's (snapshot_kernel.hip: flattened LM passes, desynchronised lanes, two-epoch window with the
// next epoch converted in LDS and the one after in flight through LDS-DMA).
// HBM layout: dist/err float4 tiles [K][2][B]; imu double [K][B][8] = q xyzw, cov c0 c4 c8, pad (64 B per tag: four
// 16-byte LDS-DMA pieces); pose state double [7][B] (t, q xyzw); outputs pose [K][7][B], chi2 [K][B], trials [K][B].
#include "fusion_kernel.h"
#include "device_math.h"
#include "numeric_jacobian.h"

#include <float.h>
#include <math.h>

namespace locamd {

namespace {

constexpr int M8 = 8;  // padded anchor count handled by this kernel

struct Sys6 {
    double h[21];  // upper triangle, row-major: (0,0) (0,1) .. (0,5) (1,1) .. (5,5)
    double b[6];
    double rchi, chi;
};
__device__ __forceinline__ constexpr int hidx(int r, int c) { return r * 6 - (r * (r - 1)) / 2 + (c - r); }

struct Pose { double R[9]; double t[3]; };

// Eigen::Quaternion(Matrix3) then g2o normalize (unit, w >= 0); q = (w, x, y, z)
__device__ __forceinline__ void mat_to_unit_quat(const double* R, double* q) {
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[0] = 0.5 * t; t = 0.5 / t;
        q[1] = (R[7] - R[5]) * t; q[2] = (R[2] - R[6]) * t; q[3] = (R[3] - R[1]) * t;
    } else if (R[0] >= R[4] && R[0] >= R[8]) {
        t = sqrt(R[0] - R[4] - R[8] + 1.0);
        q[1] = 0.5 * t; t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t; q[2] = (R[3] + R[1]) * t; q[3] = (R[6] + R[2]) * t;
    } else if (R[4] > R[0] && R[4] >= R[8]) {
        t = sqrt(R[4] - R[8] - R[0] + 1.0);
        q[2] = 0.5 * t; t = 0.5 / t;
        q[0] = (R[2] - R[6]) * t; q[3] = (R[7] + R[5]) * t; q[1] = (R[1] + R[3]) * t;
    } else {
        t = sqrt(R[8] - R[0] - R[4] + 1.0);
        q[3] = 0.5 * t; t = 0.5 / t;
        q[0] = (R[3] - R[1]) * t; q[1] = (R[2] + R[6]) * t; q[2] = (R[5] + R[7]) * t;
    }
    double n, inv;
    sqrt_and_rsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3], n, inv);
    inv = fast_rcp(n);
    if (q[0] < 0) inv = -inv;
    q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv;
}
__device__ __forceinline__ void quat_to_mat(double w, double x, double y, double z, double* R) {  // Eigen toRotationMatrix
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

// Residuals, weights, 6x6 normal equations and chi sums at pose X.
//   ranges: e = d - ||R o + t - a||;  J = [ -(R^T u)^T , (R^T u x 2o)^T ]           (analytic), or, JAC = 1,
//           g2o's central differences through X * fromVectorMQT(+-1e-9 e_d) (the reference's mode: the twelve perturbed
//           antenna points are shared by the tag's ranges)
//   prior : e = vec(q(Rm^T R)), J_rot = w I + [q]x, information pinfo (rotation rows only)
// The epoch's measurements are NOT held in registers across the passes: ep points at this lane's column of the epoch slot in
// LDS ([28][256] doubles: d0 w0 d1 w1 .. d7 w7, Rm[9], pinfo[3]) and every value is read where it is used.  The gate (first
// pass of an update only) writes the zeroed weights back.
constexpr int EP_STRIDE = 256;
template <int JAC>
__device__ __forceinline__ Sys6 evaluate6(const Pose& X, const double (&ax)[M8], const double (&ay)[M8], const double (&az)[M8],
                                          double* ep, const double ox, const double oy, const double oz, const bool gate_now, const double gate) {
    Sys6 s;
#pragma unroll
    for (int i = 0; i < 21; ++i) s.h[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) s.b[i] = 0.0;
    s.chi = 0.0;
    double prod = 1.0;
    int kexp = 0;
    const double* R = X.R;
    // antenna position p0 = R o + t (one lever arm for every range of the tag)
    const double p0x = R[0] * ox + R[1] * oy + R[2] * oz + X.t[0];
    const double p0y = R[3] * ox + R[4] * oy + R[5] * oz + X.t[1];
    const double p0z = R[6] * ox + R[7] * oy + R[8] * oz + X.t[2];
    const double o2x = 2.0 * ox, o2y = 2.0 * oy, o2z = 2.0 * oz;
    const bool near_ok = ox * ox + oy * oy + oz * oz <= 1.0;   // (wave-uniform: the lever arm is a launch parameter)
    double P0[3], Pp[6][3], Pm[6][3];
    if (JAC == 1) {
        constexpr double delta = 1e-9;
        const double off[3] = {ox, oy, oz};
        perturbed_point_plain<0>(R, X.t, off, 0.0, P0);  // dl = 0: X * identity, same operation order
        perturbed_point_plain<0>(R, X.t, off, delta, Pp[0]); perturbed_point_plain<0>(R, X.t, off, -delta, Pm[0]);
        perturbed_point_plain<1>(R, X.t, off, delta, Pp[1]); perturbed_point_plain<1>(R, X.t, off, -delta, Pm[1]);
        perturbed_point_plain<2>(R, X.t, off, delta, Pp[2]); perturbed_point_plain<2>(R, X.t, off, -delta, Pm[2]);
        perturbed_point_plain<3>(R, X.t, off, delta, Pp[3]); perturbed_point_plain<3>(R, X.t, off, -delta, Pm[3]);
        perturbed_point_plain<4>(R, X.t, off, delta, Pp[4]); perturbed_point_plain<4>(R, X.t, off, -delta, Pm[4]);
        perturbed_point_plain<5>(R, X.t, off, delta, Pp[5]); perturbed_point_plain<5>(R, X.t, off, -delta, Pm[5]);
    }
#pragma unroll
    for (int j = 0; j < M8; ++j) {
        const double dj = ep[(2 * j) * EP_STRIDE];
        double wj = ep[(2 * j + 1) * EP_STRIDE];
        if (gate_now) {  // |‖t - a‖ - d| > gate on the vertex origin, without a square root
            const double gx = X.t[0] - ax[j], gy = X.t[1] - ay[j], gz = X.t[2] - az[j];
            const double g2 = gx * gx + gy * gy + gz * gz;
            const double hi = dj + gate, lo = dj - gate;
            if (g2 > hi * hi || (lo > 0.0 && g2 < lo * lo)) { wj = 0.0; ep[(2 * j + 1) * EP_STRIDE] = 0.0; }
        }
        double e, J[6];
        if (JAC == 0) {
            const double ux = p0x - ax[j], uy = p0y - ay[j], uz = p0z - az[j];
            const double n2 = fmax(ux * ux + uy * uy + uz * uz, 1e-300);
            double n, inv;
            sqrt_and_rsqrt_fast(n2, n, inv);
            e = dj - n;
            const double vx = ux * inv, vy = uy * inv, vz = uz * inv;  // unit vector u
            // uR = R^T u
            const double rx = R[0] * vx + R[3] * vy + R[6] * vz;
            const double ry = R[1] * vx + R[4] * vy + R[7] * vz;
            const double rz = R[2] * vx + R[5] * vy + R[8] * vz;
            J[0] = -rx; J[1] = -ry; J[2] = -rz;
            J[3] = ry * o2z - rz * o2y; J[4] = rz * o2x - rx * o2z; J[5] = rx * o2y - ry * o2x;  // 2 (uR x o)
        } else {
            const double x0 = sq_to_plain(P0, ax[j], ay[j], az[j]);
            double h0, c0;
            double n0 = sqrt_ieee_unscaled_raw(x0, h0, c0);   // (+-0 / inf pass-through only in the slow branch: the fast one cannot see them)
            // the twelve perturbed norms from the central one (device_math.h: sqrt_ieee_near_c): the same correctly rounded numbers as long as the
            // perturbed antenna point stays within ~2e-9 m of the central one — translations move it by 1e-9, rotations by 2e-9 |lever arm|,
            // so lever arms of more than a metre take the full square roots, as in wave6_kernel.hip
            if (x0 >= 1e-5 && x0 < 1e300 && near_ok) {
#pragma unroll
                for (int dd = 0; dd < 6; ++dd)
                    J[dd] = central_difference_plain(dj, sqrt_ieee_near_c(sq_to_plain(Pp[dd], ax[j], ay[j], az[j]), n0, h0, c0),
                                                     sqrt_ieee_near_c(sq_to_plain(Pm[dd], ax[j], ay[j], az[j]), n0, h0, c0));
            } else {   // an antenna within millimetres of an anchor (or an estimate that has run away), or a long lever arm
                n0 = (x0 == 0.0 || x0 == __builtin_inf()) ? x0 : n0;
#pragma unroll
                for (int dd = 0; dd < 6; ++dd)
                    J[dd] = central_difference_plain(dj, norm_to_plain(Pp[dd], ax[j], ay[j], az[j]), norm_to_plain(Pm[dd], ax[j], ay[j], az[j]));
            }
            e = dj - n0;
        }
        const double we = wj * e;
        const double chi = e * we;
        const double aux = 1.0 + chi;
        const double wr = fast_rcp_1nr(aux) * wj;
        double wJ[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) wJ[r] = wr * J[r];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
#pragma unroll
            for (int c = r; c < 6; ++c) s.h[hidx(r, c)] = __builtin_fma(wJ[r], J[c], s.h[hidx(r, c)]);
            s.b[r] = __builtin_fma(-wJ[r], e, s.b[r]);
        }
        prod *= aux;
        if (j == 3) { kexp = __builtin_amdgcn_frexp_exp(prod); prod = __builtin_amdgcn_frexp_mant(prod); }
        s.chi += chi;
    }
    // ---- rotation prior ---------------------------------------------------------------------------------------
    double Rm[9], pinfo[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) Rm[i] = ep[(16 + i) * EP_STRIDE];
#pragma unroll
    for (int i = 0; i < 3; ++i) pinfo[i] = ep[(25 + i) * EP_STRIDE];
    double E[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) E[i * 3 + c] = Rm[0 * 3 + i] * R[0 * 3 + c] + Rm[1 * 3 + i] * R[1 * 3 + c] + Rm[2 * 3 + i] * R[2 * 3 + c];
    double q[4];
    mat_to_unit_quat(E, q);
    const double qw = q[0], qx = q[1], qy = q[2], qz = q[3];
    const double Jp[9] = {qw, -qz, qy, qz, qw, -qx, -qy, qx, qw};  // rows = error components, cols = rotation dofs
    const double er[3] = {qx, qy, qz};
    double pchi = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) pchi += er[i] * (pinfo[i] * er[i]);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = r; c < 3; ++c) {
            double acc = s.h[hidx(3 + r, 3 + c)];
#pragma unroll
            for (int i = 0; i < 3; ++i) acc = __builtin_fma(Jp[i * 3 + r] * pinfo[i], Jp[i * 3 + c], acc);
            s.h[hidx(3 + r, 3 + c)] = acc;
        }
        double bb = s.b[3 + r];
#pragma unroll
        for (int i = 0; i < 3; ++i) bb = __builtin_fma(-Jp[i * 3 + r] * pinfo[i], er[i], bb);
        s.b[3 + r] = bb;
    }
    s.chi += pchi;
    // Cauchy on the ranges, plain chi2 on the prior.  Sum_j log(1 + chi_j) is ONE log of the product, kept as (mantissa,
    // exponent) with the exponent taken out after four factors (snapshot_kernel.hip: evaluate): no overflow for float32 inputs.
    s.rchi = fast_log_ge1_scaled(prod, kexp) + pchi;
    return s;
}

// (H + lambda I) x = b, 6x6 LDL^T in registers; false if not positive definite
__device__ __forceinline__ bool solve6(const Sys6& s, const double lambda, double (&x)[6]) {
    double L[6][6];  // strictly lower part used
    double D[6], iD[6];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double dj = s.h[hidx(j, j)] + lambda;
#pragma unroll
        for (int k = 0; k < j; ++k) dj = __builtin_fma(-L[j][k] * D[k], L[j][k], dj);
        D[j] = dj;
        ok = ok && (dj > 0.0) && (dj < DBL_MAX);
        iD[j] = fast_rcp(dj);
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double v = s.h[hidx(j, i)];
#pragma unroll
            for (int k = 0; k < j; ++k) v = __builtin_fma(-L[i][k] * D[k], L[j][k], v);
            L[i][j] = v * iD[j];
        }
    }
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double v = s.b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) v = __builtin_fma(-L[i][k], y[k], v);
        y[i] = v;
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double v = y[i] * iD[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) v = __builtin_fma(-L[k][i], x[k], v);
        x[i] = v;
    }
    return ok;
}

// VertexSE3::oplusImpl: X <- X * fromVectorMQT(dx)                                             
__device__ __forceinline__ Pose oplus(const Pose& X, const double (&dx)[6]) {
    Pose o;
    const double ww = 1.0 - (dx[3] * dx[3] + dx[4] * dx[4] + dx[5] * dx[5]);
    double Rd[9];
    if (ww < 0) { Rd[0] = 1; Rd[1] = 0; Rd[2] = 0; Rd[3] = 0; Rd[4] = 1; Rd[5] = 0; Rd[6] = 0; Rd[7] = 0; Rd[8] = 1; }
    else quat_to_mat(sqrt(ww), dx[3], dx[4], dx[5], Rd);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int c = 0; c < 3; ++c) o.R[i * 3 + c] = X.R[i * 3 + 0] * Rd[0 * 3 + c] + X.R[i * 3 + 1] * Rd[1 * 3 + c] + X.R[i * 3 + 2] * Rd[2 * 3 + c];
        o.t[i] = X.R[i * 3 + 0] * dx[0] + X.R[i * 3 + 1] * dx[1] + X.R[i * 3 + 2] * dx[2] + X.t[i];
    }
    return o;
}

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

// LDS per block (256 lanes): TWO converted epochs (epoch e lives in slot e & 1; a wave's lanes are at epoch c or c + 1):
// d,w (8 pairs) + Rm (9) + pinfo (3) = 28 doubles per lane and slot; the accepted pose X (12 doubles per lane);
// raw epoch in flight: 4 float4 range tiles + 4 x 16 B of IMU = 8 x 16 B per lane.
constexpr int NEXT_DOUBLES = 28;
constexpr int RAW_PIECES = 8;
constexpr int FUSION_LDS_BYTES = 2 * NEXT_DOUBLES * 256 * 8 + RAW_PIECES * 256 * 16;

template <int JAC>
__global__ void __launch_bounds__(256) fusion_lm_kernel(const FusionArgs a) {
    __shared__ __attribute__((aligned(16))) char lds_bytes[FUSION_LDS_BYTES];
    double* const s_ep = reinterpret_cast<double*>(lds_bytes);                            // [2][28][256]
    float4* const s_raw = reinterpret_cast<float4*>(lds_bytes + 2 * NEXT_DOUBLES * 256 * 8);  // [8][256]
    const int tib = threadIdx.x;
    const long long inst = (long long)blockIdx.x * blockDim.x + tib;
    const long long B = a.B;
    const int K = a.K;
    const bool live = (inst < B) && K > 0;
    const long long ld_inst = live ? inst : 0;
    bool exhausted = !live;

    double ax[M8], ay[M8], az[M8];
#pragma unroll
    for (int j = 0; j < M8; ++j) { ax[j] = a.anchors[j * 3 + 0]; ay[j] = a.anchors[j * 3 + 1]; az[j] = a.anchors[j * 3 + 2]; }
    const double ox = a.offset[0], oy = a.offset[1], oz = a.offset[2];

    constexpr double tau = 1e-5, good_lo = 1.0 / 3.0, good_hi = 2.0 / 3.0;
    constexpr int max_trials = 10;

    // ---- epoch data movement (see snapshot_kernel.hip for the window scheme) -----------------------------------------
    auto convert = [&](const float (&df)[M8], const float (&sf)[M8], const double (&im)[8], double (&dd)[M8], double (&ww)[M8],
                       double (&Rm)[9], double (&pi)[3]) {
#pragma unroll
        for (int j = 0; j < M8; ++j) {
            const double x = (double)df[j], ss = (double)sf[j];
            const bool valid = (ss > 0.0) && (ss < DBL_MAX) && (fabs(x) < DBL_MAX);
            const double cov = valid ? ss * ss : 1.0;
            dd[j] = valid ? x : 0.0;
            ww[j] = fast_rcp(cov) * (valid ? 1.0 : 0.0);
        }
        quat_to_mat(im[3], im[0], im[1], im[2], Rm);  // Identity.rotate(Quaterniond(w, x, y, z)), no normalisation
        pi[0] = fast_rcp(im[4]); pi[1] = fast_rcp(im[5]); pi[2] = fast_rcp(im[6]);
    };
    auto load_direct = [&](int ke, float (&df)[M8], float (&sf)[M8], double (&im)[8]) {
#pragma unroll
        for (int q4 = 0; q4 < 2; ++q4) {
            const long long idx = ((long long)ke * 2 + q4) * B + ld_inst;
            const float4 dv = reinterpret_cast<const float4*>(a.dist)[idx];
            const float4 sv = reinterpret_cast<const float4*>(a.err)[idx];
            df[4 * q4 + 0] = dv.x; df[4 * q4 + 1] = dv.y; df[4 * q4 + 2] = dv.z; df[4 * q4 + 3] = dv.w;
            sf[4 * q4 + 0] = sv.x; sf[4 * q4 + 1] = sv.y; sf[4 * q4 + 2] = sv.z; sf[4 * q4 + 3] = sv.w;
        }
        const double* ip = a.imu + ((long long)ke * B + ld_inst) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) im[i] = ip[i];
    };
    auto put_epoch = [&](int ke, const double (&dd)[M8], const double (&ww)[M8], const double (&Rm)[9], const double (&pi)[3]) {
        double* sl = s_ep + (ke & 1) * NEXT_DOUBLES * 256 + tib;
#pragma unroll
        for (int j = 0; j < M8; ++j) { sl[(2 * j) * 256] = dd[j]; sl[(2 * j + 1) * 256] = ww[j]; }
#pragma unroll
        for (int i = 0; i < 9; ++i) sl[(16 + i) * 256] = Rm[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) sl[(25 + i) * 256] = pi[i];
    };
    auto prefetch = [&](int ke) {
        const int wave_base = (tib / 64) * 64;
#pragma unroll
        for (int q4 = 0; q4 < 2; ++q4) {
            const long long idx = ((long long)ke * 2 + q4) * B + ld_inst;
            __builtin_amdgcn_global_load_lds((glb_void_ptr)(reinterpret_cast<const float4*>(a.dist) + idx), (lds_void_ptr)(s_raw + q4 * 256 + wave_base), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void_ptr)(reinterpret_cast<const float4*>(a.err) + idx), (lds_void_ptr)(s_raw + (2 + q4) * 256 + wave_base), 16, 0, 0);
        }
        const float4* ip = reinterpret_cast<const float4*>(a.imu + ((long long)ke * B + ld_inst) * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((glb_void_ptr)(ip + i), (lds_void_ptr)(s_raw + (4 + i) * 256 + wave_base), 16, 0, 0);
    };
    auto consume = [&](double (&dd)[M8], double (&ww)[M8], double (&Rm)[9], double (&pi)[3]) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float df[M8], sf[M8];
        double im[8];
#pragma unroll
        for (int q4 = 0; q4 < 2; ++q4) {
            const float4 dv = s_raw[q4 * 256 + tib];
            const float4 sv = s_raw[(2 + q4) * 256 + tib];
            df[4 * q4 + 0] = dv.x; df[4 * q4 + 1] = dv.y; df[4 * q4 + 2] = dv.z; df[4 * q4 + 3] = dv.w;
            sf[4 * q4 + 0] = sv.x; sf[4 * q4 + 1] = sv.y; sf[4 * q4 + 2] = sv.z; sf[4 * q4 + 3] = sv.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double2 v = reinterpret_cast<const double2*>(s_raw)[(4 + i) * 256 + tib];
            im[2 * i] = v.x; im[2 * i + 1] = v.y;
        }
        convert(df, sf, im, dd, ww, Rm, pi);
    };

    Pose X;
#pragma unroll
    for (int i = 0; i < 9; ++i) X.R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    X.t[0] = X.t[1] = X.t[2] = 0.0;
    if (live) {
        X.t[0] = a.pose[0 * B + inst]; X.t[1] = a.pose[1 * B + inst]; X.t[2] = a.pose[2 * B + inst];
        quat_to_mat(a.pose[6 * B + inst], a.pose[3 * B + inst], a.pose[4 * B + inst], a.pose[5 * B + inst], X.R);
    }
    {
        float df[M8], sf[M8];
        double im[8];
        double dn[M8], wn[M8], Rn[9], pn[3];
        load_direct(0, df, sf, im);
        convert(df, sf, im, dn, wn, Rn, pn);
        put_epoch(0, dn, wn, Rn, pn);
        if (K > 1) {
            load_direct(1, df, sf, im);
            convert(df, sf, im, dn, wn, Rn, pn);
            put_epoch(1, dn, wn, Rn, pn);
        }
        if (K > 2) prefetch(2);
    }
    asm volatile("" : "+v"(X.t[0]), "+v"(X.t[1]), "+v"(X.t[2]));

    Sys6 cur;
#pragma unroll
    for (int i = 0; i < 21; ++i) cur.h[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) { cur.h[hidx(i, i)] = 1.0; cur.b[i] = 0.0; }
    cur.rchi = cur.chi = 0.0;
    double cur_chi = 0, last_chi = 0, lambda = 1.0, ni = 2.0;
    int c = 0, k = 0, it = 0, q = 0, trials = 0;
    bool init = true, waiting = false;

    while (__any(!exhausted)) {
        const bool active = !exhausted && !waiting;
        double* const ep = s_ep + (k & 1) * NEXT_DOUBLES * 256 + tib;  // this lane's epoch slot
        // a fresh epoch starts from the IMU's rotation (localization.cpp:505-513); the first pass evaluates that pose
        if (active && init) {
#pragma unroll
            for (int i = 0; i < 9; ++i) X.R[i] = ep[(16 + i) * EP_STRIDE];
        }
        double x[6];
        const bool ok2 = solve6(cur, lambda, x) && !init;
        if (!ok2) {
#pragma unroll
            for (int i = 0; i < 6; ++i) x[i] = 0.0;
        }
        const Pose T = oplus(X, x);
        const bool gate_now = active && init && (a.gate > 0.0) && (k >= a.gate_from_epoch);
        // computeScale needs the step and the accepted b only: formed here, so x is dead across the edge loop
        double scale = 1e-3;
#pragma unroll
        for (int i = 0; i < 6; ++i) scale = __builtin_fma(x[i], __builtin_fma(lambda, x[i], cur.b[i]), scale);
        const Sys6 tr = evaluate6<JAC>(T, ax, ay, az, ep, ox, oy, oz, gate_now, a.gate);

        bool finished = false;
        if (active) {
            if (init) {
                init = false;
                cur = tr;
                cur_chi = tr.rchi;
                last_chi = tr.chi;
                double md = 0.0, tr6 = 0.0;
#pragma unroll
                for (int i = 0; i < 6; ++i) { md = fmax(md, fabs(tr.h[hidx(i, i)])); tr6 += tr.h[hidx(i, i)]; }
                lambda = tau * md;  // computeLambdaInit
                ni = 2.0;
                it = 0; q = 0; trials = 0;
                finished = (a.iterations <= 0) || !(tr6 > 0.0);
            } else {
                const double temp_chi = ok2 ? tr.rchi : DBL_MAX;
                const double rho = (cur_chi - temp_chi) * fast_rcp(scale);
                const bool accept = (rho > 0.0) && (fabs(temp_chi) < DBL_MAX) && ok2;
                ++trials;
                last_chi = tr.chi;
                if (accept) {
                    const double r21 = 2.0 * rho - 1.0;
                    double alpha = 1.0 - r21 * r21 * r21;
                    alpha = fmin(alpha, good_hi);
                    lambda *= fmax(good_lo, alpha);
                    ni = 2.0;
                    cur_chi = temp_chi;
                    X = T;
                    cur = tr;
                } else {
                    lambda *= ni;
                    ni *= 2.0;
                }
                ++q;
                const bool again = (rho < 0.0) && (q < max_trials);
                if (!again) {
                    ++it;
                    finished = (q == max_trials) || (rho == 0.0) || (it >= a.iterations);
                    q = 0;
                }
            }
        }

        if (__any(finished)) {
            if (finished) {
                double qq[4];
                mat_to_unit_quat(X.R, qq);  // tf::poseEigenToMsg: Quaterniond(R), w >= 0
                double* op = a.out_pose + (long long)k * 7 * B + inst;
                op[0 * B] = X.t[0]; op[1 * B] = X.t[1]; op[2 * B] = X.t[2];
                op[3 * B] = qq[1]; op[4 * B] = qq[2]; op[5 * B] = qq[3]; op[6 * B] = qq[0];
                a.out_chi2[(long long)k * B + inst] = last_chi;
                if (a.out_trials) a.out_trials[(long long)k * B + inst] = (uint8_t)(trials > 255 ? 255 : trials);
                if (k + 1 >= K) {
                    exhausted = true;
                    a.pose[0 * B + inst] = X.t[0]; a.pose[1 * B + inst] = X.t[1]; a.pose[2 * B + inst] = X.t[2];
                    a.pose[3 * B + inst] = qq[1]; a.pose[4 * B + inst] = qq[2]; a.pose[5 * B + inst] = qq[3]; a.pose[6 * B + inst] = qq[0];
                } else if (k == c) {
                    k = c + 1;   // the next epoch is already converted, in slot (c + 1) & 1
                    init = true;
                } else {
                    waiting = true;
                }
            }
        }
        if (!__any(!exhausted && k == c)) {
            ++c;
            if (c + 1 < K) {
                double dn[M8], wn[M8], Rn[9], pn[3];
                consume(dn, wn, Rn, pn);
                put_epoch(c + 1, dn, wn, Rn, pn);   // slot (c + 1) & 1 held epoch c - 1: no lane is there any more
                asm volatile("" ::: "memory");
                if (c + 2 < K) prefetch(c + 2);
                if (waiting) {
                    k = c + 1;
                    init = true;
                    waiting = false;
                }
            }
        }
    }
}

}  // namespace

hipError_t launch_fusion(const FusionArgs& a, int block_threads, hipStream_t stream) {
    if (a.B <= 0 || a.K <= 0) return hipErrorInvalidValue;
    if (block_threads <= 0) block_threads = 256;
    if (block_threads % 64 || block_threads > 256) return hipErrorInvalidValue;
    const long long blocks = (a.B + block_threads - 1) / block_threads;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    if (a.jacobian) hipLaunchKernelGGL(fusion_lm_kernel<1>, dim3((unsigned)blocks), dim3((unsigned)block_threads), 0, stream, a);
    else hipLaunchKernelGGL(fusion_lm_kernel<0>, dim3((unsigned)blocks), dim3((unsigned)block_threads), 0, stream, a);
    return hipGetLastError();
}

}  // namespace locamd
