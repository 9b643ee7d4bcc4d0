// One EdgeSE3 factor (addPoseEdge / addTwistEdge: localization.cpp:254-290, 438-459, 560-605) evaluated with its record in registers: shared by
// chain_lm_kernel (chain_kernel.hip) and the forest kernels (tree_kernel.hip).  Internal to the including translation unit.
#pragma once
#include "window_device.h"

namespace locamd {
namespace {

// One EdgeSE3 between the consecutive poses i and j of a chain window: the math of evaluate_edges' SE3 branch, with the record in
// registers.  Returns chi; rterm = the edge's robust cost; FULL: H_ii, H_jj (lower triangles, 21), the off-diagonal block with the
// rows of the LATER pose (36, column-major), b_i, b_j.
// VS: stride of the record's entries in doubles (1: a plain array; 64: one lane's column of an [entry][lane] LDS block — the inverse
// measurement is copied to registers, the information matrix is read where it is used).
template <bool FULL, int VS = 1>
__device__ __forceinline__ double chain_se3_terms(const double* Xi, const double* Xj, const double* val_, bool robust, bool j_is_later,
                                                  double* Hii, double* Hjj, double* Hoff, double* bi_, double* bj_, double& rterm) {
    double val[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) val[k] = val_[VS * k];
    double RB[9], tB[3], dt[3] = {Xj[9] - Xi[9], Xj[10] - Xi[10], Xj[11] - Xi[11]};
    mat_tmul(Xi, Xj, RB);
    mat_tvec(Xi, dt, tB);
    double RE[9], tE[3];
    mat_mul(val, RB, RE);
    mat_vec(val, tB, tE);
    tE[0] += val[9]; tE[1] += val[10]; tE[2] += val[11];
    double qE[4];
    mat_to_quat(RE, qE);
    quat_normalize_sign(qE);
    const double err[6] = {tE[0], tE[1], tE[2], qE[1], qE[2], qE[3]};
    const double* Om_ = val_ + VS * 12;
#define Om(k) Om_[VS * (k)]
    double Oe[6];
    double chi = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double r = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) r += Om(i * 6 + j) * err[j];
        Oe[i] = r;
        chi += err[i] * r;
    }
    const double aux = 1.0 + chi;
    rterm = robust ? fast_log_ge1(aux) : chi;
    if (FULL) {
        const double w = robust ? 1.0 / aux : 1.0;
        double J0[36], J1[36];
#pragma unroll
        for (int i = 0; i < 36; ++i) { J0[i] = 0.0; J1[i] = 0.0; }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) J1[i * 6 + j] = RE[i * 3 + j];
        quat_right_jac(qE, 1.0, J1, 6);
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
        const double sg = qAB[0] < 0 ? -1.0 : 1.0;
        const double nrm = 1.0 / sqrt(qAB[0] * qAB[0] + qAB[1] * qAB[1] + qAB[2] * qAB[2] + qAB[3] * qAB[3]);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double ek[4] = {0, 0, 0, 0}, r1[4], r2[4];
            ek[1 + k] = 1.0;
            quat_mul(qA, ek, r1);
            quat_mul(r1, qB, r2);
#pragma unroll
            for (int i = 0; i < 3; ++i) J0[(3 + i) * 6 + 3 + k] = -sg * nrm * r2[1 + i];
        }
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
                for (int j = 0; j < LOCAMD_J0_HI(cc); ++j) s0 += Om(i * 6 + j) * J0[j * 6 + cc];
                WJ[i * 6 + cc] = w * s0;
            }
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int cc = 0; cc <= r; ++cc) {
                double h = 0.0;
#pragma unroll
                for (int i = 0; i < LOCAMD_J0_HI(r); ++i) h += J0[i * 6 + r] * WJ[i * 6 + cc];
                Hii[r * (r + 1) / 2 + cc] = h;
            }
        if (j_is_later) {
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int cc = 0; cc < 6; ++cc) {
                    double h = 0.0;
#pragma unroll
                    for (int i = LOCAMD_J1_LO(r); i < LOCAMD_J1_HI(r); ++i) h += J1[i * 6 + r] * WJ[i * 6 + cc];
                    Hoff[6 * cc + r] = h;
                }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int cc = 0; cc < 6; ++cc) {
                double s1 = 0.0;
#pragma unroll
                for (int j = LOCAMD_J1_LO(cc); j < LOCAMD_J1_HI(cc); ++j) s1 += Om(i * 6 + j) * J1[j * 6 + cc];
                WJ[i * 6 + cc] = w * s1;
            }
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int cc = 0; cc <= r; ++cc) {
                double h = 0.0;
#pragma unroll
                for (int i = LOCAMD_J1_LO(r); i < LOCAMD_J1_HI(r); ++i) h += J1[i * 6 + r] * WJ[i * 6 + cc];
                Hjj[r * (r + 1) / 2 + cc] = h;
            }
        if (!j_is_later) {
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int cc = 0; cc < 6; ++cc) {
                    double h = 0.0;
#pragma unroll
                    for (int i = 0; i < LOCAMD_J0_HI(r); ++i) h += J0[i * 6 + r] * WJ[i * 6 + cc];
                    Hoff[6 * cc + r] = h;
                }
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            double bi = 0.0, bj = 0.0;
#pragma unroll
            for (int i = 0; i < LOCAMD_J0_HI(r); ++i) bi += J0[i * 6 + r] * (-w * Oe[i]);
#pragma unroll
            for (int i = LOCAMD_J1_LO(r); i < LOCAMD_J1_HI(r); ++i) bj += J1[i * 6 + r] * (-w * Oe[i]);
            bi_[r] = bi;
            bj_[r] = bj;
        }
#undef LOCAMD_J1_LO
#undef LOCAMD_J1_HI
#undef LOCAMD_J0_HI
    }
#undef Om
    return chi;
}

}  // namespace
}  // namespace locamd
