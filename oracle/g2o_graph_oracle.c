/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see g2o_graph_oracle.h for the scope and the "parity unpinned" note).
 *
 * CPU restatement, in plain C / IEEE double, of the arithmetic the reference delegates to g2o:
 * Localization::solve() = initializeOptimization() + optimize(iteration_max)
 * (/root/reference/src/localization/localization.cpp:164-170) over the graph that
 * Robot::init/new_vertex (robot.cpp:31-58,75-110) and Localization::add*Edge (localization.cpp:254-535)
 * build.  g2o semantics follow SURVEY.md Appendix A (A.1 .. A.9); each block below names the item.
 */
#include "g2o_graph_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * small fixed-size algebra (Eigen::Isometry3d stand-in: R row-major, t)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { double R[9]; double t[3]; } iso3;

static void iso_identity(iso3* a) {
    memset(a, 0, sizeof(*a));
    a->R[0] = a->R[4] = a->R[8] = 1.0;
}
static void mat3_mul(const double* A, const double* B, double* C) {
    double o[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            o[i * 3 + j] = A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j] + A[i * 3 + 2] * B[2 * 3 + j];
    memcpy(C, o, sizeof(o));
}
static void mat3_vec(const double* A, const double* v, double* o) {
    double r[3];
    for (int i = 0; i < 3; ++i) r[i] = A[i * 3 + 0] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2];
    o[0] = r[0]; o[1] = r[1]; o[2] = r[2];
}
/* (Ra,ta)*(Rb,tb) = (Ra Rb, Ra tb + ta) */
static void iso_mul(const iso3* a, const iso3* b, iso3* c) {
    iso3 o;
    mat3_mul(a->R, b->R, o.R);
    mat3_vec(a->R, b->t, o.t);
    o.t[0] += a->t[0]; o.t[1] += a->t[1]; o.t[2] += a->t[2];
    *c = o;
}
/* Isometry inverse: (R^T, -R^T t) */
static void iso_inv(const iso3* a, iso3* c) {
    iso3 o;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) o.R[i * 3 + j] = a->R[j * 3 + i];
    mat3_vec(o.R, a->t, o.t);
    o.t[0] = -o.t[0]; o.t[1] = -o.t[1]; o.t[2] = -o.t[2];
    *c = o;
}

/* Eigen::Quaternion::toRotationMatrix (no normalisation; addImuEdge relies on it, localization.cpp:509) */
void og_quat_to_R(const double* q, double* R) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
/* Eigen::Quaternion(Matrix3) */
void og_R_to_quat(const double* R, double* q) {
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[1] = (R[7] - R[5]) * t;
        q[2] = (R[2] - R[6]) * t;
        q[3] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[i * 3 + i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
        q[1 + i] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[k * 3 + j] - R[j * 3 + k]) * t;
        q[1 + j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
        q[1 + k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
    }
}
static void quat_mul(const double* a, const double* b, double* o) {
    double r[4];
    r[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    r[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    r[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    r[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    memcpy(o, r, sizeof(r));
}
/* g2o internal::normalize(q): unit length, w >= 0 */
static double quat_normalize_sign(double* q) {
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    double s = 1.0;
    for (int i = 0; i < 4; ++i) q[i] /= n;
    if (q[0] < 0) { s = -1.0; for (int i = 0; i < 4; ++i) q[i] = -q[i]; }
    return s;
}

/* A.1: g2o internal::fromVectorMQT — (t, compact quaternion) -> isometry */
void og_from_vector_mqt(const double* v, double* R, double* t) {
    double w = 1.0 - (v[3] * v[3] + v[4] * v[4] + v[5] * v[5]);
    if (w < 0) {
        R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
    } else {
        double q[4] = { sqrt(w), v[3], v[4], v[5] };
        og_quat_to_R(q, R);
    }
    t[0] = v[0]; t[1] = v[1]; t[2] = v[2];
}
/* A.9: g2o internal::toVectorMQT */
void og_to_vector_mqt(const double* R, const double* t, double* v) {
    double q[4];
    og_R_to_quat(R, q);
    quat_normalize_sign(q);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2];
    v[3] = q[1]; v[4] = q[2]; v[5] = q[3];
}

/* A.4: RobustKernelCauchy::robustify with delta = 1 (localization.cpp:624 never changes it) */
double og_cauchy_rho(double e2, double* rho1) {
    const double dsqr = 1.0, dsqrReci = 1.0 / dsqr;
    double aux = dsqrReci * e2 + 1.0;
    if (rho1) *rho1 = 1.0 / aux;
    return dsqr * log(aux);
}

/* ------------------------------------------------------------------------------------------------
 * graph containers
 * ---------------------------------------------------------------------------------------------- */
enum { EK_RANGE = 1, EK_PRIOR = 2, EK_SE3 = 3 };

typedef struct {
    int alive, id, fixed;
    iso3 est, backup;
    int active; /* set by initializeOptimization */
    int hidx;   /* Hessian block index, -1 if fixed/inactive */
} vertex_t;

typedef struct {
    int alive, kind, nv, dim, robust;
    int v[2]; /* vertex slots */
    double meas;     /* range */
    iso3 Z, Zinv;    /* se3 / prior measurement */
    double off[2][3];
    double info[36]; /* dim x dim row-major */
    double err[6];   /* last computed error (what chi2() reads) */
    double J[2][36]; /* dim x 6 row-major per endpoint */
    int active;
} edge_t;

struct og_graph {
    vertex_t* V; int nV, capV;
    edge_t* E; int nE, capE;
    int* act_e; int n_act_e;
    int* idx_v; int n_idx; /* non-fixed active vertices in Hessian order */
    double *H, *Hs, *b, *x;
    int hcap;
    double lambda, ni;
};

og_graph* og_create(void) { return (og_graph*)calloc(1, sizeof(og_graph)); }
void og_destroy(og_graph* g) {
    if (!g) return;
    free(g->V); free(g->E); free(g->act_e); free(g->idx_v);
    free(g->H); free(g->Hs); free(g->b); free(g->x);
    free(g);
}
static int find_vertex(og_graph* g, int id) {
    for (int i = 0; i < g->nV; ++i)
        if (g->V[i].alive && g->V[i].id == id) return i;
    return -1;
}
int og_has_vertex(og_graph* g, int id) { return find_vertex(g, id) >= 0; }
int og_num_vertices(og_graph* g) { int n = 0; for (int i = 0; i < g->nV; ++i) n += g->V[i].alive; return n; }
int og_num_edges(og_graph* g) { int n = 0; for (int i = 0; i < g->nE; ++i) n += g->E[i].alive; return n; }

int og_add_vertex(og_graph* g, int id, const double* R, const double* t, int fixed) {
    if (find_vertex(g, id) >= 0) return -1; /* g2o addVertex refuses duplicate ids */
    int slot = -1;
    for (int i = 0; i < g->nV; ++i) if (!g->V[i].alive) { slot = i; break; }
    if (slot < 0) {
        if (g->nV == g->capV) {
            g->capV = g->capV ? 2 * g->capV : 64;
            g->V = (vertex_t*)realloc(g->V, sizeof(vertex_t) * (size_t)g->capV);
        }
        slot = g->nV++;
    }
    vertex_t* v = &g->V[slot];
    memset(v, 0, sizeof(*v));
    v->alive = 1; v->id = id; v->fixed = fixed; v->hidx = -1;
    memcpy(v->est.R, R, sizeof(double) * 9);
    memcpy(v->est.t, t, sizeof(double) * 3);
    return 0;
}
/* robot.cpp:96 optimizer.removeVertex(v,false): the vertex and every edge touching it disappear */
int og_remove_vertex(og_graph* g, int id) {
    int s = find_vertex(g, id);
    if (s < 0) return -1;
    for (int i = 0; i < g->nE; ++i) {
        edge_t* e = &g->E[i];
        if (!e->alive) continue;
        for (int k = 0; k < e->nv; ++k) if (e->v[k] == s) { e->alive = 0; break; }
    }
    g->V[s].alive = 0;
    return 0;
}
int og_set_estimate(og_graph* g, int id, const double* R, const double* t) {
    int s = find_vertex(g, id);
    if (s < 0) return -1;
    memcpy(g->V[s].est.R, R, sizeof(double) * 9);
    memcpy(g->V[s].est.t, t, sizeof(double) * 3);
    return 0;
}
int og_get_estimate(og_graph* g, int id, double* R, double* t) {
    int s = find_vertex(g, id);
    if (s < 0) return -1;
    if (R) memcpy(R, g->V[s].est.R, sizeof(double) * 9);
    if (t) memcpy(t, g->V[s].est.t, sizeof(double) * 3);
    return 0;
}
static edge_t* new_edge(og_graph* g) {
    /* edges keep creation order (g2o sorts active edges by internal id = creation order) */
    if (g->nE == g->capE) {
        /* compact dead edges first so long streams do not grow without bound */
        int w = 0;
        for (int i = 0; i < g->nE; ++i) if (g->E[i].alive) { if (w != i) g->E[w] = g->E[i]; ++w; }
        g->nE = w;
        if (g->nE * 2 >= g->capE) {
            g->capE = g->capE ? 2 * g->capE : 128;
            g->E = (edge_t*)realloc(g->E, sizeof(edge_t) * (size_t)g->capE);
        }
    }
    edge_t* e = &g->E[g->nE++];
    memset(e, 0, sizeof(*e));
    e->alive = 1;
    return e;
}
/* localization.cpp:608-627 create_range_edge + types_edge_se3range.cpp:99-103 setVertexOffset */
int og_add_range_edge(og_graph* g, int id0, int id1, double meas, double info,
                      const double* off0, const double* off1, int robust) {
    int s0 = find_vertex(g, id0), s1 = find_vertex(g, id1);
    if (s0 < 0 || s1 < 0) return -1;
    edge_t* e = new_edge(g);
    e->kind = EK_RANGE; e->nv = 2; e->dim = 1; e->robust = robust;
    e->v[0] = s0; e->v[1] = s1;
    e->meas = meas; e->info[0] = info;
    if (off0) memcpy(e->off[0], off0, sizeof(double) * 3);
    if (off1) memcpy(e->off[1], off1, sizeof(double) * 3);
    return 0;
}
/* localization.cpp:481-486,520-525 EdgeSE3Prior with parameter id 0 = identity offset (:54-56) */
int og_add_prior_edge(og_graph* g, int id, const double* Rm, const double* tm, const double* info36) {
    int s = find_vertex(g, id);
    if (s < 0) return -1;
    edge_t* e = new_edge(g);
    e->kind = EK_PRIOR; e->nv = 1; e->dim = 6; e->robust = 0;
    e->v[0] = s; e->v[1] = -1;
    memcpy(e->Z.R, Rm, sizeof(double) * 9);
    memcpy(e->Z.t, tm, sizeof(double) * 3);
    iso_inv(&e->Z, &e->Zinv);
    memcpy(e->info, info36, sizeof(double) * 36);
    return 0;
}
/* localization.cpp:263-281 (pose), :588-602 (twist) EdgeSE3 */
int og_add_se3_edge(og_graph* g, int id0, int id1, const double* Rm, const double* tm,
                    const double* info36, int robust) {
    int s0 = find_vertex(g, id0), s1 = find_vertex(g, id1);
    if (s0 < 0 || s1 < 0) return -1;
    edge_t* e = new_edge(g);
    e->kind = EK_SE3; e->nv = 2; e->dim = 6; e->robust = robust;
    e->v[0] = s0; e->v[1] = s1;
    memcpy(e->Z.R, Rm, sizeof(double) * 9);
    memcpy(e->Z.t, tm, sizeof(double) * 3);
    iso_inv(&e->Z, &e->Zinv);
    memcpy(e->info, info36, sizeof(double) * 36);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * errors (computeError) and Jacobians (linearizeOplus)
 * ---------------------------------------------------------------------------------------------- */
/* A.1 VertexSE3::oplusImpl: X <- X * fromVectorMQT(d)  (right multiplication; t += R dt) */
static void vertex_oplus(const iso3* x, const double* d6, iso3* out) {
    iso3 inc;
    og_from_vector_mqt(d6, inc.R, inc.t);
    iso_mul(x, &inc, out);
}
/* types_edge_se3range.cpp:105-114 */
static double range_error(const edge_t* e, const iso3* x0, const iso3* x1) {
    double p0[3], p1[3];
    mat3_vec(x0->R, e->off[0], p0);
    mat3_vec(x1->R, e->off[1], p1);
    double dx = (p0[0] + x0->t[0]) - (p1[0] + x1->t[0]);
    double dy = (p0[1] + x0->t[1]) - (p1[1] + x1->t[1]);
    double dz = (p0[2] + x0->t[2]) - (p1[2] + x1->t[2]);
    return e->meas - sqrt(dx * dx + dy * dy + dz * dz);
}
static void se3_error_iso(const edge_t* e, const og_graph* g, iso3* Eout) {
    if (e->kind == EK_PRIOR) {
        iso_mul(&e->Zinv, &g->V[e->v[0]].est, Eout); /* Z^-1 * X * P, P = identity */
    } else {
        iso3 xi_inv, tmp;
        iso_inv(&g->V[e->v[0]].est, &xi_inv);
        iso_mul(&e->Zinv, &xi_inv, &tmp);
        iso_mul(&tmp, &g->V[e->v[1]].est, Eout); /* Z^-1 * Xi^-1 * Xj */
    }
}
static void edge_compute_error(og_graph* g, edge_t* e) {
    if (e->kind == EK_RANGE) {
        e->err[0] = range_error(e, &g->V[e->v[0]].est, &g->V[e->v[1]].est);
    } else {
        iso3 E;
        se3_error_iso(e, g, &E);
        og_to_vector_mqt(E.R, E.t, e->err);
    }
}
static double edge_chi2(const edge_t* e) {
    double s = 0;
    for (int i = 0; i < e->dim; ++i) {
        double r = 0;
        for (int j = 0; j < e->dim; ++j) r += e->info[i * e->dim + j] * e->err[j];
        s += e->err[i] * r;
    }
    return s;
}

/* A.3: BaseBinaryEdge::linearizeOplus numeric central difference, delta = 1e-9, per non-fixed endpoint */
static void range_jacobian_numeric(og_graph* g, edge_t* e) {
    const double delta = 1e-9;
    const double scalar = 1.0 / (2 * delta);
    for (int k = 0; k < 2; ++k) {
        const vertex_t* vk = &g->V[e->v[k]];
        memset(e->J[k], 0, sizeof(double) * 36);
        if (vk->fixed) continue;
        double add[6] = { 0, 0, 0, 0, 0, 0 };
        for (int d = 0; d < 6; ++d) {
            iso3 xp, xm;
            add[d] = delta;
            vertex_oplus(&vk->est, add, &xp);
            add[d] = -delta;
            vertex_oplus(&vk->est, add, &xm);
            add[d] = 0.0;
            double ep = (k == 0) ? range_error(e, &xp, &g->V[e->v[1]].est) : range_error(e, &g->V[e->v[0]].est, &xp);
            double em = (k == 0) ? range_error(e, &xm, &g->V[e->v[1]].est) : range_error(e, &g->V[e->v[0]].est, &xm);
            double bak = ep;
            bak -= em;
            e->J[k][d] = scalar * bak;
        }
    }
}
/* exact derivative of the same error; 0 where the endpoints coincide (= what A.3 evaluates to there) */
static void range_jacobian_analytic(og_graph* g, edge_t* e) {
    const iso3* x0 = &g->V[e->v[0]].est;
    const iso3* x1 = &g->V[e->v[1]].est;
    double p0[3], p1[3], u[3];
    mat3_vec(x0->R, e->off[0], p0);
    mat3_vec(x1->R, e->off[1], p1);
    for (int i = 0; i < 3; ++i) u[i] = (p0[i] + x0->t[i]) - (p1[i] + x1->t[i]);
    double n = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    memset(e->J[0], 0, sizeof(double) * 36);
    memset(e->J[1], 0, sizeof(double) * 36);
    if (n == 0.0) return;
    for (int i = 0; i < 3; ++i) u[i] /= n;
    for (int k = 0; k < 2; ++k) {
        const vertex_t* vk = &g->V[e->v[k]];
        if (vk->fixed) continue;
        const double* R = vk->est.R;
        const double* o = e->off[k];
        double sgn = (k == 0) ? -1.0 : 1.0; /* A.2: de/dp0 = -u^T, de/dp1 = +u^T */
        double uR[3]; /* u^T R */
        for (int j = 0; j < 3; ++j) uR[j] = u[0] * R[0 * 3 + j] + u[1] * R[1 * 3 + j] + u[2] * R[2 * 3 + j];
        double c[3] = { uR[1] * o[2] - uR[2] * o[1], uR[2] * o[0] - uR[0] * o[2], uR[0] * o[1] - uR[1] * o[0] }; /* uR x o = (uR^T [o]x)^T * -1 ... see note */
        /* uR^T [o]x = (o x uR)^T * (-1) = (uR x o)^T ; so de/dv = sgn * (-2) * (uR x o) */
        for (int j = 0; j < 3; ++j) {
            e->J[k][j] = sgn * uR[j];
            e->J[k][3 + j] = sgn * (-2.0) * c[j];
        }
    }
}
/* A.9: analytic Jacobians of EdgeSE3 / EdgeSE3Prior (exact derivative of toVectorMQT(E) w.r.t. the
 * right-multiplied MQT increments of Xi, Xj), written with quaternion products. */
static void se3_jacobians(og_graph* g, edge_t* e) {
    iso3 E;
    se3_error_iso(e, g, &E);
    memset(e->J[0], 0, sizeof(double) * 36);
    memset(e->J[1], 0, sizeof(double) * 36);
    if (e->kind == EK_PRIOR) {
        double qE[4];
        og_R_to_quat(E.R, qE);
        double s = quat_normalize_sign(qE); (void)s; /* qE now has w >= 0: it is the error quaternion */
        double* J = e->J[0];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) J[i * 6 + j] = E.R[i * 3 + j];
        for (int k = 0; k < 3; ++k) {
            double ek[4] = { 0, 0, 0, 0 }, r[4];
            ek[1 + k] = 1.0;
            quat_mul(qE, ek, r);
            for (int i = 0; i < 3; ++i) J[(3 + i) * 6 + 3 + k] = r[1 + i];
        }
        return;
    }
    /* binary: A = Z^-1, B = Xi^-1 Xj, E = A B */
    iso3 xi_inv, B;
    iso_inv(&g->V[e->v[0]].est, &xi_inv);
    iso_mul(&xi_inv, &g->V[e->v[1]].est, &B);
    double qA[4], qB[4], qE[4];
    og_R_to_quat(e->Zinv.R, qA);
    og_R_to_quat(B.R, qB);
    quat_mul(qA, qB, qE);
    double s = quat_normalize_sign(qE);
    /* Jj : E' = E * Delta */
    if (!g->V[e->v[1]].fixed) {
        double* J = e->J[1];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) J[i * 6 + j] = E.R[i * 3 + j];
        for (int k = 0; k < 3; ++k) {
            double ek[4] = { 0, 0, 0, 0 }, r[4];
            ek[1 + k] = 1.0;
            quat_mul(qE, ek, r);
            for (int i = 0; i < 3; ++i) J[(3 + i) * 6 + 3 + k] = r[1 + i];
        }
    }
    /* Ji : E' = A * Delta^-1 * B */
    if (!g->V[e->v[0]].fixed) {
        double* J = e->J[0];
        const double* RA = e->Zinv.R;
        const double* tB = B.t;
        double S[9] = { 0, -tB[2], tB[1], tB[2], 0, -tB[0], -tB[1], tB[0], 0 }; /* [tB]x */
        double RAS[9];
        mat3_mul(RA, S, RAS);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                J[i * 6 + j] = -RA[i * 3 + j];
                J[i * 6 + 3 + j] = 2.0 * RAS[i * 3 + j];
            }
        for (int k = 0; k < 3; ++k) {
            double ek[4] = { 0, 0, 0, 0 }, r1[4], r2[4];
            ek[1 + k] = 1.0;
            quat_mul(qA, ek, r1);
            quat_mul(r1, qB, r2);
            for (int i = 0; i < 3; ++i) J[(3 + i) * 6 + 3 + k] = -s * r2[1 + i];
        }
    }
}
static void edge_linearize(og_graph* g, edge_t* e, int jac_mode) {
    if (e->kind == EK_RANGE) {
        if (jac_mode == OG_JAC_NUMERIC_G2O) range_jacobian_numeric(g, e);
        else range_jacobian_analytic(g, e);
    } else {
        se3_jacobians(g, e);
    }
}

int og_debug_linearize(og_graph* g, int edge_index, int jac_mode, double* err6, double* J0, double* J1) {
    int n = 0;
    for (int i = 0; i < g->nE; ++i) {
        edge_t* e = &g->E[i];
        if (!e->alive) continue;
        if (n++ != edge_index) continue;
        edge_compute_error(g, e);
        edge_linearize(g, e, jac_mode);
        if (err6) memcpy(err6, e->err, sizeof(double) * 6);
        if (J0) memcpy(J0, e->J[0], sizeof(double) * 36);
        if (J1) memcpy(J1, e->J[1], sizeof(double) * 36);
        return e->dim;
    }
    return -1;
}

/* ------------------------------------------------------------------------------------------------
 * A.5 initializeOptimization
 * ---------------------------------------------------------------------------------------------- */
static void initialize_optimization(og_graph* g) {
    for (int i = 0; i < g->nV; ++i) { g->V[i].active = 0; g->V[i].hidx = -1; }
    g->act_e = (int*)realloc(g->act_e, sizeof(int) * (size_t)(g->nE + 1));
    g->n_act_e = 0;
    for (int i = 0; i < g->nE; ++i) {
        edge_t* e = &g->E[i];
        e->active = 0;
        if (!e->alive) continue;
        int all_fixed = 1;
        for (int k = 0; k < e->nv; ++k) if (!g->V[e->v[k]].fixed) all_fixed = 0;
        if (all_fixed) continue;
        e->active = 1;
        g->act_e[g->n_act_e++] = i;
        for (int k = 0; k < e->nv; ++k) g->V[e->v[k]].active = 1;
    }
    g->idx_v = (int*)realloc(g->idx_v, sizeof(int) * (size_t)(g->nV + 1));
    g->n_idx = 0;
    for (int i = 0; i < g->nV; ++i)
        if (g->V[i].alive && g->V[i].active && !g->V[i].fixed) g->idx_v[g->n_idx++] = i;
    for (int i = 1; i < g->n_idx; ++i) { /* Hessian order = ascending vertex id */
        int s = g->idx_v[i], j = i - 1;
        while (j >= 0 && g->V[g->idx_v[j]].id > g->V[s].id) { g->idx_v[j + 1] = g->idx_v[j]; --j; }
        g->idx_v[j + 1] = s;
    }
    for (int i = 0; i < g->n_idx; ++i) g->V[g->idx_v[i]].hidx = i;
}

static void compute_active_errors(og_graph* g) {
    for (int i = 0; i < g->n_act_e; ++i) edge_compute_error(g, &g->E[g->act_e[i]]);
}
/* A.4 activeRobustChi2 */
static double active_robust_chi2(og_graph* g) {
    double chi = 0;
    for (int i = 0; i < g->n_act_e; ++i) {
        const edge_t* e = &g->E[g->act_e[i]];
        double c = edge_chi2(e);
        if (e->robust) c = og_cauchy_rho(c, NULL);
        chi += c;
    }
    return chi;
}

/* A.4 buildSystem: linearize every active edge, accumulate H (lower + mirrored upper) and b */
static void build_system(og_graph* g, int jac_mode) {
    const int n = 6 * g->n_idx;
    memset(g->H, 0, sizeof(double) * (size_t)n * (size_t)n);
    memset(g->b, 0, sizeof(double) * (size_t)n);
    for (int a = 0; a < g->n_act_e; ++a) {
        edge_t* e = &g->E[g->act_e[a]];
        const double err_before[6] = { e->err[0], e->err[1], e->err[2], e->err[3], e->err[4], e->err[5] };
        edge_linearize(g, e, jac_mode);
        memcpy(e->err, err_before, sizeof(err_before)); /* linearizeOplus restores _error */
        const int D = e->dim;
        double omega[36], omega_r[6];
        double w = 1.0;
        if (e->robust) {
            double rho1;
            (void)og_cauchy_rho(edge_chi2(e), &rho1);
            w = rho1;
        }
        for (int i = 0; i < D; ++i) {
            double r = 0;
            for (int j = 0; j < D; ++j) { omega[i * D + j] = w * e->info[i * D + j]; r += e->info[i * D + j] * e->err[j]; }
            omega_r[i] = -w * r; /* -rho' * Omega * e */
        }
        for (int ka = 0; ka < e->nv; ++ka) {
            const vertex_t* va = &g->V[e->v[ka]];
            if (va->hidx < 0) continue;
            const double* Ja = e->J[ka];
            const int oa = 6 * va->hidx;
            /* b_a += Ja^T omega_r */
            for (int c = 0; c < 6; ++c) {
                double s = 0;
                for (int i = 0; i < D; ++i) s += Ja[i * 6 + c] * omega_r[i];
                g->b[oa + c] += s;
            }
            /* WJ = omega * Ja  (D x 6) */
            double WJ[36];
            for (int i = 0; i < D; ++i)
                for (int c = 0; c < 6; ++c) {
                    double s = 0;
                    for (int j = 0; j < D; ++j) s += omega[i * D + j] * Ja[j * 6 + c];
                    WJ[i * 6 + c] = s;
                }
            for (int kb = 0; kb < e->nv; ++kb) {
                const vertex_t* vb = &g->V[e->v[kb]];
                if (vb->hidx < 0) continue;
                const double* Jb = e->J[kb];
                const int ob = 6 * vb->hidx;
                /* H_ba += Jb^T * WJ_a  -> block (ob.., oa..) */
                for (int r = 0; r < 6; ++r)
                    for (int c = 0; c < 6; ++c) {
                        double s = 0;
                        for (int i = 0; i < D; ++i) s += Jb[i * 6 + r] * WJ[i * 6 + c];
                        g->H[(size_t)(ob + r) * n + (oa + c)] += s;
                    }
            }
        }
    }
}

/* exact solve of (H) x = b by dense Cholesky (A.8: the reference's CHOLMOD is an exact LL^T too; ordering
 * only changes round-off).  Row envelopes skip leading zeros so banded windows stay cheap. */
static int cholesky_solve(int n, double* A, const double* b, double* x) {
    int* first = (int*)malloc(sizeof(int) * (size_t)(n + 1));
    for (int i = 0; i < n; ++i) {
        int f = 0;
        while (f < i && A[(size_t)i * n + f] == 0.0) ++f;
        first[i] = f;
    }
    int ok = 1;
    for (int i = 0; i < n && ok; ++i) {
        for (int j = first[i]; j <= i; ++j) {
            double s = A[(size_t)i * n + j];
            int k0 = first[i] > first[j] ? first[i] : first[j];
            for (int k = k0; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
            if (i == j) {
                if (!(s > 0.0) || !isfinite(s)) { ok = 0; break; }
                A[(size_t)i * n + i] = sqrt(s);
            } else {
                A[(size_t)i * n + j] = s / A[(size_t)j * n + j];
            }
        }
    }
    if (ok) {
        for (int i = 0; i < n; ++i) {
            double s = b[i];
            for (int k = first[i]; k < i; ++k) s -= A[(size_t)i * n + k] * x[k];
            x[i] = s / A[(size_t)i * n + i];
        }
        for (int i = n - 1; i >= 0; --i) {
            double s = x[i];
            for (int k = i + 1; k < n; ++k)
                if (first[k] <= i) s -= A[(size_t)k * n + i] * x[k];
            x[i] = s / A[(size_t)i * n + i];
        }
    }
    free(first);
    return ok;
}

/* ------------------------------------------------------------------------------------------------
 * A.6 optimize(n) with OptimizationAlgorithmLevenberg::solve
 * ---------------------------------------------------------------------------------------------- */
int og_optimize(og_graph* g, int iterations, int jac_mode, og_stats* st) {
    og_stats local;
    if (!st) st = &local;
    memset(st, 0, sizeof(*st));
    initialize_optimization(g);
    if (g->n_idx == 0) return -1; /* "0 vertices to optimize" */
    const int n = 6 * g->n_idx;
    if (n > g->hcap) {
        g->hcap = n;
        g->H = (double*)realloc(g->H, sizeof(double) * (size_t)n * n);
        g->Hs = (double*)realloc(g->Hs, sizeof(double) * (size_t)n * n);
        g->b = (double*)realloc(g->b, sizeof(double) * (size_t)n);
        g->x = (double*)realloc(g->x, sizeof(double) * (size_t)n);
    }
    memset(g->x, 0, sizeof(double) * (size_t)n);

    const double tau = 1e-5, good_lo = 1.0 / 3.0, good_hi = 2.0 / 3.0;
    const int max_trials = 10;
    int ok = 1, it = 0;
    double current_chi = 0;
    for (it = 0; it < iterations && ok; ++it) {
        compute_active_errors(g);
        current_chi = active_robust_chi2(g);
        double temp_chi = current_chi;
        build_system(g, jac_mode);
        if (it == 0) { /* computeLambdaInit */
            double max_diag = 0;
            for (int j = 0; j < n; ++j) {
                double d = fabs(g->H[(size_t)j * n + j]);
                if (d > max_diag) max_diag = d;
            }
            g->lambda = tau * max_diag;
            g->ni = 2;
        }
        double rho = 0;
        int qmax = 0;
        do {
            for (int i = 0; i < g->n_idx; ++i) g->V[g->idx_v[i]].backup = g->V[g->idx_v[i]].est; /* push */
            memcpy(g->Hs, g->H, sizeof(double) * (size_t)n * n);
            for (int j = 0; j < n; ++j) g->Hs[(size_t)j * n + j] += g->lambda; /* setLambda: +lambda*I */
            int ok2 = cholesky_solve(n, g->Hs, g->b, g->x);
            /* (a failed factorisation leaves x as the previous solve's; g2o applies it all the same) */
            for (int i = 0; i < g->n_idx; ++i) {
                vertex_t* v = &g->V[g->idx_v[i]];
                iso3 nx;
                vertex_oplus(&v->est, g->x + 6 * i, &nx);
                v->est = nx;
            }
            ++st->lm_trials;
            compute_active_errors(g);
            temp_chi = active_robust_chi2(g);
            if (!ok2) temp_chi = DBL_MAX;
            rho = current_chi - temp_chi;
            double scale = 0; /* computeScale */
            for (int j = 0; j < n; ++j) scale += g->x[j] * (g->lambda * g->x[j] + g->b[j]);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && isfinite(temp_chi)) {
                double alpha = 1. - pow((2 * rho - 1), 3);
                alpha = alpha < good_hi ? alpha : good_hi;
                double scale_factor = good_lo > alpha ? good_lo : alpha;
                g->lambda *= scale_factor;
                g->ni = 2;
                current_chi = temp_chi;
                /* discardTop */
            } else {
                g->lambda *= g->ni;
                g->ni *= 2;
                for (int i = 0; i < g->n_idx; ++i) g->V[g->idx_v[i]].est = g->V[g->idx_v[i]].backup; /* pop */
            }
            ++qmax;
        } while (rho < 0 && qmax < max_trials);
        if (qmax == max_trials || rho == 0) { ok = 0; st->terminated = 1; }
    }
    st->outer_iterations = it;
    st->lambda = g->lambda;
    st->robust_chi2 = current_chi;
    return it;
}

/* A.7 OptimizableGraph::chi2(): all edges, last computed errors, not robustified */
double og_chi2(og_graph* g) {
    double chi = 0;
    for (int i = 0; i < g->nE; ++i)
        if (g->E[i].alive) chi += edge_chi2(&g->E[i]);
    return chi;
}

/* ------------------------------------------------------------------------------------------------
 * BASELINE config 2 through the general graph: one moving tag vertex, M fixed anchor vertices,
 * M Cauchy range edges per update (cost definition: localization.cpp:306-318,608-627).
 * ---------------------------------------------------------------------------------------------- */
int og_snapshot_batch(int B, int K, int M, const double* anchors, const float* dist, const float* err,
                      double* pos, double* out_pos, double* out_chi2, unsigned char* out_trials,
                      int iterations, double gate, int gate_from_epoch, int jac_mode) {
    iso3 I;
    iso_identity(&I);
    for (int b = 0; b < B; ++b) {
        double p[3] = { pos[0 * (size_t)B + b], pos[1 * (size_t)B + b], pos[2 * (size_t)B + b] };
        for (int k = 0; k < K; ++k) {
            og_graph* g = og_create();
            og_add_vertex(g, 1000, I.R, p, 0);
            int n_edges = 0;
            for (int m = 0; m < M; ++m) {
                const double* a = anchors + 3 * m;
                og_add_vertex(g, m, I.R, a, 1);
                size_t o = ((size_t)k * M + m) * (size_t)B + b;
                double d = (double)dist[o];
                double e = (double)err[o];
                /* localization.cpp:306-313 outlier gate on vertex origins */
                double dhat = sqrt((p[0] - a[0]) * (p[0] - a[0]) + (p[1] - a[1]) * (p[1] - a[1]) + (p[2] - a[2]) * (p[2] - a[2]));
                if (gate > 0 && k >= gate_from_epoch && fabs(dhat - d) > gate) continue;
                if (!(e > 0) || !isfinite(e) || !isfinite(d)) continue; /* padded / invalid slot */
                double cov = pow(e, 2);                                  /* :318 */
                og_add_range_edge(g, 1000, m, d, 1.0 / cov, NULL, NULL, 1); /* :608-627 */
                ++n_edges;
            }
            og_stats st;
            memset(&st, 0, sizeof(st));
            double chi = 0;
            if (n_edges > 0) {
                og_optimize(g, iterations, jac_mode, &st);
                chi = og_chi2(g);
                og_get_estimate(g, 1000, NULL, p);
            }
            og_destroy(g);
            out_pos[((size_t)k * 3 + 0) * B + b] = p[0];
            out_pos[((size_t)k * 3 + 1) * B + b] = p[1];
            out_pos[((size_t)k * 3 + 2) * B + b] = p[2];
            out_chi2[(size_t)k * B + b] = chi;
            if (out_trials) out_trials[(size_t)k * B + b] = (unsigned char)(st.lm_trials > 255 ? 255 : st.lm_trials);
        }
        pos[0 * (size_t)B + b] = p[0]; pos[1 * (size_t)B + b] = p[1]; pos[2 * (size_t)B + b] = p[2];
    }
    return 0;
}

/* tf::poseEigenToMsg: Quaterniond(R), flipped to w >= 0; out x y z w */
static void R_to_qxyzw(const double* R, double* q) {
    double w[4];
    og_R_to_quat(R, w);
    double s = w[0] < 0 ? -1.0 : 1.0;
    q[0] = s * w[1]; q[1] = s * w[2]; q[2] = s * w[3]; q[3] = s * w[0];
}

int og_fusion_batch(int B, int K, int M, const double* anchors, const double* off, const float* dist, const float* err,
                    const double* imu, double* pose, double* out_pose, double* out_chi2, unsigned char* out_trials,
                    int iterations, double gate, int gate_from_epoch, int jac_mode) {
    const double I3[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    for (int b = 0; b < B; ++b) {
        double t[3] = { pose[0 * (size_t)B + b], pose[1 * (size_t)B + b], pose[2 * (size_t)B + b] };
        double q0[4] = { pose[6 * (size_t)B + b], pose[3 * (size_t)B + b], pose[4 * (size_t)B + b], pose[5 * (size_t)B + b] };
        double R[9];
        og_quat_to_R(q0, R);
        for (int k = 0; k < K; ++k) {
            const double* im = imu + ((size_t)k * B + b) * 8;
            double qi[4] = { im[3], im[0], im[1], im[2] };
            og_quat_to_R(qi, R); /* rotation overwritten, translation kept */
            og_graph* g = og_create();
            og_add_vertex(g, 1000, R, t, 0);
            double info[36];
            memset(info, 0, sizeof(info));
            info[3 * 6 + 3] = 1.0 / im[4]; info[4 * 6 + 4] = 1.0 / im[5]; info[5 * 6 + 5] = 1.0 / im[6];
            og_add_prior_edge(g, 1000, R, t, info);
            for (int m = 0; m < M; ++m) {
                const double* a = anchors + 3 * m;
                og_add_vertex(g, m, I3, a, 1);
                size_t o = ((size_t)k * M + m) * (size_t)B + b;
                double d = (double)dist[o], e = (double)err[o];
                double dhat = sqrt((t[0] - a[0]) * (t[0] - a[0]) + (t[1] - a[1]) * (t[1] - a[1]) + (t[2] - a[2]) * (t[2] - a[2]));
                if (gate > 0 && k >= gate_from_epoch && fabs(dhat - d) > gate) continue;
                if (!(e > 0) || !isfinite(e) || !isfinite(d)) continue;
                og_add_range_edge(g, 1000, m, d, 1.0 / pow(e, 2), off, NULL, 1);
            }
            og_stats st;
            memset(&st, 0, sizeof(st));
            og_optimize(g, iterations, jac_mode, &st);
            double chi = og_chi2(g);
            og_get_estimate(g, 1000, R, t);
            og_destroy(g);
            double q[4];
            R_to_qxyzw(R, q);
            double* op = out_pose + (size_t)k * 7 * B;
            op[0 * (size_t)B + b] = t[0]; op[1 * (size_t)B + b] = t[1]; op[2 * (size_t)B + b] = t[2];
            op[3 * (size_t)B + b] = q[0]; op[4 * (size_t)B + b] = q[1]; op[5 * (size_t)B + b] = q[2]; op[6 * (size_t)B + b] = q[3];
            out_chi2[(size_t)k * B + b] = chi;
            if (out_trials) out_trials[(size_t)k * B + b] = (unsigned char)(st.lm_trials > 255 ? 255 : st.lm_trials);
        }
        double q[4];
        R_to_qxyzw(R, q);
        pose[0 * (size_t)B + b] = t[0]; pose[1 * (size_t)B + b] = t[1]; pose[2 * (size_t)B + b] = t[2];
        pose[3 * (size_t)B + b] = q[0]; pose[4 * (size_t)B + b] = q[1]; pose[5 * (size_t)B + b] = q[2]; pose[6 * (size_t)B + b] = q[3];
    }
    return 0;
}
