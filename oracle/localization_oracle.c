/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see localization_oracle.h; PARITY UNPINNED).
 * Restates /root/reference/src/localization/localization.cpp and robot.cpp; line cites inline.
 */
#include "localization_oracle.h"
#include "g2o_graph_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define FRAME_LEN 192
enum { ST_GENERAL = 0, ST_POSE = 1, ST_RANGE = 2, ST_TWIST = 3, ST_IMU = 4, ST_COUNT = 5 }; /* localization.h:90-97 */

typedef struct { double stamp; char frame_id[FRAME_LEN]; } header_t;

/* class Robot, robot.h:60-120 */
typedef struct {
    int ID, is_static, T;
    int index;                    /* current vertex slot */
    header_t* header;             /* per slot */
    int type_has[ST_COUNT]; int type_index[ST_COUNT];
    int hdr_has[ST_COUNT]; header_t headers[ST_COUNT];
} robot_t;

struct lo_state {
    lo_config cfg;
    og_graph* g;
    int n_nodes; robot_t* robots;
    int self_id;
    int n_antenna; double* antenna; /* offsets[] translations */
    int number_measurements;
    int key_vertex; /* vertex id, -1 = unset (reference: uninitialised pointer, localization.h:149) */
};

static int slot_vertex_id(const robot_t* r, int slot) { return r->ID + slot * 300; } /* robot.cpp:43,94 */

static robot_t* robot_at(lo_state* s, int id) {
    for (int i = 0; i < s->n_nodes; ++i) if (s->robots[i].ID == id) return &s->robots[i];
    return NULL;
}

/* Robot::init, robot.cpp:31-58 */
static void robot_init(lo_state* s, robot_t* r, const double* R, const double* t) {
    r->index = 0;
    r->header = (header_t*)calloc((size_t)r->T, sizeof(header_t));
    for (int i = 0; i < r->T; ++i) og_add_vertex(s->g, slot_vertex_id(r, i), R, t, r->is_static);
    strcpy(r->header[0].frame_id, "none");
}
/* Robot::last_vertex(type), robot.cpp:113-118 */
static int robot_last_vertex_type(robot_t* r, int type) {
    if (!r->type_has[type]) { r->type_has[type] = 1; r->type_index[type] = r->index; }
    if (!r->hdr_has[type]) { r->hdr_has[type] = 1; r->headers[type] = r->header[r->index]; }
    return slot_vertex_id(r, r->type_index[type]);
}
static int robot_last_vertex(const robot_t* r) { return slot_vertex_id(r, r->index); } /* robot.cpp:121-124 */
/* Robot::last_header(type), robot.cpp:127-131 */
static header_t robot_last_header_type(robot_t* r, int type) {
    if (!r->hdr_has[type]) { r->hdr_has[type] = 1; r->headers[type] = r->header[r->index]; }
    return r->headers[type];
}
/* Robot::new_vertex, robot.cpp:75-110 */
static int robot_new_vertex(lo_state* s, robot_t* r, int type, const header_t* h) {
    if (!r->type_has[type]) { r->type_has[type] = 1; r->type_index[type] = r->index; }
    if (!r->hdr_has[type]) { r->hdr_has[type] = 1; r->headers[type] = *h; }
    if (r->is_static) {
        r->header[r->index] = *h;
        return robot_last_vertex_type(r, type);
    }
    double R[9], t[3];
    og_get_estimate(s->g, slot_vertex_id(r, r->index), R, t); /* copy previous estimate, :90 */
    r->index = (r->index + 1) % r->T;
    int id = slot_vertex_id(r, r->index);
    og_remove_vertex(s->g, id); /* oldest pose and its edges, :96 */
    og_add_vertex(s->g, id, R, t, 0);
    r->header[r->index] = *h;
    r->type_index[type] = r->index;
    r->headers[type] = *h;
    return id;
}

lo_state* lo_create(const lo_config* cfg, int n_nodes, const int* ids, const double* pos,
                    int n_antenna, const double* antenna_xyz) {
    if (n_nodes <= 0) return NULL;
    lo_state* s = (lo_state*)calloc(1, sizeof(lo_state));
    s->cfg = *cfg;
    s->g = og_create();
    s->n_nodes = n_nodes;
    s->robots = (robot_t*)calloc((size_t)n_nodes, sizeof(robot_t));
    s->self_id = ids[n_nodes - 1]; /* localization.cpp:89 */
    s->key_vertex = -1;
    const double I[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    for (int i = 0; i < n_nodes; ++i) { /* :92-108 */
        robot_t* r = &s->robots[i];
        r->ID = ids[i];
        if (cfg->has_relative_range || ids[i] == s->self_id) { r->is_static = 0; r->T = cfg->trajectory_length; }
        else { r->is_static = 1; r->T = 1; }
        robot_init(s, r, I, pos + 3 * i);
    }
    if (antenna_xyz && n_antenna > 0) { /* :111-123 */
        s->n_antenna = n_antenna;
        s->antenna = (double*)malloc(sizeof(double) * 3 * (size_t)n_antenna);
        memcpy(s->antenna, antenna_xyz, sizeof(double) * 3 * (size_t)n_antenna);
    } else { /* localization.h:170: three identity offsets */
        s->n_antenna = 3;
        s->antenna = (double*)calloc(9, sizeof(double));
    }
    return s;
}
void lo_destroy(lo_state* s) {
    if (!s) return;
    for (int i = 0; i < s->n_nodes; ++i) free(s->robots[i].header);
    free(s->robots); free(s->antenna);
    og_destroy(s->g);
    free(s);
}
int lo_number_measurements(lo_state* s) { return s->number_measurements; }

static void pose_from_vertex(lo_state* s, int vid, double stamp, double* out8) {
    double R[9], t[3], q[4];
    og_get_estimate(s->g, vid, R, t);
    og_R_to_quat(R, q); /* tf::poseEigenToMsg: Quaterniond(e.linear()), flipped to w >= 0 */
    if (q[0] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    out8[0] = stamp; out8[1] = t[0]; out8[2] = t[1]; out8[3] = t[2];
    out8[4] = q[1]; out8[5] = q[2]; out8[6] = q[3]; out8[7] = q[0];
}
/* Robot::vertices2path, robot.cpp:61-72 */
int lo_get_path(lo_state* s, int node_id, double* out) {
    robot_t* r = robot_at(s, node_id);
    if (!r) return -1;
    for (int i = 0; i < r->T; ++i) {
        int idx = (r->index + 1 + i) % r->T;
        pose_from_vertex(s, slot_vertex_id(r, idx), r->header[idx].stamp, out + 8 * i);
    }
    return r->T;
}

/* Localization::solve + publish, localization.cpp:164-251 */
static void solve_and_publish(lo_state* s, lo_output* out) {
    og_stats st;
    og_optimize(s->g, s->cfg.maximum_iteration, s->cfg.jac_mode, &st); /* return value ignored, :170 */
    if (!out) return;
    out->solved = 1;
    out->outer_iterations = st.outer_iterations;
    out->lm_trials = st.lm_trials;
    out->chi2 = og_chi2(s->g);
    out->published = out->chi2 < s->cfg.minimum_optimize_error; /* :199-205 */
    robot_t* r = robot_at(s, s->self_id);
    pose_from_vertex(s, robot_last_vertex(r), r->header[r->index].stamp, out->realtime);
    int i = s->cfg.trajectory_length / 2; /* :220 */
    int idx = (r->index + 1 + i) % r->T;
    pose_from_vertex(s, slot_vertex_id(r, idx), r->header[idx].stamp, out->optimized);
}

int lo_solve(lo_state* s, lo_output* out) {
    if (out) memset(out, 0, sizeof(*out));
    solve_and_publish(s, out);
    return 1;
}

/* Localization::addRangeEdge, localization.cpp:297-376 */
int lo_add_range(lo_state* s, int requester_id, int responder_id, double stamp, float distance,
                 float distance_err, int antenna, const char* frame_id, lo_output* out) {
    if (out) memset(out, 0, sizeof(*out));
    robot_t* rq = robot_at(s, requester_id);
    robot_t* rs = robot_at(s, responder_id);
    if (!rq || !rs) return -2;
    ++s->number_measurements; /* :303 */
    double tq[3], tr[3];
    og_get_estimate(s->g, robot_last_vertex(rq), NULL, tq);
    og_get_estimate(s->g, robot_last_vertex(rs), NULL, tr);
    double distance_estimation = sqrt((tq[0] - tr[0]) * (tq[0] - tr[0]) + (tq[1] - tr[1]) * (tq[1] - tr[1]) + (tq[2] - tr[2]) * (tq[2] - tr[2]));
    if (s->number_measurements > s->cfg.trajectory_length &&
        fabs(distance_estimation - (double)distance) > s->cfg.distance_outlier) /* :309-313 */
        return 0;
    double dt_requester = stamp - rq->header[rq->index].stamp; /* :316 */
    double dt_responder = stamp - rs->header[rs->index].stamp; /* :317 */
    double distance_cov = pow((double)distance_err, 2);         /* :318 */
    double cov_requester = pow(s->cfg.maximum_velocity * dt_requester / 3, 2); /* :319 */
    int vertex_last_requester = robot_last_vertex(rq);
    int vertex_last_responder = robot_last_vertex(rs);
    header_t h;
    memset(&h, 0, sizeof(h));
    h.stamp = stamp;
    snprintf(h.frame_id, FRAME_LEN, "%s", frame_id ? frame_id : "");
    int vertex_responder = robot_new_vertex(s, rs, ST_RANGE, &h); /* :323 */
    /* a moving responder's previous pose may have been the slot just recycled when T == 1; ignore */
    const char* last_frame = rq->header[rq->index].frame_id; /* :325 */
    if (strstr(last_frame, h.frame_id) != NULL || strstr(last_frame, "none") != NULL) { /* :327 */
        int vertex_requester = robot_new_vertex(s, rq, ST_RANGE, &h); /* :329 */
        const double* off0 = NULL;
        if (antenna > 0) { /* :333-334 */
            if (antenna - 1 >= s->n_antenna) return -3;
            off0 = s->antenna + 3 * (antenna - 1);
        }
        og_add_range_edge(s->g, vertex_requester, vertex_responder, (double)distance, 1.0 / distance_cov, off0, NULL, 1); /* :331-336 */
        og_add_range_edge(s->g, vertex_last_requester, vertex_requester, 0.0, 1.0 / cov_requester, NULL, NULL, 1);      /* :338-340 */
    } else {
        og_add_range_edge(s->g, vertex_last_requester, vertex_responder, (double)distance,
                          1.0 / (distance_cov + cov_requester), NULL, NULL, 1); /* :348-350 */
    }
    if (!rs->is_static) { /* :360-369 */
        double cov_responder = pow(s->cfg.maximum_velocity * dt_responder / 3, 2);
        og_add_range_edge(s->g, vertex_last_responder, vertex_responder, 0.0, 1.0 / cov_responder, NULL, NULL, 1);
    }
    if (s->cfg.publish_range && s->number_measurements > s->cfg.trajectory_length) { /* :371-375 */
        solve_and_publish(s, out);
        return 1;
    }
    return 0;
}

/* Localization::addRLRangeEdge, localization.cpp:378-436 */
int lo_add_rl_range(lo_state* s, int requester_id, int responder_id, double stamp, double d,
                    const double* v, lo_output* out) {
    if (out) memset(out, 0, sizeof(*out));
    robot_t* rq = robot_at(s, requester_id);
    robot_t* rs = robot_at(s, responder_id);
    if (!rq || !rs) return -2;
    header_t h; /* :380-382 */
    memset(&h, 0, sizeof(h));
    h.stamp = stamp;
    snprintf(h.frame_id, FRAME_LEN, "uwb");
    double dt_requester = stamp - rq->header[rq->index].stamp; /* :384 */
    double dt_responder = stamp - rs->header[rs->index].stamp; /* :385 */
    double distance_cov = pow(0.054, 2);                        /* :387 */
    double cov_requester = pow(s->cfg.maximum_velocity * dt_requester / 3, 2); /* :388 */
    double cov_responder = pow(s->cfg.maximum_velocity * dt_responder / 3, 2); /* :389 */
    int vertex_last_requester = robot_last_vertex(rq);
    int vertex_last_responder = robot_last_vertex(rs);
    int vertex_requester = robot_new_vertex(s, rq, ST_RANGE, &h); /* :394 */
    int vertex_responder = robot_new_vertex(s, rs, ST_RANGE, &h); /* :395 */
    og_add_range_edge(s->g, vertex_requester, vertex_responder, d, 1.0 / distance_cov, NULL, NULL, 1); /* :397-398 */
    if (!rs->is_static) /* :400-405 */
        og_add_range_edge(s->g, vertex_last_responder, vertex_responder, 0.0, 1.0 / cov_responder, NULL, NULL, 1);
    if (!rq->is_static) { /* :408-428: EdgeSE3 from the velocity, information on the translation only, no robust kernel */
        const double I[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
        double tm[3] = { dt_requester * v[0], dt_requester * v[1], dt_requester * v[2] };
        double info[36];
        memset(info, 0, sizeof(info));
        info[0] = info[7] = info[14] = 1.0 / cov_requester;
        og_add_se3_edge(s->g, vertex_last_requester, vertex_requester, I, tm, info, 0);
    }
    if (s->cfg.publish_relative_range) { solve_and_publish(s, out); return 1; } /* :430-434 */
    return 0;
}

/* Localization::addImuEdge, localization.cpp:499-535 */
int lo_add_imu(lo_state* s, double stamp, const double* q_xyzw, const double* cov9, const char* frame_id, lo_output* out) {
    (void)stamp;
    if (out) memset(out, 0, sizeof(*out));
    robot_t* r = robot_at(s, s->self_id);
    if (strstr(r->header[r->index].frame_id, frame_id) == NULL) { /* :501 */
        size_t L = strlen(r->header[r->index].frame_id);
        snprintf(r->header[r->index].frame_id + L, FRAME_LEN - L, "-%s", frame_id); /* robot.cpp:140-143 */
        int vid = robot_last_vertex_type(r, ST_RANGE); /* :505 */
        double R[9], t[3];
        double q[4] = { q_xyzw[3], q_xyzw[0], q_xyzw[1], q_xyzw[2] };
        og_quat_to_R(q, R); /* Identity.rotate(q) */
        og_get_estimate(s->g, vid, NULL, t);
        og_set_estimate(s->g, vid, R, t); /* :513 */
        double info[36];
        memset(info, 0, sizeof(info));
        info[3 * 6 + 3] = 1.0 / cov9[0];
        info[4 * 6 + 4] = 1.0 / cov9[4];
        info[5 * 6 + 5] = 1.0 / cov9[8]; /* :516-518 */
        og_add_prior_edge(s->g, vid, R, t, info); /* :520-525 */
    }
    if (s->cfg.publish_imu) { solve_and_publish(s, out); return 1; } /* :530-534 */
    return 0;
}

/* Localization::addLidarEdge, localization.cpp:462-496 */
int lo_add_lidar(lo_state* s, double stamp, double z, const char* frame_id, lo_output* out) {
    (void)stamp;
    if (out) memset(out, 0, sizeof(*out));
    robot_t* r = robot_at(s, s->self_id);
    if (strstr(r->header[r->index].frame_id, frame_id) == NULL) {
        size_t L = strlen(r->header[r->index].frame_id);
        snprintf(r->header[r->index].frame_id + L, FRAME_LEN - L, "-%s", frame_id);
        int vid = robot_last_vertex_type(r, ST_RANGE);
        double R[9], t[3];
        og_get_estimate(s->g, vid, R, t);
        t[2] = z; /* :474 */
        og_set_estimate(s->g, vid, R, t);
        double info[36];
        memset(info, 0, sizeof(info));
        info[2 * 6 + 2] = 1 / 0.05; /* :479 */
        og_add_prior_edge(s->g, vid, R, t, info);
    }
    if (s->cfg.publish_lidar) { solve_and_publish(s, out); return 1; }
    return 0;
}

/* 6x6 inverse (Eigen MatrixXd::inverse = partial-pivot LU; same result up to round-off) */
static int inv6(const double* A, double* Ainv) {
    double M[6][12];
    for (int i = 0; i < 6; ++i) {
        for (int j = 0; j < 6; ++j) { M[i][j] = A[i * 6 + j]; M[i][6 + j] = (i == j) ? 1.0 : 0.0; }
    }
    for (int c = 0; c < 6; ++c) {
        int p = c;
        for (int r = c + 1; r < 6; ++r) if (fabs(M[r][c]) > fabs(M[p][c])) p = r;
        if (M[p][c] == 0.0) return -1;
        if (p != c) for (int j = 0; j < 12; ++j) { double tmp = M[c][j]; M[c][j] = M[p][j]; M[p][j] = tmp; }
        double d = M[c][c];
        for (int j = 0; j < 12; ++j) M[c][j] /= d;
        for (int r = 0; r < 6; ++r) {
            if (r == c) continue;
            double f = M[r][c];
            if (f != 0.0) for (int j = 0; j < 12; ++j) M[r][j] -= f * M[c][j];
        }
    }
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) Ainv[i * 6 + j] = M[i][6 + j];
    return 0;
}

/* Localization::addPoseEdge, localization.cpp:254-290 */
int lo_add_pose(lo_state* s, double stamp, const double* p7, const double* cov36, const char* frame_id, lo_output* out) {
    if (out) memset(out, 0, sizeof(*out));
    robot_t* r = robot_at(s, s->self_id);
    header_t h;
    memset(&h, 0, sizeof(h));
    h.stamp = stamp;
    snprintf(h.frame_id, FRAME_LEN, "%s", frame_id ? frame_id : "");
    header_t lh = robot_last_header_type(r, ST_POSE);
    if (strcmp(h.frame_id, lh.frame_id) != 0) s->key_vertex = robot_last_vertex_type(r, ST_POSE); /* :258-259 */
    int nv = robot_new_vertex(s, r, ST_POSE, &h); /* :261 */
    if (s->key_vertex < 0 || !og_has_vertex(s->g, s->key_vertex)) return -4;
    double Rm[9], q[4] = { p7[6], p7[3], p7[4], p7[5] };
    og_quat_to_R(q, Rm); /* tf::poseMsgToEigen: Translation * Quaterniond */
    double info[36];
    if (inv6(cov36, info) != 0) return -5; /* :275-277 */
    og_add_se3_edge(s->g, s->key_vertex, nv, Rm, p7, info, 1); /* :263-281 */
    if (s->cfg.publish_pose) { solve_and_publish(s, out); return 1; } /* :285-289 */
    return 0;
}

/* Localization::addTwistEdge + twist2transform + create_se3_edge_from_twist, localization.cpp:438-459,560-605 */
int lo_add_twist(lo_state* s, double stamp, const double* tw, const double* cov36, const char* frame_id, lo_output* out) {
    if (out) memset(out, 0, sizeof(*out));
    robot_t* r = robot_at(s, s->self_id);
    double dt = stamp - r->header[r->index].stamp; /* :442 */
    int last_vertex = robot_last_vertex(r);
    header_t h;
    memset(&h, 0, sizeof(h));
    h.stamp = stamp;
    snprintf(h.frame_id, FRAME_LEN, "%s", frame_id ? frame_id : "");
    int nv = robot_new_vertex(s, r, ST_TWIST, &h);
    /* tf::Quaternion::setRPY(roll, pitch, yaw) */
    double hr = tw[3] * dt * 0.5, hp = tw[4] * dt * 0.5, hy = tw[5] * dt * 0.5;
    double cy = cos(hy), sy = sin(hy), cp = cos(hp), sp = sin(hp), cr = cos(hr), sr = sin(hr);
    double qx = sr * cp * cy - cr * sp * sy;
    double qy = cr * sp * cy + sr * cp * sy;
    double qz = cr * cp * sy - sr * sp * cy;
    double qw = cr * cp * cy + sr * sp * sy;
    /* tf::Matrix3x3::setRotation normalises by 2/length2 */
    double d = qx * qx + qy * qy + qz * qz + qw * qw;
    double sc = 2.0 / d;
    double xs = qx * sc, ys = qy * sc, zs = qz * sc;
    double wx = qw * xs, wy = qw * ys, wz = qw * zs, xx = qx * xs, xy = qx * ys, xz = qx * zs, yy = qy * ys, yz = qy * zs, zz = qz * zs;
    double Rm[9] = { 1.0 - (yy + zz), xy - wz, xz + wy, xy + wz, 1.0 - (xx + zz), yz - wx, xz - wy, yz + wx, 1.0 - (xx + yy) };
    double tm[3] = { tw[0] * dt, tw[1] * dt, tw[2] * dt };
    double cov[36], info[36];
    for (int i = 0; i < 36; ++i) cov[i] = cov36[i] * dt * dt; /* :579 */
    if (inv6(cov, info) != 0) return -5;
    og_add_se3_edge(s->g, last_vertex, nv, Rm, tm, info, 1);
    if (s->cfg.publish_twist) { solve_and_publish(s, out); return 1; }
    return 0;
}
