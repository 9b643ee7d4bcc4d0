/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see g2o_graph_oracle.h; PARITY UNPINNED).
 *
 * CPU restatement of the reference's graph front-end — class Localization
 * (/root/reference/src/localization/localization.{h,cpp}) and class Robot
 * (/root/reference/src/localization/robot.{h,cpp}) — on top of the g2o restatement in
 * g2o_graph_oracle.c.  ROS messages are replaced by their numeric fields; everything that decides the
 * cost function (gates, covariances, edge topology, ring window, publish selection) follows the cited lines.
 */
#ifndef LOCALIZATION_ORACLE_H
#define LOCALIZATION_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lo_config {
    int trajectory_length;          /* robot/trajectory_length        localization.cpp:72 */
    double maximum_velocity;        /* robot/maximum_velocity (1.0)    :75 */
    double distance_outlier;        /* robot/distance_outlier (1.0)    :78 */
    int maximum_iteration;          /* optimizer/maximum_iteration (20) :65 */
    double minimum_optimize_error;  /* optimizer/minimum_optimize_error (1000) :68 */
    int publish_range, publish_pose, publish_twist, publish_lidar, publish_imu; /* :146-158 */
    int has_relative_range;         /* topic/relative_range present => every node moves, :94 */
    int jac_mode;                   /* OG_JAC_NUMERIC_G2O (reference behaviour) or OG_JAC_ANALYTIC */
    int publish_relative_range;     /* publish_flag/relative_range, :158 */
} lo_config;

typedef struct lo_output {
    int solved;         /* a solve() ran for this message */
    int published;      /* chi2 < minimum_optimize_error (localization.cpp:197-205) */
    double chi2;        /* optimizer.chi2() */
    double realtime[8];  /* stamp, x y z, qx qy qz qw  — robots[self].current_pose()   :208 */
    double optimized[8]; /* path->poses[trajectory_length/2]                           :220 */
    int outer_iterations, lm_trials;
} lo_output;

typedef struct lo_state lo_state;

/* ids[n-1] is the moving tag (self_id = nodesId.back(), localization.cpp:89). antenna_xyz may be NULL
 * (default 3 identity offsets, localization.h:170). */
lo_state* lo_create(const lo_config* cfg, int n_nodes, const int* ids, const double* pos_xyz,
                    int n_antenna, const double* antenna_xyz);
void lo_destroy(lo_state* s);

/* Each returns <0 on error (unknown node id = the reference's std::map::at throw, :306). */
int lo_add_range(lo_state* s, int requester_id, int responder_id, double stamp, float distance,
                 float distance_err, int antenna, const char* frame_id, lo_output* out);
int lo_add_imu(lo_state* s, double stamp, const double* q_xyzw, const double* orientation_cov9,
               const char* frame_id, lo_output* out);
int lo_add_pose(lo_state* s, double stamp, const double* pose_xyz_qxyzw, const double* cov36,
                const char* frame_id, lo_output* out);
int lo_add_twist(lo_state* s, double stamp, const double* twist_lin_ang6, const double* cov36,
                 const char* frame_id, lo_output* out);
int lo_add_lidar(lo_state* s, double stamp, double z, const char* frame_id, lo_output* out);
/* Localization::addRLRangeEdge (localization.cpp:378-436; compiled only with -DRELATIVE_LOCALIZATION, CMakeLists.txt:137):
 * uwb_reloc::uwbTalkData fields time_stamp, rqstrId, rspdrId, d, rqstr_vx/vy/vz */
int lo_add_rl_range(lo_state* s, int requester_id, int responder_id, double stamp, double d,
                    const double* requester_velocity_xyz, lo_output* out);

/* Localization::solve() + publish() on demand (localization.cpp:164-251) */
int lo_solve(lo_state* s, lo_output* out);

/* Robot::vertices2path for a node: out[T][8] (stamp, xyz, qxyzw), oldest first. Returns T. */
int lo_get_path(lo_state* s, int node_id, double* out);
int lo_number_measurements(lo_state* s);

#ifdef __cplusplus
}
#endif
#endif
