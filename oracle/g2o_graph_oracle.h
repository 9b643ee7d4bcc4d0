/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under localization_amd/ may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY UNPINNED: the arithmetic this file restates lives in g2o @ deafc01ee8315b9405351fb145238c5d62f82dc7
 * (reference README.md:24-33), which is neither vendored under /root/reference nor installed here, and the
 * reference ships no tests or golden vectors for this path (reference CMakeLists.txt:206-213 are commented out).
 * The restatement follows the published g2o algorithm and the reference's call sites;
 * it is cross-checked against scipy least_squares(loss='cauchy') minima and closed-form known answers
 * (tests/golden/, tests/test_oracle_*.py), not against outputs of the reference itself.
 *
 * What is restated (plain C, IEEE double, single thread — like the reference, localization_node.cpp:98):
 *   - g2o::VertexSE3 (estimate = Isometry3d, oplus X <- X * fromVectorMQT(d))      robot.cpp:41-52,88-94
 *   - g2o::EdgeSE3Range  e = d - ||(X0*O0).t - (X1*O1).t||, numeric Jacobian       types_edge_se3range.cpp:105-114
 *   - g2o::EdgeSE3Prior  e = toVectorMQT(Z^-1 * X * P), P = identity               localization.cpp:481-486,520-525
 *   - g2o::EdgeSE3       e = toVectorMQT(Z^-1 * Xi^-1 * Xj)                        localization.cpp:263-281,588-602
 *   - RobustKernelCauchy(delta=1), quadratic form with rho'                         localization.cpp:622-624
 *   - SparseOptimizer::initializeOptimization + optimize(n) with
 *     OptimizationAlgorithmLevenberg + exact linear solve                          localization.cpp:44-52,164-170
 *   - OptimizableGraph::chi2()                                                     localization.cpp:197
 */
#ifndef G2O_GRAPH_ORACLE_H
#define G2O_GRAPH_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct og_graph og_graph;

/* Jacobian mode for range edges: the reference inherits g2o's numeric central difference (delta = 1e-9)
 * because EdgeSE3Range has no linearizeOplus (types_edge_se3range.h:45-74). */
enum { OG_JAC_NUMERIC_G2O = 0, OG_JAC_ANALYTIC = 1 };

typedef struct og_stats {
    int outer_iterations;   /* iterations of SparseOptimizer::optimize actually run */
    int lm_trials;          /* total LM trials (linear solves) */
    int terminated;         /* 1 if the LM returned Terminate (10 failed trials or rho == 0) */
    double lambda;          /* final damping */
    double robust_chi2;     /* activeRobustChi2 at the accepted state */
} og_stats;

og_graph* og_create(void);
void og_destroy(og_graph* g);

/* R row-major 3x3, t 3-vector. Returns 0 or a negative error. */
int og_add_vertex(og_graph* g, int id, const double* R, const double* t, int fixed);
int og_remove_vertex(og_graph* g, int id); /* also removes incident edges (g2o removeVertex) */
int og_has_vertex(og_graph* g, int id);
int og_set_estimate(og_graph* g, int id, const double* R, const double* t);
int og_get_estimate(og_graph* g, int id, double* R, double* t);

/* off0/off1 may be NULL (identity offsets). robust != 0 attaches RobustKernelCauchy(delta = 1). */
int og_add_range_edge(og_graph* g, int id0, int id1, double meas, double info,
                      const double* off0, const double* off1, int robust);
int og_add_prior_edge(og_graph* g, int id, const double* Rm, const double* tm, const double* info36);
int og_add_se3_edge(og_graph* g, int id0, int id1, const double* Rm, const double* tm,
                    const double* info36, int robust);

int og_num_vertices(og_graph* g);
int og_num_edges(og_graph* g);

/* initializeOptimization() + optimize(iterations). Returns iterations run, or <0 on error. */
int og_optimize(og_graph* g, int iterations, int jac_mode, og_stats* stats);

/* OptimizableGraph::chi2(): sum over all edges of e^T Omega e with each edge's last computed error. */
double og_chi2(og_graph* g);

/* helpers exposed for known-answer tests */
/* error (dim 1 or 6) and Jacobians (dim x 6 row-major per endpoint) of the edge_index-th live edge (creation order) */
int og_debug_linearize(og_graph* g, int edge_index, int jac_mode, double* err6, double* J0_36, double* J1_36);
void og_quat_to_R(const double* q_wxyz, double* R);          /* Eigen toRotationMatrix, no normalisation */
void og_R_to_quat(const double* R, double* q_wxyz);          /* Eigen Quaternion(R) */
void og_from_vector_mqt(const double* v6, double* R, double* t);
void og_to_vector_mqt(const double* R, const double* t, double* v6);
double og_cauchy_rho(double chi2, double* rho1);

/* Batched 3-DoF snapshot driven through the general graph (one moving vertex, M fixed anchors,
 * M Cauchy range edges; rebuilt per update).  Layout: dist/err [K][M][B] float, pos [3][B] double
 * (in: prior, out: last), out_pos [K][3][B], out_chi2 [K][B].  gate <= 0 disables the outlier gate;
 * epochs k < gate_from_epoch are un-gated (the reference gates only after warm-up, localization.cpp:309). */
int og_snapshot_batch(int B, int K, int M, const double* anchors /*[M][3]*/,
                      const float* dist, const float* err, double* pos,
                      double* out_pos, double* out_chi2, unsigned char* out_trials,
                      int iterations, double gate, int gate_from_epoch, int jac_mode);

/* BASELINE config 3 through the general graph: per tag and epoch one 6-DoF vertex whose rotation is overwritten by the
 * IMU quaternion (localization.cpp:505-513), a rotation-only EdgeSE3Prior with information 1/cov (:515-525), M range
 * edges with the antenna lever arm on endpoint 0 (:331-336), the outlier gate on vertex origins (:306-313).
 * Layouts: dist/err [K][M][B] float; imu [K][B][8] double (q xyzw, cov c0 c4 c8, pad); pose [7][B] double
 * (t xyz, q xyzw; in: prior, out: last); out_pose [K][7][B]; out_chi2 [K][B]. */
int og_fusion_batch(int B, int K, int M, const double* anchors, const double* offset_xyz, const float* dist,
                    const float* err, const double* imu, double* pose, double* out_pose, double* out_chi2,
                    unsigned char* out_trials, int iterations, double gate, int gate_from_epoch, int jac_mode);

#ifdef __cplusplus
}
#endif
#endif
