"""GPU parity of the sliding-window graph kernel (loc_window_*) vs the CPU oracle's general graph, same graphs.

fp64 both sides.  Same Jacobian mode on both sides:
  analytic vs analytic: 1e-7 m / 1e-7 rad on every pose after the reference's 10 iterations, 1e-9 median — same reasoning
    as the snapshot tests (an LM accept/reject decided at the rounding edge can move an iterate by the converged step size);
  numeric vs numeric (g2o's central differences, delta = 1e-9 — the reference's configuration,
    types_edge_se3range.h:45-74): 1e-5 m max, 1e-7 median: the difference quotient multiplies every last-bit difference of
    the two implementations by 5e8.
The kernel orders the poses itself (minimum degree) and factors level by level; the oracle factors in vertex-id order.
"""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

pytestmark = pytest.mark.gpu

ANCH = np.array([[3, -3, 0.58], [3, 3, 1.97], [-3, 3, 0.54], [-3, -3, 1.76]], dtype=float)


def _random_window(rng, T, with_imu, with_pose_edges, lever):
    """One instance in the reference's topology: per pose one anchor range (+ lever arm), a zero-range smoothness edge
    to the previous pose, optionally an IMU rotation prior and key-frame EdgeSE3 factors."""
    truth_t = np.cumsum(rng.normal(0, 0.05, (T, 3)), axis=0) + np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1.5, 1.5), 1.1])
    truth_R = Rotation.from_rotvec(np.cumsum(rng.normal(0, 0.03, (T, 3)), axis=0) + rng.normal(0, 0.3, 3))
    est_t = truth_t + rng.normal(0, 0.05, (T, 3))
    est_R = truth_R * Rotation.from_rotvec(rng.normal(0, 0.02, (T, 3)))
    off = np.array([0.1, 0.0, -0.05]) if lever else np.zeros(3)
    ranges, smooth, priors, se3 = [], [], [], []
    for k in range(T):
        # one range per pose as in the reference's stream; a lone pose gets all four anchors (snapshot shape)
        for a in ([int(rng.integers(0, 4))] if T > 1 else [0, 1, 2, 3]):
            d = np.linalg.norm(truth_t[k] + truth_R[k].apply(off) - ANCH[a]) + rng.normal(0, 0.03)
            ranges.append((k, a, float(np.float32(d)), 1.0 / 0.055 ** 2))
        if k > 0:
            smooth.append((k - 1, k, 0.0, 1.0 / (5.0 * (1 / 32) / 3) ** 2))
        if with_imu:
            priors.append((k, est_t[k].copy(), (truth_R[k] * Rotation.from_rotvec(rng.normal(0, 2e-3, 3))).as_matrix(),
                           np.array([0, 0, 0, 1, 1, 1.0]) / 4.592449e-06))
    if with_pose_edges:
        for k in range(1, T):
            key = (k // 4) * 4 if k % 4 else max(k - 4, 0)
            if key == k:
                continue
            Zt = truth_R[key].inv().apply(truth_t[k] - truth_t[key]) + rng.normal(0, 0.01, 3)
            ZR = (truth_R[key].inv() * truth_R[k] * Rotation.from_rotvec(rng.normal(0, 0.01, 3))).as_matrix()
            A = rng.normal(size=(6, 6)); info = A @ A.T + 6 * np.eye(6); info *= 1e3 / np.trace(info)
            se3.append((key, k, Zt, ZR, info))
    return est_t, est_R.as_matrix(), off, ranges, smooth, priors, se3


def _solve_both(gpu, B, T, iters, seed, with_imu, with_pose_edges, lever, jac="analytic", natural=False):
    import localization_amd as la
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    nr_max, np_max, ns_max = max(2 * T, 4), (T if with_imu else 0), (T if with_pose_edges else 0)
    wb = la.WindowBatch(B, T, nr_max, np_max, ns_max)
    want_t = np.zeros((B, T, 3)); want_R = np.zeros((B, T, 3, 3)); want_chi = np.zeros(B); want_trials = np.zeros(B)
    for i in range(B):
        est_t, est_R, off, ranges, smooth, priors, se3 = _random_window(rng, T, with_imu, with_pose_edges, lever)
        g = O.Graph()
        for m, a in enumerate(ANCH): g.add_vertex(m, a, fixed=True)
        for k in range(T):
            g.add_vertex(100 + k, est_t[k], est_R[k])
            assert wb.add_pose(i, est_t[k], est_R[k]) == k
        for (k, a, d, info) in ranges:
            g.add_range_edge(100 + k, a, d, info, off0=off); wb.add_range(i, k, a, d, info, off, anchor=True)
        for (k0, k1, d, info) in smooth:
            g.add_range_edge(100 + k0, 100 + k1, d, info); wb.add_range(i, k0, k1, d, info)
        for (k, t, R, dg) in priors:
            g.add_prior_edge(100 + k, t, R, np.diag(dg)); wb.add_prior(i, k, t, R, dg)
        for (ki, kj, t, R, info) in se3:
            g.add_se3_edge(100 + ki, 100 + kj, t, R, info, robust=True); wb.add_se3(i, ki, kj, t, R, info, True)
        n, st = g.optimize(iters, O.JAC_ANALYTIC if jac == "analytic" else O.JAC_NUMERIC_G2O)
        for k in range(T):
            want_R[i, k], want_t[i, k] = g.estimate(100 + k)
        want_chi[i] = g.chi2(); want_trials[i] = st.lm_trials
    solver = la.WindowSolver(ANCH, B, T, nr_max, np_max, ns_max, maximum_iteration=iters, jacobian=jac, natural_order=natural)
    res = solver.solve(wb)
    ms = solver.last_kernel_ms()
    solver.close()
    got_t = wb.poses[:, :, 9:]
    got_R = wb.poses[:, :, :9].reshape(B, T, 3, 3)
    return got_t, got_R, res, want_t, want_R, want_chi, want_trials, ms


@pytest.mark.parametrize("T,with_imu,with_pose_edges,lever", [
    (10, False, False, False),   # cfg/uwb_only.yaml shape: T = 10, ranges + smoothness
    (12, True, False, True),     # cfg/uwb_imu.yaml shape: T = 12, IMU rotation priors, antenna lever arm
    (8, False, True, False),     # pose factors around key frames (addPoseEdge)
    (16, True, True, True),      # everything at the capacity limit (96 unknowns)
    (1, True, False, True),      # one pose: BASELINE config 3's snapshot shape through the general kernel
    (24, True, False, True),     # beyond 16 poses the matrix lives in the HBM workspace (two rows per lane)
    (64, False, True, False),    # BASELINE config 5's shape: 64-pose window, key-frame pose factors (seven rows per lane)
])
def test_window_matches_oracle(gpu, T, with_imu, with_pose_edges, lever):
    B = 96 if T <= 16 else 12
    got_t, got_R, res, want_t, want_R, want_chi, want_trials, ms = _solve_both(gpu, B, T, 10, 7 * T + 1, with_imu,
                                                                               with_pose_edges, lever)
    dt = np.abs(got_t - want_t)
    dR = np.abs(got_R - want_R)
    assert np.isfinite(got_t).all() and np.isfinite(got_R).all()
    # result[6]: binary edges that share their pair of poses with another edge (their blocks are accumulated edge by edge).
    # A key-frame pose edge onto the previous pose shares its pair with the smoothness edge; without pose edges: none.
    assert ((res[:, 6] > 0) == (with_pose_edges and T > 1)).all()
    assert dt.max() < 1e-7 and np.median(dt) < 1e-9, (dt.max(), np.median(dt))
    assert dR.max() < 1e-7, dR.max()
    assert np.abs(res[:, 0] - want_chi).max() <= 1e-6 * max(1.0, np.abs(want_chi).max())
    # trial counts agree while the iteration is still converging; a lone well-observed pose converges in ~4
    # iterations and the remaining accept/reject decisions are taken on chi differences at the rounding level
    if T > 1:
        assert (res[:, 4] != want_trials).mean() < 0.05


@pytest.mark.parametrize("T,with_imu,with_pose_edges,lever", [
    (10, False, False, False),   # cfg/uwb_only.yaml shape (BASELINE config 1's window)
    (12, True, False, True),     # cfg/uwb_imu.yaml shape
    (64, False, True, False),    # BASELINE config 5's shape
    (24, True, False, True),     # HBM-workspace mode
])
def test_window_numeric_jacobian_matches_numeric_oracle(gpu, T, with_imu, with_pose_edges, lever):
    """The reference's own configuration on both sides: g2o's numeric range Jacobians in the kernel and in the oracle."""
    B = 64 if T <= 16 else 12
    got_t, got_R, res, want_t, want_R, want_chi, want_trials, ms = _solve_both(gpu, B, T, 10, 11 * T + 3, with_imu,
                                                                               with_pose_edges, lever, jac="numeric")
    dt = np.abs(got_t - want_t)
    assert np.isfinite(got_t).all() and np.isfinite(got_R).all()
    assert dt.max() < 1e-5 and np.median(dt) < 1e-7, (dt.max(), np.median(dt))
    assert np.abs(got_R - want_R).max() < 1e-5
    assert np.abs(res[:, 0] - want_chi).max() <= 1e-4 * max(1.0, np.abs(want_chi).max())


def test_window_analytic_kernel_vs_numeric_oracle_cfg5(gpu):
    """The default (analytic) kernel against the reference's configuration (numeric oracle) on BASELINE config 5's shape:
    the two Jacobians are the same derivative up to 1e-7 relative noise; on this shape the poses agree to better than 1e-6 m
    after the reference's 10 iterations (tests/perf/bench_window.py prints ~7e-8)."""
    import os
    import sys
    import localization_amd as la
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "perf"))
    import bench_window as bw
    B = 16
    wb, graphs, anchors, T = bw.build_pose64(B, np.random.default_rng(5))
    solver = la.WindowSolver(anchors, B, *wb.caps, maximum_iteration=10, bw_max=8, jacobian="analytic")
    solver.solve(wb)
    solver.close()
    _, want = bw.oracle_time(graphs, anchors, T, B)
    d = np.abs(wb.poses[:, :, 9:] - want)
    assert d.max() < 1e-6, d.max()


def test_elimination_order_is_transparent(gpu):
    """The in-kernel minimum-degree ordering changes the order of the eliminations (round-off), nothing else: the natural
    order gives the same poses to 1e-9, and the structure the kernel reports is what the graph theory says — BASELINE
    config 5's key-frame tree factors without fill (64 diagonal + 63 off-diagonal blocks) in 6 levels (the leaves, then the chain of 8 keys from both ends) instead of 64
    sequential block columns, a 10-pose chain is eaten from both ends (6 levels)."""
    import os
    import sys
    import localization_amd as la
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "perf"))
    import bench_window as bw
    B = 8
    wb, graphs, anchors, T = bw.build_pose64(B, np.random.default_rng(6))
    poses0 = wb.poses.copy()
    out = {}
    for natural in (False, True):
        wb.poses[:] = poses0
        solver = la.WindowSolver(anchors, B, *wb.caps, maximum_iteration=10, bw_max=8, natural_order=natural, jacobian="analytic")
        res = solver.solve(wb).copy()
        solver.close()
        out[natural] = (wb.poses.copy(), res)
    assert np.abs(out[False][0] - out[True][0]).max() < 1e-9
    lev, blocks = out[False][1][:, 7] // 65536, out[False][1][:, 7] % 65536
    assert (blocks == 64 + 63).all() and (lev == 6).all(), (lev, blocks)
    assert (out[True][1][:, 7] % 65536 > 300).all()          # the caller's (time) order fills the 8-pose band
    wb2, _, anch2, T2 = bw.build(4, "uwb_only")
    s2 = la.WindowSolver(anch2, 4, *wb2.caps, maximum_iteration=10, bw_max=1, jacobian="analytic", chain_threshold=0)   # (the general kernel: small translation-only chains take wave3_lm_kernel otherwise)
    r2 = s2.solve(wb2)
    s2.close()
    assert (r2[:, 7] % 65536 == 19).all() and (r2[:, 7] // 65536 == 6).all(), r2[:, 7]


def test_singular_system_fails_like_g2o(gpu):
    """H + lambda I not positive definite: every range has zero information, so H = 0, lambda_0 = 1e-5 * max diag = 0 and
    the Cholesky fails in every trial.  g2o (and the oracle) score each trial tempChi = max double, pop the (stale = zero)
    step, multiply lambda by nu and return Terminate after maxTrialsAfterFailure = 10: poses untouched, 10 trials, 1 outer
    iteration, terminated."""
    import localization_amd as la
    from oracle import oracle as O
    wb = la.WindowBatch(2, 3, 6, 0, 0)
    g = O.Graph()
    for m, a in enumerate(ANCH): g.add_vertex(m, a, fixed=True)
    for k in range(3):
        t = np.array([0.3 * k, -0.2, 1.0 + 0.1 * k])
        g.add_vertex(100 + k, t)
        for i in range(2): wb.add_pose(i, t)
        g.add_range_edge(100 + k, k, 2.5, 0.0)
        for i in range(2): wb.add_range(i, k, k, 2.5, 0.0, anchor=True)
    before = wb.poses.copy()
    n, st = g.optimize(10, O.JAC_ANALYTIC)
    solver = la.WindowSolver(ANCH, 2, 3, 6, 0, 0, jacobian="analytic")
    res = solver.solve(wb)
    solver.close()
    assert st.terminated == 1 and st.lm_trials == 10 and n == 1
    assert (res[:, 5] == 1).all() and (res[:, 4] == 10).all() and (res[:, 3] == 1).all()
    assert np.array_equal(wb.poses, before)
    for k in range(3):
        assert np.array_equal(g.estimate(100 + k)[1], before[0, k, 9:])
    assert (res[:, 0] == g.chi2()).all()


def test_selfcalibration_cfg4_real_shape(gpu):
    """BASELINE config 4 at its stated per-hypothesis shape: 256 timesteps x 10 unknown anchors (1596 unknowns, 2815 edges
    + 10 priors per hypothesis), B = 4 hypotheses, against the oracle's dense solve — analytic both sides, and the
    reference's numeric configuration both sides."""
    import os
    import sys
    import localization_amd as la
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "perf"))
    import bench_window as bw
    from oracle import oracle as O
    rng = np.random.default_rng(4)
    T, A, B = 256, 10, 4
    wb, graphs, anchors, nv = bw.build_selfcal(B, rng, T=T, A=A)
    poses0 = wb.poses.copy()
    for jac, omode, tol in (("analytic", O.JAC_ANALYTIC, 1e-6), ("numeric", O.JAC_NUMERIC_G2O, 1e-4)):
        wb.poses[:] = poses0
        solver = la.WindowSolver(anchors, B, *wb.caps, maximum_iteration=10, bw_max=nv - 1, jacobian=jac)
        res = solver.solve(wb).copy()
        assert solver.last_kernel_kind() == "arrow3_lm_kernel"   # (translation-only chain + border of 10: arrow3_kernel.hip)
        solver.close()
        assert np.isfinite(wb.poses).all()
        worst = 0.0
        for i in range(B if jac == "analytic" else 2):
            g = graphs[i]
            G = O.Graph()
            for k in range(T): G.add_vertex(100 + k, g["et"][k])
            for a in range(A):
                G.add_vertex(100 + T + a, g["hyp"][a]); G.add_prior_edge(100 + T + a, g["hyp"][a], np.eye(3), np.diag([1.0, 1, 1, 0, 0, 0]))
            for (k, a, d) in g["ranges"]: G.add_range_edge(100 + k, 100 + T + a, d, 1 / 0.055 ** 2)
            for (k0, k1) in g["smooth"]: G.add_range_edge(100 + k0, 100 + k1, 0.0, 1 / (5.0 / 32 / 3) ** 2)
            G.optimize(10, omode)
            want = np.array([G.estimate(100 + k)[1] for k in range(T + A)])
            worst = max(worst, np.abs(wb.poses[i, :, 9:] - want).max())
            assert abs(res[i, 0] - G.chi2()) <= 1e-5 * max(1.0, G.chi2())
        assert worst < tol, (jac, worst)


def test_cfg5_full_batch_properties(gpu):
    """BASELINE config 5 at its full batch (16 384 windows of 64 poses): determinism, permutation equivariance over the
    instances, resident path = host path, and an oracle spot check on every 1024th window."""
    import os
    import sys
    import localization_amd as la
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "perf"))
    import bench_window as bw
    B = 16384
    wb, graphs, anchors, T = bw.build_pose64(B, np.random.default_rng(0), n_graphs=0)
    poses0 = wb.poses.copy()
    solver = la.WindowSolver(anchors, B, *wb.caps, maximum_iteration=10, bw_max=8, jacobian="analytic")
    res1 = solver.solve(wb).copy(); p1 = wb.poses.copy()
    wb.poses[:] = poses0
    solver.upload(wb)
    solver.solve_resident(); solver.solve_resident()
    solver.download(wb)
    assert np.array_equal(wb.poses, p1) and np.array_equal(wb.result, res1)           # deterministic, resident = host path
    perm = np.random.default_rng(1).permutation(B)
    wbp = la.WindowBatch(B, *wb.caps)
    for name in ("counts", "r_idx", "r_val", "p_idx", "p_val", "s_idx", "s_val"):
        getattr(wbp, name)[:] = getattr(wb, name)[perm]
    wbp.poses[:] = poses0[perm]
    solver.solve(wbp)
    solver.close()
    assert np.array_equal(wbp.poses, p1[perm]) and np.array_equal(wbp.result, res1[perm])   # instances are independent
    assert np.isfinite(p1).all() and (res1[:, 3] == 10).all()
    from oracle import oracle as O
    worst = 0.0
    for i in range(0, B, 1024):
        G = O.Graph()
        for m, a in enumerate(anchors): G.add_vertex(m, a, fixed=True)
        for k in range(T): G.add_vertex(100 + k, poses0[i, k, 9:], poses0[i, k, :9].reshape(3, 3))
        for e in range(T):
            G.add_range_edge(100 + int(wb.r_idx[i, e, 0]), -1 - int(wb.r_idx[i, e, 1]), wb.r_val[i, e, 0], wb.r_val[i, e, 1])
        for e in range(T - 1):
            Ri = wb.s_val[i, e, :9].reshape(3, 3); ti = wb.s_val[i, e, 9:12]
            G.add_se3_edge(100 + int(wb.s_idx[i, e, 0]), 100 + int(wb.s_idx[i, e, 1]), -Ri.T @ ti, Ri.T, wb.s_val[i, e, 12:].reshape(6, 6), True)
        G.optimize(10, O.JAC_ANALYTIC)
        want = np.array([G.estimate(100 + k)[1] for k in range(T)])
        worst = max(worst, np.abs(p1[i, :, 9:] - want).max())
    assert worst < 1e-7, worst


def test_selfcalibration_arrowhead_graph(gpu):
    """BASELINE config 4's shape at test size: every node moves (topic/relative_range, localization.cpp:94-98) — A unknown
    anchors ranged from every pose of a tag trajectory, weak position priors on the anchor hypotheses.  Tag poses first,
    anchors last: the skyline is an arrowhead (long rows only for the anchors)."""
    import os
    import sys
    import localization_amd as la
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "perf"))
    import bench_window as bw
    rng = np.random.default_rng(3)
    T, A, B = 24, 4, 6
    wb, graphs, anchors, nv = bw.build_selfcal(B, rng, T=T, A=A)
    solver = la.WindowSolver(anchors, B, *wb.caps, maximum_iteration=10, jacobian="analytic")
    res = solver.solve(wb)
    solver.close()
    from oracle import oracle as O
    worst = 0.0
    for i in range(B):
        g = graphs[i]
        G = O.Graph()
        for k in range(T): G.add_vertex(100 + k, g["et"][k])
        for a in range(A):
            G.add_vertex(100 + T + a, g["hyp"][a]); G.add_prior_edge(100 + T + a, g["hyp"][a], np.eye(3), np.diag([1.0, 1, 1, 0, 0, 0]))
        for (k, a, d) in g["ranges"]: G.add_range_edge(100 + k, 100 + T + a, d, 1 / 0.055 ** 2)
        for (k0, k1) in g["smooth"]: G.add_range_edge(100 + k0, 100 + k1, 0.0, 1 / (5.0 / 32 / 3) ** 2)
        G.optimize(10, O.JAC_ANALYTIC)
        want = np.array([G.estimate(100 + k)[1] for k in range(T + A)])
        worst = max(worst, np.abs(wb.poses[i, :, 9:] - want).max())
        assert abs(res[i, 0] - G.chi2()) <= 1e-6 * max(1.0, G.chi2())
    assert worst < 1e-6, worst


def test_window_converged_matches_oracle_tightly(gpu):
    got_t, got_R, res, want_t, want_R, _, _, _ = _solve_both(gpu, 32, 10, 60, 99, True, False, True)
    assert np.abs(got_t - want_t).max() < 1e-7


def test_window_empty_and_partial_instances(gpu):
    """Instances with no edges (or no poses) come back untouched; capacities larger than what is used are fine."""
    import localization_amd as la
    wb = la.WindowBatch(4, 6, 12, 4, 4)
    wb.add_pose(1, [1.0, 2.0, 3.0])                       # pose without any edge
    v = wb.add_pose(2, [0.2, 0.1, 1.0]); wb.add_range(2, v, 0, 3.0, 100.0, anchor=True)
    before = wb.poses.copy()
    solver = la.WindowSolver(ANCH, 4, 6, 12, 4, 4, jacobian="analytic")
    res = solver.solve(wb)
    solver.close()
    assert np.array_equal(wb.poses[0], before[0]) and np.array_equal(wb.poses[1], before[1]) and np.array_equal(wb.poses[3], before[3])
    assert res[0, 4] == 0 and res[1, 4] == 0
    assert np.linalg.norm(wb.poses[2, 0, 9:] - ANCH[0]) == pytest.approx(3.0, abs=1e-6)


def test_window_rejects_bad_indices(gpu):
    import localization_amd as la
    wb = la.WindowBatch(1, 4, 4, 0, 0)
    wb.add_pose(0, [0, 0, 1.0])
    wb.add_range(0, 0, 9, 1.0, 1.0, anchor=True)          # anchor 9 does not exist
    solver = la.WindowSolver(ANCH, 1, 4, 4, 0, 0, jacobian="analytic")
    with pytest.raises(la.LocalizationAmdError):
        solver.solve(wb)
    solver.close()


def test_several_edges_on_one_pair_of_poses(gpu):
    """A smoothness range AND an EdgeSE3 (and, in half the instances, a second range) between the same two poses — what the
    reference produces when a twist edge and the key-frame pose edge meet (localization.cpp:276-281, 446-450).  Their
    off-diagonal block then has more than one contributor: the kernel finds those pairs once per solve and accumulates
    their blocks edge by edge; the result matches the oracle like every other graph."""
    import localization_amd as la
    from oracle import oracle as O
    rng = np.random.default_rng(11)
    B, T = 32, 6
    wb = la.WindowBatch(B, T, 3 * T, 0, T)
    want_t = np.zeros((B, T, 3))
    for i in range(B):
        truth_t = np.cumsum(rng.normal(0, 0.05, (T, 3)), axis=0) + np.array([0.3, -0.4, 1.1])
        truth_R = Rotation.from_rotvec(np.cumsum(rng.normal(0, 0.03, (T, 3)), axis=0))
        est_t = truth_t + rng.normal(0, 0.05, (T, 3)); est_R = (truth_R * Rotation.from_rotvec(rng.normal(0, 0.02, (T, 3)))).as_matrix()
        g = O.Graph()
        for m, a in enumerate(ANCH): g.add_vertex(m, a, fixed=True)
        for k in range(T):
            g.add_vertex(100 + k, est_t[k], est_R[k]); wb.add_pose(i, est_t[k], est_R[k])
        for k in range(T):
            a = int(rng.integers(0, 4)); d = float(np.float32(np.linalg.norm(truth_t[k] - ANCH[a]) + rng.normal(0, 0.03)))
            g.add_range_edge(100 + k, a, d, 1 / 0.055 ** 2); wb.add_range(i, k, a, d, 1 / 0.055 ** 2, anchor=True)
            if k:
                info_s = 1.0 / (5.0 / 32 / 3) ** 2
                g.add_range_edge(100 + k - 1, 100 + k, 0.0, info_s); wb.add_range(i, k - 1, k, 0.0, info_s)
                if i % 2:   # a measured peer-to-peer range on the same pair as well
                    dd = float(np.linalg.norm(truth_t[k] - truth_t[k - 1]) + rng.normal(0, 0.01))
                    g.add_range_edge(100 + k, 100 + k - 1, dd, 1 / 0.03 ** 2); wb.add_range(i, k, k - 1, dd, 1 / 0.03 ** 2)
                Zt = truth_R[k - 1].inv().apply(truth_t[k] - truth_t[k - 1]) + rng.normal(0, 0.01, 3)
                ZR = (truth_R[k - 1].inv() * truth_R[k] * Rotation.from_rotvec(rng.normal(0, 0.01, 3))).as_matrix()
                info = np.eye(6) * 2e3
                g.add_se3_edge(100 + k - 1, 100 + k, Zt, ZR, info, robust=True); wb.add_se3(i, k - 1, k, Zt, ZR, info, True)
        g.optimize(10, O.JAC_ANALYTIC)
        for k in range(T):
            want_t[i, k] = g.estimate(100 + k)[1]
    solver = la.WindowSolver(ANCH, B, *wb.caps, maximum_iteration=10, bw_max=1, jacobian="analytic")
    res = solver.solve(wb)
    solver.close()
    assert (res[::2, 6] == 2 * (T - 1)).all() and (res[1::2, 6] == 3 * (T - 1)).all()   # every binary edge is on a shared pair
    d = np.abs(wb.poses[:, :, 9:] - want_t)
    assert np.isfinite(wb.poses).all() and d.max() < 1e-7 and np.median(d) < 1e-9, (d.max(), np.median(d))


@pytest.mark.parametrize("jac", ["analytic", "numeric"])
def test_wide_windows_ragged_batch(gpu, jac):
    """Windows of 65 .. 512 poses run on eight waves each (set-up on wave 0, LM loop on all 512 threads, block-wide
    reductions): a batch sized for 96 poses whose instances use 0, 1, 1, 5, 70 and 96 of them — no pose at all, a pose
    without edges, the snapshot shape, small and full windows with IMU priors, key-frame pose factors and a lever arm — and
    one whose every range has zero information (the failed-Cholesky path with all waves)."""
    import localization_amd as la
    from _oracle_window import oracle_solve_instance
    rng = np.random.default_rng(4242)
    sizes = [0, 1, 1, 5, 70, 96, 3]
    B, T = len(sizes), 96
    wb = la.WindowBatch(B, T, 2 * T, T, T)
    for i, Ti in enumerate(sizes):
        if Ti == 0:
            continue
        if i == 1:
            wb.add_pose(i, [1.0, 2.0, 3.0])
            continue
        est_t, est_R, off, ranges, smooth, priors, se3 = _random_window(rng, Ti, True, Ti > 1, True)
        for k in range(Ti):
            assert wb.add_pose(i, est_t[k], est_R[k]) == k
        singular = i == 6
        for (k, a, d, info) in ranges: wb.add_range(i, k, a, d, 0.0 if singular else info, off, anchor=True)
        if singular:
            continue
        for (k0, k1, d, info) in smooth: wb.add_range(i, k0, k1, d, info)
        for (k, t, R, dg) in priors: wb.add_prior(i, k, t, R, dg)
        for (ki, kj, t, R, info) in se3: wb.add_se3(i, ki, kj, t, R, info, True)
    before = wb.poses.copy()
    from oracle import oracle as O
    want = [oracle_solve_instance(wb, i, ANCH, jac_mode=O.JAC_ANALYTIC if jac == "analytic" else O.JAC_NUMERIC_G2O) for i in range(B)]
    solver = la.WindowSolver(ANCH, B, T, 2 * T, T, T, jacobian=jac)
    res = solver.solve(wb)
    solver.close()
    assert np.array_equal(wb.poses[0], before[0]) and np.array_equal(wb.poses[1], before[1]) and res[0, 4] == 0 and res[1, 4] == 0
    assert np.array_equal(wb.poses[6], before[6]) and res[6, 5] == 1 and res[6, 4] == 10 and res[6, 3] == 1
    for i in (2, 3, 4, 5):
        poses, chi, st = want[i]
        d = np.abs(wb.poses[i, :sizes[i]] - poses).max()
        # (numeric vs numeric: the central differences amplify the last bits of the summation order, as in the other numeric tests)
        assert d < (1e-7 if jac == "analytic" else 1e-5), (i, d)
        assert abs(res[i, 0] - chi) <= (1e-6 if jac == "analytic" else 1e-4) * max(1.0, abs(chi)), (i, res[i, 0], chi)


@pytest.mark.parametrize("T,with_imu,lever,jac,twist", [
    (10, False, False, "analytic", False),   # cfg/uwb_only.yaml's window
    (12, True, True, "analytic", False),     # cfg/uwb_imu.yaml's window: IMU priors, lever arm
    (10, False, False, "numeric", False),    # the reference's Jacobian mode
    (1, True, True, "analytic", False),      # a lone pose
    (16, True, True, "numeric", False),
    (40, False, True, "analytic", False),    # beyond 32 poses the coupling blocks are always stored in full
    (10, False, True, "analytic", True),     # cfg/uwb_twist.yaml's window: an EdgeSE3 between consecutive poses (addTwistEdge)
    (12, True, True, "numeric", True),
])
def test_chain_kernel_matches_oracle(gpu, T, with_imu, lever, jac, twist):
    """Large batches of chain windows run one lane per window (chain_lm_kernel: block-tridiagonal Cholesky in pose order,
    the state in an [entry][lane] HBM workspace).  Forced here for a small batch through loc_window_set_chain_threshold:
    against the oracle, against the wave-per-window kernel on the same batch, and through the resident API."""
    import localization_amd as la
    from oracle import oracle as O
    from _oracle_window import oracle_solve_instance
    B = 70   # (one full wave of windows + a partial one)
    rng = np.random.default_rng(100 * T + len(jac))
    nr_max, np_max, ns_max = max(2 * T + 2, 4), (T if with_imu else 0), (T if twist else 0)
    wb = la.WindowBatch(B, T, nr_max, np_max, ns_max)
    for i in range(B):
        Ti = T if i % 7 else max(T // 2, 1)   # ragged lengths inside the wave
        est_t, est_R, off, ranges, smooth, priors, _ = _random_window(rng, Ti, with_imu, False, lever)
        if twist:   # odometry-like relative poses between consecutive poses, every other one stored the other way round
            for k in range(1, Ti):
                a, b = (k - 1, k) if k % 2 else (k, k - 1)
                Ra, Rb = Rotation.from_matrix(est_R[a]), Rotation.from_matrix(est_R[b])
                Zt = Ra.inv().apply(est_t[b] - est_t[a]) + rng.normal(0, 0.01, 3)
                ZR = (Ra.inv() * Rb * Rotation.from_rotvec(rng.normal(0, 0.01, 3))).as_matrix()
                A = rng.normal(size=(6, 6)); info = A @ A.T + 6 * np.eye(6); info *= 1e3 / np.trace(info)
                wb.add_se3(i, a, b, Zt, ZR, info, bool(k % 3))
        for k in range(Ti):
            wb.add_pose(i, est_t[k], est_R[k])
        # (the reference's creation order: a pose's anchor range, then its smoothness edge to the previous pose)
        for k in range(Ti):
            for (kk, a, d, info) in ranges:
                if kk == k: wb.add_range(i, k, a, d, info, off, anchor=True)
            for (k0, k1, d, info) in smooth:
                if k1 == k and not (i % 5 == 2 and k == 3):   # (some windows miss a link: two independent chains)
                    wb.add_range(i, k0, k1, d, info)
                    if i % 5 == 1 and k == 2: wb.add_range(i, k1, k0, 0.02, 0.5 * info, off)   # (and some have two edges on one pair, the second the other way round with a lever arm)
        for (k, t, R, dg) in priors: wb.add_prior(i, k, t, R, dg)
    wb.counts[3, 1:] = 0   # an instance whose poses have no edge at all: comes back untouched
    before = wb.poses.copy()
    mode = O.JAC_ANALYTIC if jac == "analytic" else O.JAC_NUMERIC_G2O
    want = [oracle_solve_instance(wb, i, ANCH, jac_mode=mode) for i in range(B)]
    ref = la.WindowBatch(B, T, nr_max, np_max, ns_max)
    for name in ("counts", "poses", "r_idx", "r_val", "p_idx", "p_val", "s_idx", "s_val"):
        getattr(ref, name)[:] = getattr(wb, name)
    general = la.WindowSolver(ANCH, B, T, nr_max, np_max, ns_max, jacobian=jac, chain_threshold=0)
    res_general = general.solve(ref).copy()
    general.close()
    chain = la.WindowSolver(ANCH, B, T, nr_max, np_max, ns_max, jacobian=jac, chain_threshold=1)
    res = chain.solve(wb).copy()
    assert (res[res[:, 3] > 0, 7] % 65536 == 2 * (res[res[:, 3] > 0, 7] // 65536) - 1).all()   # (the chain kernel's signature: n levels, 2 n - 1 blocks)
    # (numeric vs numeric: the central differences of the near-zero ranges between consecutive poses — some of them doubled here —
    #  amplify the last bits of a different summation order; the analytic mode holds 1e-7)
    tol = 1e-7 if jac == "analytic" else 3e-5
    for i in range(B):
        nv = int(wb.counts[i, 0])
        if nv == 0 or wb.counts[i, 1] + wb.counts[i, 2] + wb.counts[i, 3] == 0:
            assert np.array_equal(wb.poses[i], before[i]) and res[i, 4] == 0
            continue
        poses, chi, st = want[i]
        d = np.abs(wb.poses[i, :nv] - poses).max()
        assert d < tol, (i, d)
        assert abs(res[i, 0] - chi) <= (1e-6 if jac == "analytic" else 1e-4) * max(1.0, abs(chi)), (i, res[i, 0], chi)
    assert np.abs(wb.poses - ref.poses).max() < tol
    if T > 1:   # (a lone well-observed pose converges early: the remaining decisions are taken on rounding-level chi differences)
        assert (res[:, 4] != res_general[:, 4]).mean() < 0.05   # LM trial counts
    # resident API: the same answer again
    wb2 = la.WindowBatch(B, T, nr_max, np_max, ns_max)
    for name in ("counts", "r_idx", "r_val", "p_idx", "p_val", "s_idx", "s_val"):
        getattr(wb2, name)[:] = getattr(wb, name)
    wb2.poses[:] = before
    chain.upload(wb2)
    chain.solve_resident()
    chain.download(wb2)
    chain.close()
    for i in range(B):   # (pose slots beyond an instance's nv are not written by the resident path)
        nv = int(wb.counts[i, 0])
        assert np.array_equal(wb2.poses[i, :nv], wb.poses[i, :nv]), i
    assert np.array_equal(wb2.result, res)


def test_skyline_window_matches_oracle_and_chain_kernel_600_poses(gpu):
    """Windows of 513 .. 1024 poses take the envelope (skyline) factorisation in the caller's order: against the oracle (3600
    unknowns per window) and against the other independent implementation of the same problem, the one-lane-per-window
    block-tridiagonal kernel (forced by the threshold).  Same LM trajectory, same poses."""
    import localization_amd as la
    T, B = 600, 2
    rng = np.random.default_rng(600)
    nr_max = 2 * T
    wbs = []
    for rep in range(2):
        wb = la.WindowBatch(B, T, nr_max, 0, 0)
        wbs.append(wb)
    for i in range(B):
        est_t, est_R, off, ranges, smooth, priors, _ = _random_window(rng, T, False, False, True)
        for wb in wbs:
            for k in range(T):
                wb.add_pose(i, est_t[k], est_R[k])
            for k in range(T):
                for (kk, a, d, info) in ranges:
                    if kk == k: wb.add_range(i, k, a, d, info, off, anchor=True)
                for (k0, k1, d, info) in smooth:
                    if k1 == k: wb.add_range(i, k0, k1, d, info)
    before = wbs[0].poses.copy()
    sky = la.WindowSolver(ANCH, B, T, nr_max, 0, 0, bw_max=1, chain_threshold=0, jacobian="analytic")
    res_sky = sky.solve(wbs[0]).copy()
    sky.close()
    chain = la.WindowSolver(ANCH, B, T, nr_max, 0, 0, bw_max=1, chain_threshold=1, jacobian="analytic")
    res_chain = chain.solve(wbs[1]).copy()
    chain.close()
    assert (res_chain[:, 7] == T * 65536 + 2 * T - 1).all() and (res_sky[:, 7] == 0).all()   # (which kernel ran)
    assert np.isfinite(wbs[0].poses).all()
    assert np.abs(wbs[0].poses - wbs[1].poses).max() < 1e-7
    assert np.array_equal(res_sky[:, 3:6], res_chain[:, 3:6])                                # iterations, trials, terminated
    assert np.abs(res_sky[:, 0] - res_chain[:, 0]).max() <= 1e-6 * np.abs(res_sky[:, 0]).max()
    assert (res_sky[:, 1] < 0.5 * 1e9).all() and (res_sky[:, 4] >= 10).all()
    # ... and against the oracle itself (its envelope Cholesky takes ~0.2 s per 600-pose solve): the 513 .. 1024-pose path has its own
    # oracle check, not only the comparison with another kernel
    from _oracle_window import oracle_solve_instance
    fresh = la.WindowBatch(B, T, nr_max, 0, 0)
    for name in ("counts", "r_idx", "r_val", "p_idx", "p_val", "s_idx", "s_val"):
        getattr(fresh, name)[:] = getattr(wbs[0], name)
    fresh.poses[:] = before
    for i in range(B):
        poses, chi, st = oracle_solve_instance(fresh, i, ANCH)
        assert np.abs(wbs[0].poses[i] - poses).max() < 1e-7, (i, np.abs(wbs[0].poses[i] - poses).max())
        assert abs(res_sky[i, 0] - chi) <= 1e-6 * max(1.0, abs(chi)) and res_sky[i, 4] == st.lm_trials


def test_chain_kernel_failed_cholesky_like_g2o(gpu):
    """The lane-per-window kernel on a window whose every range has zero information (H = 0, lambda_0 = 0: the factorisation fails
    in every trial): 10 trials, 1 outer iteration, terminated, poses untouched — next to healthy windows in the same wave."""
    import localization_amd as la
    from _oracle_window import oracle_solve_instance
    T, B = 6, 5
    rng = np.random.default_rng(66)
    wb = la.WindowBatch(B, T, 2 * T, 0, 0)
    for i in range(B):
        est_t, est_R, off, ranges, smooth, _, _ = _random_window(rng, T, False, False, False)
        for k in range(T):
            wb.add_pose(i, est_t[k], est_R[k])
        for k in range(T):
            for (kk, a, d, info) in ranges:
                if kk == k: wb.add_range(i, k, a, d, 0.0 if i == 2 else info, off, anchor=True)
            if i != 2:
                for (k0, k1, d, info) in smooth:
                    if k1 == k: wb.add_range(i, k0, k1, d, info)
    before = wb.poses.copy()
    want = [oracle_solve_instance(wb, i, ANCH) for i in range(B)]
    s = la.WindowSolver(ANCH, B, T, 2 * T, 0, 0, chain_threshold=1, jacobian="analytic")
    res = s.solve(wb).copy()
    s.close()
    assert (res[[0, 1, 3, 4], 7] == T * 65536 + 2 * T - 1).all()
    assert res[2, 5] == 1 and res[2, 4] == 10 and res[2, 3] == 1 and np.array_equal(wb.poses[2], before[2])
    assert want[2][2].terminated == 1 and want[2][2].lm_trials == 10
    for i in (0, 1, 3, 4):
        assert np.abs(wb.poses[i] - want[i][0]).max() < 1e-7


@pytest.mark.parametrize("T,jac", [(8, "analytic"), (8, "numeric"), (24, "analytic"), (70, "numeric")])
def test_range_lever_arm_on_endpoint1_and_on_fixed_endpoints(gpu, T, jac):
    """EdgeSE3Range carries a lever arm per endpoint (Isometry3d offset[2], types_edge_se3range.h:73; setVertexOffset(int, ...),
    types_edge_se3range.cpp:99-103; both in the residual, :108-112).  The reference only ever sets endpoint 0's; endpoint 1's goes
    through loc_window_set_endpoint1_offsets: peer ranges between poses with antennas on BOTH ends, ranges to anchors whose own
    antenna sits off the surveyed point, in LDS (T = 8), workspace (24) and eight-wave (70) windows."""
    import localization_amd as la
    from oracle import oracle as O
    from _oracle_window import oracle_solve_instance
    B = 5
    rng = np.random.default_rng(900 + T + len(jac))
    wb = la.WindowBatch(B, T, 3 * T + 4, T, 0)
    for i in range(B):
        est_t, est_R, off, ranges, smooth, priors, _ = _random_window(rng, T, True, False, True)
        for k in range(T): wb.add_pose(i, est_t[k], est_R[k])
        for k in range(T):
            for (kk, a, d, info) in ranges:
                if kk == k: wb.add_range(i, k, a, d, info, off, anchor=True, off1=(0.05, -0.02, 0.1) if k % 2 else None)   # the anchor's own antenna
            for (k0, k1, d, info) in smooth:
                if k1 == k: wb.add_range(i, k0, k1, d, info)
            if k >= 3 and k % 3 == 0:    # a peer range between two poses of the window, antennas on both ends (either storage order)
                a, b = (k, k - 3) if k % 2 else (k - 3, k)
                wb.add_range(i, a, b, float(np.linalg.norm(est_t[a] - est_t[b]) + rng.normal(0, 0.03)), 1 / 0.055 ** 2, (0.1, 0.0, -0.05), off1=(-0.08, 0.03, 0.02))
        for (k, t, R, dg) in priors: wb.add_prior(i, k, t, R, dg)
    wb._initial = wb.poses.copy()
    mode = O.JAC_ANALYTIC if jac == "analytic" else O.JAC_NUMERIC_G2O
    want = [oracle_solve_instance(wb, i, ANCH, jac_mode=mode) for i in range(B)]
    zero = [oracle_solve_instance(_without_off1(la, wb), i, ANCH, jac_mode=mode)[0] for i in range(1)]
    s = la.WindowSolver(ANCH, B, *wb.caps, jacobian=jac, bw_max=3, chain_threshold=1)
    res = s.solve(wb).copy()
    assert s.last_kernel_kind() == "window_lm_kernel"
    tol = 1e-7 if jac == "analytic" else 1e-5
    for i in range(B):
        poses, chi, st = want[i]
        d = np.abs(wb.poses[i] - poses).max()
        assert d < tol, (i, d)
        assert abs(res[i, 0] - chi) <= (1e-6 if jac == "analytic" else 1e-4) * max(1.0, abs(chi))
    assert np.abs(want[0][0] - zero[0]).max() > 1e-3      # (the lever arms matter: without them the oracle lands elsewhere)
    # clearing them again: the next solve is the plain one
    plain = _without_off1(la, wb)
    s.solve(plain)
    assert np.abs(plain.poses[0] - zero[0]).max() < tol
    s.close()


def _without_off1(la, wb):
    out = la.WindowBatch(wb.B, *wb.caps)
    for name in ("counts", "r_idx", "r_val", "p_idx", "p_val", "s_idx", "s_val"):
        getattr(out, name)[:] = getattr(wb, name)
    out.poses[:] = wb._initial if hasattr(wb, "_initial") else wb.poses
    return out


def _chain_batch(la, B, T, seed):
    rng = np.random.default_rng(seed)
    wb = la.WindowBatch(B, T, 2 * T, 0, 0)
    for i in range(B):
        est_t, est_R, off, ranges, smooth, priors, se3 = _random_window(rng, T, False, False, False)
        for k in range(T):
            wb.add_pose(i, est_t[k])
            a = ranges[k]
            if k > 0:
                wb.add_range(i, k - 1, k, 0.0, smooth[k - 1][3])
            wb.add_range(i, k, a[1], a[2], a[3], anchor=True)
    return wb


def test_large_host_solve_drops_the_resident_batch(gpu):
    """include/localization_amd.h, "Mixing with loc_window_solve_host": a host solve too large for the staging block reuses the device
    arrays the resident batch lives in, so the resident batch is gone afterwards — loc_window_solve_resident / loc_window_download say
    LOC_ERR_INVALID until the next upload (before: the resident kernel chosen at upload ran on the other batch's data).  A small host
    solve in between leaves the resident batch alone."""
    import localization_amd as la
    from localization_amd import _lib
    B, T = 3000, 10                                        # 3000 x (960 + 800 + ...) B > 4 MiB: the large path
    big = _chain_batch(la, B, T, 5)
    small = _chain_batch(la, 2, T, 6)
    solver = la.WindowSolver(ANCH, B, T, 2 * T, 0, 0, maximum_iteration=10, jacobian="analytic", bw_max=1)
    ref = _chain_batch(la, B, T, 5)
    solver.solve(ref)                                     # what the batch solves to
    solver.upload(big)
    solver.solve_resident()
    keep = small.poses.copy()
    solver.solve(small)                                   # small: through the staging block
    assert solver.last_host_timing()[2] > 0.0
    assert not np.array_equal(small.poses, keep)
    out = _chain_batch(la, B, T, 5)
    solver.solve_resident()                               # the resident batch is still there
    solver.download(out)
    assert np.array_equal(out.poses, ref.poses)
    other = _chain_batch(la, B, T, 7)
    solver.solve(other)                                   # large: takes over the device arrays
    for call in (solver.solve_resident, lambda: solver.download(out)):
        with pytest.raises(la.LocalizationAmdError) as exc:
            call()
        assert exc.value.code == -1     # LOC_ERR_INVALID
    solver.upload(big)                                    # ... until the next upload
    solver.solve_resident()
    solver.download(out)
    assert np.array_equal(out.poses, ref.poses)
    solver.close()


def test_window_options_and_topology_cache(gpu):
    """loc_window_set_option replaces the environment switches (read once at create); the structural verdict of a host-path batch is
    cached on a hash of its counts / index tables: the same graph with new measurements skips the analysis, another structure does not;
    the result is the same with the cache off."""
    import localization_amd as la
    B, T = 600, 10
    def other_measurements():   # the graph of `a` replayed with other ranges and another start: same counts and index tables
        w = _chain_batch(la, B, T, 1)
        w.r_val[:, :, 0] += 0.01 * np.sin(np.arange(w.r_val.shape[1]))[None, :] * (w.r_val[:, :, 0] > 0)
        w.poses[:, :, 9:] += 0.02
        return w
    a, b = _chain_batch(la, B, T, 1), other_measurements()
    solver = la.WindowSolver(ANCH, B, T, 2 * T, 0, 0, maximum_iteration=10, jacobian="analytic", bw_max=1)
    with pytest.raises(la.LocalizationAmdError):
        solver.set_option("no_such_switch", 1)
    with pytest.raises(la.LocalizationAmdError):
        solver.set_option("wave3", 7)
    solver.solve(a)
    assert solver.last_kernel_kind() == "wave3_lm_kernel" and solver.last_host_timing()[3] is False
    solver.solve(b)                                        # same structure, other measurements: verdict from the cache
    t = solver.last_host_timing()
    assert t[3] is True and solver.last_kernel_kind() == "wave3_lm_kernel"
    cached = b.poses.copy()
    solver.set_option("topology_cache", 0)
    b2 = other_measurements()
    solver.solve(b2)
    assert solver.last_host_timing()[3] is False and np.array_equal(b2.poses, cached)
    solver.set_option("topology_cache", 1)
    solver.set_option("wave3", 0)                          # A/B switch of the handle (was: LOCAMD_WAVE3=0 read at every solve)
    b3 = other_measurements()
    solver.solve(b3)
    assert solver.last_kernel_kind() == "window_lm_kernel"
    assert np.abs(b3.poses - cached).max() < 1e-6
    solver.set_option("wave3", 1)
    solver.set_option("chain_min_batch", 0)                # "never anything but the general kernel"
    solver.solve(b3)
    assert solver.last_kernel_kind() == "window_lm_kernel"
    solver.set_option("chain_min_batch", -1)
    c = _chain_batch(la, B, T, 3)
    c.r_idx[5, 1, 1] = -1 - 2                               # another structure: instance 5's smoothness edge becomes an anchor range
    solver.solve(c)
    assert solver.last_host_timing()[3] is False
    solver.close()
    # "kernel_events" = 0 (what the node's own handle runs with): a handful of small windows solved from the staging block without HIP
    # events around the kernel — the same bits, and loc_window_last_kernel_ms reports launch to completion on the host clock
    small = _chain_batch(la, 2, T, 4)
    want = la.WindowBatch(2, *small.caps)
    for name in ("counts", "poses", "r_idx", "r_val", "p_idx", "p_val", "s_idx", "s_val"): getattr(want, name)[:] = getattr(small, name)
    s2 = la.WindowSolver(ANCH, 2, *small.caps, jacobian="numeric")
    r_with = s2.solve(want).copy()
    ms_with = s2.last_kernel_ms()
    s2.set_option("kernel_events", 0)
    r_without = s2.solve(small).copy()
    assert s2.last_kernel_kind() == "wave3_lm_kernel" and np.array_equal(small.poses, want.poses) and np.array_equal(r_with, r_without)
    assert 0.0 < ms_with < 5.0 and 0.0 < s2.last_kernel_ms() < 5.0
    s2.close()
