"""Semantics of the oracle's restatement of class Localization / class Robot (reference localization.cpp, robot.cpp):
the gates, covariances, edge topology and ring window that DEFINE the cost function the GPU path has to reproduce."""
import numpy as np
import pytest

from oracle import oracle as O

ANCH = np.array([[3, -3, 0.58], [3, 3, 1.97], [-3, 3, 0.54], [-3, -3, 1.76]], dtype=float)
IDS = [100, 101, 102, 103, 200]


def make(T=5, **kw):
    cfg = dict(trajectory_length=T, maximum_velocity=5.0, distance_outlier=1.0, maximum_iteration=10,
               minimum_optimize_error=2000.0, publish_range=True)
    cfg.update(kw)
    return O.LocalizationOracle(IDS, np.concatenate([ANCH, [[0, 0, 1.0]]]), **cfg)


def feed(lo, truth, n, t0=100.0, dt=1 / 32.0, noise=0.0, rng=None):
    outs = []
    for i in range(n):
        a = i % 4
        d = np.linalg.norm(truth - ANCH[a]) + (rng.normal(0, noise) if rng is not None else 0.0)
        outs.append(lo.add_range(200, IDS[a], t0 + i * dt, d, 0.055, 0, "uwb"))
    return outs


def test_warm_up_then_solve_every_range():
    # nothing is solved until number_measurements > trajectory_length (localization.cpp:371)
    lo = make(T=5)
    outs = feed(lo, np.array([0.5, -0.4, 1.1]), 12)
    assert [o["solved"] for o in outs] == [False] * 5 + [True] * 7
    assert all(o["rc"] == (1 if o["solved"] else 0) for o in outs)
    last = outs[-1]
    assert last["published"] and last["realtime"][0] == pytest.approx(100.0 + 11 / 32.0)
    # the zero-range smoothness edges (sigma_v = vmax*dt/3 = 5 cm per step) let the estimate walk to the truth
    outs = feed(lo, np.array([0.5, -0.4, 1.1]), 120, t0=100.0 + 12 / 32.0)
    assert np.allclose(outs[-1]["realtime"][1:4], [0.5, -0.4, 1.1], atol=2e-2)


def test_outlier_gate_only_after_warm_up_and_counts_rejected():
    # localization.cpp:303-313: the counter increments even for rejected ranges; gate uses |d_hat - d| > outlier
    lo = make(T=4)
    truth = np.array([0.2, 0.1, 1.0])
    feed(lo, truth, 8)
    n0 = lo.L.lo_number_measurements(lo.h)
    o = lo.add_range(200, 100, 101.0, np.linalg.norm(truth - ANCH[0]) + 2.5, 0.055, 0, "uwb")   # +2.5 m: rejected
    assert o["rc"] == 0 and not o["solved"]
    assert lo.L.lo_number_measurements(lo.h) == n0 + 1
    o = lo.add_range(200, 100, 101.1, np.linalg.norm(truth - ANCH[0]) + 0.5, 0.055, 0, "uwb")   # within 1 m: used
    assert o["solved"]
    # during warm-up a wild range is accepted (no gate yet)
    lo2 = make(T=4)
    o = lo2.add_range(200, 100, 100.0, 40.0, 0.055, 0, "uwb")
    assert o["rc"] == 0 and lo2.L.lo_number_measurements(lo2.h) == 1


def test_unknown_node_is_an_error_not_a_crash():
    lo = make()
    assert lo.add_range(200, 177, 100.0, 3.0, 0.055, 0, "uwb")["rc"] < 0      # reference: std::map::at throws (:306)
    assert lo.add_range(9, 100, 100.0, 3.0, 0.055, 0, "uwb")["rc"] < 0


def test_ring_window_drops_oldest_and_orders_path_by_age():
    # robot.cpp:61-72,86-109: ring of T poses, oldest removed with its edges, vertices2path oldest -> newest
    T = 4
    lo = make(T=T)
    truth = np.array([0.0, 0.0, 1.0])
    for i in range(11):
        lo.add_range(200, IDS[i % 4], 100.0 + i, np.linalg.norm(truth - ANCH[i % 4]), 0.055, 0, "uwb")
    path = lo.path(200)
    assert path.shape == (T, 8)
    assert list(path[:, 0]) == [107.0, 108.0, 109.0, 110.0]
    assert len(lo.path(100)) == 1                                          # anchors: one fixed vertex
    assert np.allclose(lo.path(100)[0, 1:4], ANCH[0])


def test_optimized_pose_is_window_middle():
    # publish(): optimized = path->poses[trajectory_length / 2]  (localization.cpp:220)
    T = 6
    lo = make(T=T)
    truth = np.array([0.3, 0.3, 1.2])
    outs = feed(lo, truth, 20, t0=50.0, dt=1.0)
    o = outs[-1]
    path = lo.path(200)
    assert np.array_equal(o["optimized"], path[T // 2])
    assert np.array_equal(o["realtime"], path[-1])
    assert o["optimized"][0] == 50.0 + 19 - (T - 1 - T // 2)


def test_publish_gate_keeps_state_but_reports_unpublished():
    # chi2 >= minimum_optimize_error -> not published, estimate not rolled back (localization.cpp:197-205)
    lo = make(T=3, minimum_optimize_error=1e-6)
    outs = feed(lo, np.array([0.1, 0.2, 1.0]), 8, noise=0.05, rng=np.random.default_rng(0))
    solved = [o for o in outs if o["solved"]]
    assert solved and not any(o["published"] for o in solved)
    assert all(np.isfinite(o["chi2"]) and o["chi2"] >= 1e-6 for o in solved)


def test_imu_prior_once_per_vertex_and_rotation_overwrite():
    # addImuEdge acts once per range vertex (frame tag check, localization.cpp:501-503), overwrites R keeping t (:505-513)
    lo = make(T=4, publish_imu=False)
    feed(lo, np.array([0.0, 0.0, 1.0]), 6)
    q = np.array([0.0, 0.0, np.sin(0.25), np.cos(0.25)])      # yaw 0.5 rad, xyzw
    cov = np.diag([4.592449e-06] * 3).ravel()
    before = lo.path(200)[-1].copy()
    assert lo.add_imu(200.0, q, cov)["rc"] == 0
    after = lo.path(200)[-1]
    assert np.allclose(after[1:4], before[1:4]) and np.allclose(after[4:8], q, atol=1e-12)
    q2 = np.array([0.0, 0.0, np.sin(0.4), np.cos(0.4)])
    lo.add_imu(200.01, q2, cov)                                   # same vertex: ignored
    assert np.allclose(lo.path(200)[-1][4:8], q, atol=1e-12)
    # next range copies the estimate (incl. rotation) into the new vertex (robot.cpp:90); IMU then applies again
    lo.add_range(200, 100, 200.1, np.linalg.norm(np.array([0, 0, 1.0]) - ANCH[0]), 0.055, 0, "uwb")
    lo.add_imu(200.11, q2, cov)
    assert np.allclose(lo.path(200)[-1][4:8], q2, atol=1e-9)


def test_range_after_pose_vertex_attaches_to_last_vertex():
    # frame_id of the newest vertex does not contain the UWB frame -> one edge on the previous vertex with inflated
    # covariance, no new vertex (localization.cpp:327, :346-357)
    lo = make(T=6, publish_range=False)
    feed(lo, np.array([0.0, 0.0, 1.0]), 3)
    pose = np.array([0.0, 0.0, 0.0, 0, 0, 0, 1.0])
    cov = (np.eye(6) * 1e-4).ravel()
    assert lo.add_pose(300.0, pose, cov, "keyframe_1")["rc"] == 0
    stamps_before = lo.path(200)[:, 0].copy()
    assert lo.add_range(200, 100, 300.5, 3.0, 0.055, 0, "uwb")["rc"] == 0
    assert np.array_equal(lo.path(200)[:, 0], stamps_before)       # no new vertex was created


def test_twist_edge_creates_vertex_and_moves_estimate():
    lo = make(T=5, publish_range=False, publish_twist=True)
    feed(lo, np.array([0.0, 0.0, 1.0]), 7, t0=10.0, dt=0.1)
    p0 = lo.path(200)[-1].copy()
    tw = np.array([1.0, 0.0, 0.0, 0.0, 0.0, 0.0])
    o = lo.add_twist(10.0 + 0.6 + 0.5, tw, (np.eye(6) * 1e-2).ravel())
    assert o["rc"] == 1 and o["solved"]
    p1 = lo.path(200)[-1]
    assert p1[0] == pytest.approx(11.1) and p1[1] - p0[1] == pytest.approx(0.5, abs=0.1)


def test_lidar_prior_sets_height():
    lo = make(T=4, publish_range=False, publish_lidar=True)
    feed(lo, np.array([0.0, 0.0, 1.0]), 6)
    o = lo.add_lidar(500.0, 1.234, "lidar")
    # information 1/0.05 = 20 on z (localization.cpp:479) against ~330 per range: the height is pulled, not pinned
    assert o["solved"] and 1.05 < lo.path(200)[-1][3] < 1.234
