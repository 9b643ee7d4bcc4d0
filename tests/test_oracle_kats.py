"""Known-answer tests that pin the CPU oracle (oracle/) — the checker every GPU parity test leans on.

The reference ships no tests or golden vectors (its CMakeLists.txt:206-213 are commented out) and g2o is not
available here, so the oracle is "parity unpinned" against the reference itself; what CAN be pinned is pinned here:
closed-form answers for every g2o rule the restatement encodes and, in
test_oracle_golden.py, minima found by an independent optimiser (scipy).
"""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from oracle import oracle as O


def rot(rv):
    return Rotation.from_rotvec(rv).as_matrix()


def test_cauchy_kernel_closed_form():
    # RobustKernelCauchy, delta = 1: rho = ln(1 + e2), rho' = 1 / (1 + e2)     
    import ctypes as C
    for e2 in [0.0, 1e-12, 0.3, 1.0, 17.5, 1e6]:
        r1 = C.c_double()
        r0 = O.lib().og_cauchy_rho(e2, C.byref(r1))
        assert r0 == pytest.approx(np.log(1.0 + e2), rel=1e-15, abs=1e-300)   # g2o: log(aux), aux = 1 + e2
        assert r1.value == pytest.approx(1.0 / (1.0 + e2), rel=1e-15)


def test_quaternion_and_mqt_round_trip():
    L = O.lib()
    rng = np.random.default_rng(0)
    for _ in range(200):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        if q[0] < 0: q = -q
        R = np.zeros((3, 3)); L.og_quat_to_R(O._dp(q), O._dp(R))
        assert np.allclose(R, Rotation.from_quat([q[1], q[2], q[3], q[0]]).as_matrix(), atol=1e-14)
        q2 = np.zeros(4); L.og_R_to_quat(O._dp(np.ascontiguousarray(R)), O._dp(q2))
        assert np.allclose(q2 * np.sign(q2[0]), q, atol=1e-12)
        t = rng.normal(size=3)
        v = np.zeros(6); L.og_to_vector_mqt(O._dp(np.ascontiguousarray(R)), O._dp(t), O._dp(v))
        assert np.allclose(v[:3], t) and np.allclose(v[3:], q[1:], atol=1e-12)   # toVectorMQT = (t, q_xyz), w >= 0
        R3 = np.zeros((3, 3)); t3 = np.zeros(3); L.og_from_vector_mqt(O._dp(v), O._dp(R3), O._dp(t3))
        assert np.allclose(R3, R, atol=1e-12) and np.allclose(t3, t)
    # fromCompactQuaternion with ||v|| > 1 returns identity (g2o)
    v = np.array([1.0, 2.0, 3.0, 0.9, 0.9, 0.9]); R3 = np.zeros((3, 3)); t3 = np.zeros(3)
    L.og_from_vector_mqt(O._dp(v), O._dp(R3), O._dp(t3))
    assert np.array_equal(R3, np.eye(3))


def _two_pose_graph(rng, fixed1=False):
    g = O.Graph()
    R0, R1 = rot(rng.normal(size=3)), rot(rng.normal(size=3))
    t0, t1 = rng.normal(size=3) * 2, rng.normal(size=3) * 2 + 3
    g.add_vertex(0, t0, R0)
    g.add_vertex(1, t1, R1, fixed=fixed1)
    return g, (R0, t0), (R1, t1)


def test_range_error_and_sign():
    # e = measurement - ||(X0*O0).t - (X1*O1).t||        (types_edge_se3range.cpp:105-114)
    rng = np.random.default_rng(1)
    g, (R0, t0), (R1, t1) = _two_pose_graph(rng)
    o0, o1 = np.array([0.1, 0.0, -0.05]), np.array([0.0, 0.2, 0.0])
    g.add_range_edge(0, 1, 4.2, 1.0 / 0.055 ** 2, o0, o1)
    err, J0, J1 = g.linearize(0, O.JAC_ANALYTIC)
    want = 4.2 - np.linalg.norm((R0 @ o0 + t0) - (R1 @ o1 + t1))
    assert err[0] == pytest.approx(want, abs=1e-14)


def test_numeric_and_analytic_range_jacobians_agree():
    # g2o's central difference with delta = 1e-9 vs the exact derivative: noise ~1e-7 relative
    rng = np.random.default_rng(2)
    for trial in range(50):
        g, _, _ = _two_pose_graph(rng)
        o0 = rng.normal(size=3) * 0.2 if trial % 2 else None
        o1 = rng.normal(size=3) * 0.2 if trial % 3 == 0 else None
        g.add_range_edge(0, 1, 3.0, 1.0, o0, o1)
        _, Jn0, Jn1 = g.linearize(0, O.JAC_NUMERIC_G2O)
        _, Ja0, Ja1 = g.linearize(0, O.JAC_ANALYTIC)
        assert np.allclose(Jn0, Ja0, atol=2e-6) and np.allclose(Jn1, Ja1, atol=2e-6)
        if o0 is None:
            assert np.all(Ja0[0, 3:] == 0)   # no lever arm: rotation unobservable (SURVEY §8(a) note)
        assert np.linalg.norm(Ja0[0, :3]) == pytest.approx(1.0, abs=1e-12)  # -u^T R, unit length


def test_coincident_zero_range_edge_has_zero_jacobian():
    # the smoothness edge is created with both endpoints at the same estimate (robot.cpp:90, localization.cpp:338):
    # g2o's central difference gives 0 there up to the rounding of t + R*delta (|J| ~ 1e-7); the analytic
    # mode must give 0 too, not 0/0.
    g = O.Graph()
    R = rot(np.array([0.3, -0.2, 0.1])); t = np.array([1.0, 2.0, 3.0])
    g.add_vertex(0, t, R); g.add_vertex(1, t, R)
    g.add_range_edge(0, 1, 0.0, 100.0)
    for mode in (O.JAC_NUMERIC_G2O, O.JAC_ANALYTIC):
        err, J0, J1 = g.linearize(0, mode)
        assert err[0] == 0.0
        if mode == O.JAC_ANALYTIC:
            assert np.all(J0 == 0) and np.all(J1 == 0)
        else:
            assert np.abs(J0).max() < 1e-6 and np.abs(J1).max() < 1e-6


def _fd_jac(g, vids, vid, edge=0, h=1e-6):
    """central differences of the edge error w.r.t. the MQT increment of vertex vid (X <- X * fromVectorMQT(d))."""
    L = O.lib()
    R, t = g.estimate(vid)
    cols = []
    for d in range(6):
        es = []
        for sgn in (+1, -1):
            v = np.zeros(6); v[d] = sgn * h
            Rd = np.zeros((3, 3)); td = np.zeros(3); L.og_from_vector_mqt(O._dp(v), O._dp(Rd), O._dp(td))
            g.set_estimate(vid, R @ td + t, R @ Rd)
            e, _, _ = g.linearize(edge, O.JAC_ANALYTIC)
            es.append(e)
        g.set_estimate(vid, t, R)
        cols.append((es[0] - es[1]) / (2 * h))
    return np.stack(cols, axis=1)


def test_se3_edge_and_prior_jacobians_match_finite_differences():
    # EdgeSE3: e = toVectorMQT(Z^-1 Xi^-1 Xj); EdgeSE3Prior: e = toVectorMQT(Z^-1 X)       
    rng = np.random.default_rng(3)
    for _ in range(20):
        g, (R0, t0), (R1, t1) = _two_pose_graph(rng)
        Zr, Zt = rot(rng.normal(size=3) * 0.5), rng.normal(size=3)
        g.add_se3_edge(0, 1, Zt, Zr, np.eye(6))
        err, J0, J1 = g.linearize(0)
        # error itself
        E = np.linalg.inv(np.block([[Zr, Zt[:, None]], [np.zeros((1, 3)), np.ones((1, 1))]])) @ \
            np.linalg.inv(np.block([[R0, t0[:, None]], [np.zeros((1, 3)), np.ones((1, 1))]])) @ \
            np.block([[R1, t1[:, None]], [np.zeros((1, 3)), np.ones((1, 1))]])
        q = Rotation.from_matrix(E[:3, :3]).as_quat()
        q = q * np.sign(q[3])
        assert np.allclose(err, np.concatenate([E[:3, 3], q[:3]]), atol=1e-12)
        assert np.allclose(J0, _fd_jac(g, (0, 1), 0), atol=1e-7)
        assert np.allclose(J1, _fd_jac(g, (0, 1), 1), atol=1e-7)
        g2 = O.Graph()
        g2.add_vertex(5, t0, R0)
        g2.add_prior_edge(5, Zt, Zr, np.eye(6))
        _, Jp, _ = g2.linearize(0)
        assert np.allclose(Jp, _fd_jac(g2, (5,), 5), atol=1e-7)


def test_lm_lambda_schedule_hand_computed():
    # One tag at (2,0,0), one fixed anchor at the origin, range 1, Omega 1, no robust kernel  (SURVEY A.6):
    #   e = -1, J = (-1,0,0), H = diag(1,0,0,0,0,0), b = (-1,0,..), lambda0 = 1e-5 * max diag = 1e-5
    #   dx = -1/(1 + 1e-5); accepted; gain ratio ~ 0.999 -> alpha clamps to 1/3 -> lambda = 1e-5 / 3
    g = O.Graph()
    g.add_vertex(0, [0, 0, 0], fixed=True)
    g.add_vertex(1, [2, 0, 0])
    g.add_range_edge(1, 0, 1.0, 1.0, robust=False)
    n, st = g.optimize(1, O.JAC_ANALYTIC)
    assert n == 1 and st.lm_trials == 1 and st.terminated == 0
    _, t = g.estimate(1)
    assert t[0] == pytest.approx(2.0 - 1.0 / (1.0 + 1e-5), abs=1e-14) and t[1] == 0 and t[2] == 0
    assert st.lambda_ == pytest.approx(1e-5 / 3.0, rel=1e-12)
    assert g.chi2() == pytest.approx((1.0 - t[0]) ** 2, rel=1e-9)   # chi2() = last evaluated errors


def test_lm_rejects_and_terminates_at_a_minimum():
    # Start exactly at the minimum: every trial has rho == 0 or < 0 -> Terminate without moving the estimate.
    g = O.Graph()
    g.add_vertex(0, [0, 0, 0], fixed=True)
    g.add_vertex(1, [1, 0, 0])
    g.add_range_edge(1, 0, 1.0, 4.0)
    n, st = g.optimize(10, O.JAC_ANALYTIC)
    assert st.terminated == 1 and n == 1
    assert np.array_equal(g.estimate(1)[1], [1, 0, 0])


def test_noise_free_trilateration_is_exact():
    rng = np.random.default_rng(4)
    anchors = np.array([[3, -3, 0.58], [3, 3, 1.97], [-3, 3, 0.54], [-3, -3, 1.76]], dtype=float)
    for _ in range(10):
        truth = np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(0.5, 1.5)])
        for mode in (O.JAC_NUMERIC_G2O, O.JAC_ANALYTIC):
            g = O.Graph()
            for i, a in enumerate(anchors):
                g.add_vertex(i, a, fixed=True)
            g.add_vertex(10, truth + rng.normal(size=3) * 0.3)
            for i, a in enumerate(anchors):
                g.add_range_edge(10, i, np.linalg.norm(truth - a), 1.0 / 0.055 ** 2)
            g.optimize(30, mode)
            assert np.allclose(g.estimate(10)[1], truth, atol=1e-7)
            assert g.chi2() < 1e-10


def test_fixed_vertices_and_inactive_slots():
    # initializeOptimization: vertices without an active edge are ignored; all-fixed edges are inactive
    g = O.Graph()
    g.add_vertex(0, [0, 0, 0], fixed=True); g.add_vertex(1, [5, 0, 0], fixed=True)
    g.add_vertex(2, [1, 1, 1]); g.add_vertex(3, [9, 9, 9])          # 3 has no edge at all
    g.add_range_edge(0, 1, 1.0, 1.0)                                   # both ends fixed: inactive
    g.add_range_edge(2, 0, 2.0, 1.0)
    g.optimize(5)
    assert np.array_equal(g.estimate(3)[1], [9, 9, 9])
    assert np.array_equal(g.estimate(0)[1], [0, 0, 0])
    assert np.linalg.norm(g.estimate(2)[1]) == pytest.approx(2.0, abs=1e-3)
    g2 = O.Graph(); g2.add_vertex(0, [0, 0, 0])
    assert g2.optimize(3)[0] == -1                                     # "0 vertices to optimize"


def test_remove_vertex_drops_its_edges():
    # optimizer.removeVertex(v, false) at robot.cpp:96
    g = O.Graph()
    for i in range(3):
        g.add_vertex(i, [float(i), 0, 0], fixed=(i == 0))
    g.add_range_edge(1, 0, 1.0, 1.0); g.add_range_edge(2, 1, 1.0, 1.0); g.add_range_edge(2, 0, 2.0, 1.0)
    L = O.lib()
    assert L.og_num_edges(g.h) == 3
    g.remove_vertex(1)
    assert L.og_num_edges(g.h) == 1 and L.og_num_vertices(g.h) == 2
    assert g.add_vertex(0, [0, 0, 0]) == -1    # duplicate id refused


def test_rotation_stays_put_without_lever_arm():
    # SURVEY §8(a) note: identity offsets => rotation rows of J are 0, LM's +lambda*I keeps H PD, d_rot == 0 exactly
    rng = np.random.default_rng(5)
    R = rot(np.array([0.4, -0.3, 0.2]))
    g = O.Graph()
    anchors = rng.uniform(-3, 3, size=(5, 3))
    for i, a in enumerate(anchors): g.add_vertex(i, a, fixed=True)
    g.add_vertex(10, [0.1, 0.2, 0.3], R)
    for i, a in enumerate(anchors): g.add_range_edge(10, i, np.linalg.norm(a - np.array([1.0, -1.0, 0.5])), 100.0)
    g.optimize(15, O.JAC_ANALYTIC)
    R2, t2 = g.estimate(10)
    assert np.array_equal(R2, R)
    assert np.allclose(t2, [1.0, -1.0, 0.5], atol=1e-6)
