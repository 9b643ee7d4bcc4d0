"""GPU parity of wave3_lm_kernel (wave3_kernel.hip) — translation-only chain windows, one WAVE per window: what the drop-in node's
own solve (one ten-pose window per range message, cfg/uwb_only.yaml) and small batches of such windows take — against the 6-DoF
oracle, against the general wave-per-window kernel on the same batches, and the selection rules.

Tolerances as for chain3_lm_kernel (tests/test_gpu_chain3_parity.py): analytic 1e-7 m, numeric (delta = 1e-9) 3e-5 m (64-pose windows:
no further from the oracle than twice the general kernel's own distance, 1e-3 m at most)."""
import os

import numpy as np
import pytest

from test_gpu_chain3_parity import ANCH, _copy_batch, _translation_only_batch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("T,jac,with_z_prior", [
    (10, "analytic", False),   # cfg/uwb_only.yaml's window
    (10, "numeric", False),    # the reference's Jacobian mode
    (1, "numeric", False),     # a lone pose with four ranges
    (12, "analytic", True),    # + lidar-style z priors (translation-only information)
    (16, "numeric", True),     # two groups of 32 lanes, 31 edges: the two trial states scored in one pass
    (20, "numeric", True),
    (64, "analytic", False),   # every lane a pose
    (64, "numeric", True),     # ... and two passes over the edges (more than 64 of them)
])
def test_wave3_kernel_matches_6dof_oracle_and_general_kernel(gpu, T, jac, with_z_prior):
    import localization_amd as la
    from oracle import oracle as O
    from _oracle_window import oracle_solve_instance
    B = 70 if T < 64 else 24
    rng = np.random.default_rng(2000 + 10 * T + len(jac))
    wb = _translation_only_batch(la, rng, B, T, with_z_prior)
    wb.counts[3, 1:] = 0   # an instance whose poses have no edge at all: comes back untouched
    before = wb.poses.copy()
    mode = O.JAC_ANALYTIC if jac == "analytic" else O.JAC_NUMERIC_G2O
    want = [oracle_solve_instance(wb, i, ANCH, jac_mode=mode) for i in range(B)]
    ref = _copy_batch(la, wb)
    general = la.WindowSolver(ANCH, B, *wb.caps, jacobian=jac, chain_threshold=0)
    res_general = general.solve(ref).copy()
    assert general.last_kernel_kind() == "window_lm_kernel"
    general.close()
    s = la.WindowSolver(ANCH, B, *wb.caps, jacobian=jac)       # default thresholds: a small batch
    res = s.solve(wb).copy()
    assert s.last_kernel_kind() == "wave3_lm_kernel"
    tol = 1e-7 if jac == "analytic" else 3e-5
    same_it = 0
    escaped = 0
    for i in range(B):
        nv = int(wb.counts[i, 0])
        if nv == 0 or wb.counts[i, 1] + wb.counts[i, 2] == 0:
            assert np.array_equal(wb.poses[i], before[i]) and res[i, 4] == 0
            same_it += 1
            continue
        poses, chi, st = want[i]
        assert np.array_equal(wb.poses[i, :nv, :9], before[i, :nv, :9])     # rotations never move
        d = np.abs(wb.poses[i, :nv] - poses).max()
        dg = np.abs(ref.poses[i, :nv] - poses).max()
        # (numeric mode, 64 poses: the difference quotient's 5e8 turns last-bit differences of the summation order into 1e-4 m on an
        #  unconverged window — the general kernel is then as far from the oracle as this one)
        # numeric mode: 3e-5 m, up to the few windows whose LM accept / reject sequence the 1e-7 noise of the difference quotient flips —
        # there the general kernel sits as far from the oracle as this one (d < 2 dg), and such windows are COUNTED and bounded below
        if not d < tol:
            assert jac == "numeric" and d < max(2 * dg, tol) and d < 1e-3, (i, d, dg)
            escaped += 1
        assert abs(res[i, 0] - chi) <= (1e-6 if jac == "analytic" else (1e-4 if T < 64 else 1e-3)) * max(1.0, abs(chi)), (i, res[i, 0], chi)
        assert res[i, 7] == nv * 65536 + 2 * nv - 1
        same_it += res[i, 3] == st.outer_iterations
    assert same_it >= 0.9 * B or T == 1   # (a lone well-observed pose converges early: g2o's Terminate is then a rounding-edge event)
    # none beyond the tolerance in analytic mode and on windows of up to 40 poses; measured on the 64-pose numeric case: 2 of 24 (unconverged
    # 64-pose windows: the difference quotient's 5e8 turns last-bit differences of the summation order into 1e-4 m)
    assert escaped <= (B // 8 if (jac == "numeric" and T >= 64) else 0), escaped
    assert np.abs(wb.poses - ref.poses).max() < (tol if jac == "analytic" or T < 64 else 2e-3)   # (both within 1e-3 of the oracle)
    assert np.array_equal(res[:, 6], res_general[:, 6])     # edges sharing their pair with another
    if T > 1:   # (a lone well-observed pose converges early: its remaining accept / reject decisions are taken on rounding-level chi differences)
        assert (res[:, 4] != res_general[:, 4]).mean() < 0.06
    # the same window in another workgroup: the same bits; resident API: the same answer again
    wb2 = _copy_batch(la, wb); wb2.poses[:] = before
    wb2.poses[5] = before[1]
    for name in ("counts", "r_idx", "r_val", "p_idx", "p_val"):
        getattr(wb2, name)[5] = getattr(wb2, name)[1]
    s.upload(wb2); s.solve_resident(); s.download(wb2)
    assert s.last_kernel_kind() == "wave3_lm_kernel"
    n1 = int(wb.counts[1, 0])
    assert np.array_equal(wb2.poses[5, :n1], wb.poses[1, :n1]) and np.array_equal(wb2.result[5], res[1])
    for i in range(B):
        if i != 5:
            nv = int(wb.counts[i, 0])
            assert np.array_equal(wb2.poses[i, :nv], wb.poses[i, :nv]) and np.array_equal(wb2.result[i], res[i]), i
    s.close()


def test_wave3_single_window_many_ranges_per_pose(gpu):
    """One window (the node's case) whose poses each range to all four anchors (the else-branch of addRangeEdge,
    localization.cpp:348, piles ranges onto one vertex): 149 edges = three passes of the edge lanes."""
    import localization_amd as la
    from oracle import oracle as O
    from _oracle_window import oracle_solve_instance
    rng = np.random.default_rng(77)
    T = 30
    wb = la.WindowBatch(1, T, 5 * T, 0, 0)
    truth = np.cumsum(rng.normal(0, 0.05, (T, 3)), axis=0) + np.array([0.4, -0.3, 1.2])
    est = truth + rng.normal(0, 0.05, (T, 3))
    for k in range(T):
        wb.add_pose(0, est[k])
    for k in range(T):
        for a in range(4):
            wb.add_range(0, k, a, float(np.float32(np.linalg.norm(truth[k] - ANCH[a]) + rng.normal(0, 0.03))), 1.0 / 0.055 ** 2, anchor=True)
        if k > 0:
            wb.add_range(0, k - 1, k, 0.0, 1.0 / (5.0 / 32 / 3) ** 2)
    for jac, mode, tol in (("analytic", O.JAC_ANALYTIC, 1e-7), ("numeric", O.JAC_NUMERIC_G2O, 3e-5)):
        w = _copy_batch(la, wb)
        poses, chi, st = oracle_solve_instance(w, 0, ANCH, jac_mode=mode)
        s = la.WindowSolver(ANCH, 1, *w.caps, jacobian=jac)
        res = s.solve(w).copy()
        assert s.last_kernel_kind() == "wave3_lm_kernel"
        s.close()
        assert np.abs(w.poses[0] - poses).max() < tol
        assert abs(res[0, 0] - chi) <= 1e-6 * max(1.0, chi) and res[0, 3] == st.outer_iterations and res[0, 4] == st.lm_trials


def test_wave3_failed_cholesky_like_g2o(gpu):
    """A window whose every range has zero information (H = 0, lambda_0 = 0: the factorisation fails in every trial): 10 trials,
    1 outer iteration, terminated, poses untouched — next to healthy windows of the same launch."""
    import localization_amd as la
    from _oracle_window import oracle_solve_instance
    rng = np.random.default_rng(8)
    B, T = 5, 6
    wb = _translation_only_batch(la, rng, B, T, False)
    wb.r_val[2, :, 1] = 0.0
    before = wb.poses.copy()
    want = [oracle_solve_instance(wb, i, ANCH) for i in range(B)]
    s = la.WindowSolver(ANCH, B, *wb.caps, jacobian="analytic")
    res = s.solve(wb).copy()
    assert s.last_kernel_kind() == "wave3_lm_kernel"
    s.close()
    assert res[2, 5] == 1 and res[2, 4] == 10 and res[2, 3] == 1 and np.array_equal(wb.poses[2], before[2])
    assert want[2][2].terminated == 1 and want[2][2].lm_trials == 10
    for i in (0, 1, 3, 4):
        nv = int(wb.counts[i, 0])
        assert np.abs(wb.poses[i, :nv] - want[i][0]).max() < 1e-7


def test_wave3_selection_rules(gpu):
    import localization_amd as la
    rng = np.random.default_rng(5)
    B, T = 16, 6
    base = _translation_only_batch(la, rng, B, T, True)
    s = la.WindowSolver(ANCH, B, *base.caps, jacobian="analytic")

    def kind(mut, solver=s):
        wb = _copy_batch(la, base)
        mut(wb)
        solver.solve(wb)
        return solver.last_kernel_kind()

    assert kind(lambda wb: None) == "wave3_lm_kernel"

    def lever(wb): wb.r_val[7, 0, 2:5] = (0.0, 0.0, 1e-300)
    def turned(wb): wb.poses[4, 2, :9] = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]]).reshape(9)
    def rot_info(wb): wb.p_val[9, 0, 15] = 1e-3
    def far_pair(wb): wb.r_idx[11, int(wb.counts[11, 1]) - 1] = (4, 1)       # a moving-moving edge that skips poses: not a chain
    for mut in (lever, turned, rot_info, far_pair):
        assert kind(mut) == "window_lm_kernel", mut.__name__
    s.set_option("wave3", 0)
    assert kind(lambda wb: None) == "window_lm_kernel"
    s.set_option("wave3", 1)
    s.L.loc_window_set_chain_threshold(s.h, 0)                                # 0: never anything but the general kernel
    assert kind(lambda wb: None) == "window_lm_kernel"
    s.L.loc_window_set_chain_threshold(s.h, 8)                                # a batch of 16 is then large enough for one lane per window
    assert kind(lambda wb: None) == "chain3_lm_kernel"
    s.close()
    # more than 64 poses per window: not this kernel's
    wide = _translation_only_batch(la, np.random.default_rng(6), 4, 70, False)
    w = la.WindowSolver(ANCH, 4, *wide.caps, jacobian="analytic")
    w.solve(wide)
    assert w.last_kernel_kind() == "window_lm_kernel"
    w.close()
