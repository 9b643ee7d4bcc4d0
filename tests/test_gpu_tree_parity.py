"""GPU parity of tree_lm_kernel (window_kernel.hip) — large batches of forest windows that all share ONE topology (BASELINE config 5:
the key-frame star of addPoseEdge + one anchor range per pose), one lane per window on a host-built elimination schedule — against
the oracle, against the wave-per-window kernel on the same batch, and the selection rules.

Tolerances: analytic vs analytic 1e-7 m / rad on every pose, numeric vs numeric 1e-5."""
import os

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

pytestmark = pytest.mark.gpu

ANCH = np.array([[3, -3, 0.58], [3, 3, 1.97], [-3, 3, 0.54], [-3, -3, 1.76]], dtype=float)


def _copy_batch(la, wb):
    out = la.WindowBatch(wb.B, *wb.caps)
    for name in ("counts", "poses", "r_idx", "r_val", "p_idx", "p_val", "s_idx", "s_val"):
        getattr(out, name)[:] = getattr(wb, name)
    return out


def _forest_batch(la, rng, B, T, every, rich):
    """Key-frame stars as addPoseEdge builds them (a new key every `every` poses; the pose before a new key IS that key), one anchor range
    per pose.  rich: two trees (one link missing), a smoothness range next to the EdgeSE3 of some pairs, IMU-style priors on some poses,
    lever arms, EdgeSE3 stored in either direction, a non-robust EdgeSE3."""
    wb = la.WindowBatch(B, T, 2 * T + 2, T, T)
    key = np.where(np.arange(T) < every, 0, (np.arange(T) // every) * every - 1)
    for i in range(B):
        tt = np.cumsum(rng.normal(0, 0.05, (T, 3)), axis=0) + np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1.5, 1.5), 1.1])
        tR = Rotation.from_rotvec(np.cumsum(rng.normal(0, 0.03, (T, 3)), axis=0) + rng.normal(0, 0.3, 3))
        et = tt + rng.normal(0, 0.05, (T, 3))
        eR = (tR * Rotation.from_rotvec(rng.normal(0, 0.02, (T, 3)))).as_matrix()
        off = np.array([0.1, 0.0, -0.05]) if rich else np.zeros(3)
        for k in range(T): wb.add_pose(i, et[k], eR[k])
        for k in range(T):
            a = k % 4
            wb.add_range(i, k, a, float(np.float32(np.linalg.norm(tt[k] + tR[k].apply(off) - ANCH[a]) + rng.normal(0, 0.03))), 1 / 0.055 ** 2, off, anchor=True)
            if k == 0 or (rich and k == T // 2):
                continue                                                     # (rich: pose T/2 starts a second tree)
            kk = int(key[k])
            Zt = tR[kk].inv().apply(tt[k] - tt[kk]) + rng.normal(0, 0.01, 3)
            ZR = (tR[kk].inv() * tR[k] * Rotation.from_rotvec(rng.normal(0, 0.01, 3))).as_matrix()
            A = rng.normal(size=(6, 6)); info = A @ A.T + 6 * np.eye(6); info *= 6e4 / np.trace(info)
            if rich and k % 3 == 0:                                           # stored child -> key: the inverse measurement
                wb.add_se3(i, k, kk, -ZR.T @ Zt, ZR.T, info, True)
            else:
                wb.add_se3(i, kk, k, Zt, ZR, info, not (rich and k % 5 == 0))
            if rich and kk == k - 1:
                wb.add_range(i, k - 1, k, 0.0, 1 / (5.0 / 32 / 3) ** 2)       # the smoothness edge lands on the same pair as the EdgeSE3
            if rich and k % 4 == 1:
                wb.add_prior(i, k, et[k], (tR[k] * Rotation.from_rotvec(rng.normal(0, 2e-3, 3))).as_matrix(), np.array([0, 0, 0, 1, 1, 1.0]) / 4.592449e-06)
    return wb


@pytest.mark.parametrize("T,every,rich,jac", [
    (64, 8, False, "analytic"),    # BASELINE config 5's shape
    (64, 8, False, "numeric"),
    (24, 4, True, "analytic"),     # two trees, doubled pairs, priors, lever arms
    (24, 4, True, "numeric"),
    (10, 1, True, "analytic"),     # every pose a key: a chain of EdgeSE3 + smoothness edges with a gap
])
def test_tree_kernel_matches_oracle_and_general_kernel(gpu, T, every, rich, jac):
    import localization_amd as la
    from oracle import oracle as O
    from _oracle_window import oracle_solve_instance
    B = 70 if T < 64 else 66
    rng = np.random.default_rng(7000 + T + every + len(jac))
    wb = _forest_batch(la, rng, B, T, every, rich)
    before = wb.poses.copy()
    ref = _copy_batch(la, wb)
    mode = O.JAC_ANALYTIC if jac == "analytic" else O.JAC_NUMERIC_G2O
    n_or = B if T < 64 else 12
    want = [oracle_solve_instance(wb, i, ANCH, jac_mode=mode) for i in range(n_or)]
    g = la.WindowSolver(ANCH, B, *wb.caps, jacobian=jac, bw_max=T - 1, chain_threshold=0)
    res_g = g.solve(ref).copy()
    assert g.last_kernel_kind() == "window_lm_kernel"
    g.close()
    s = la.WindowSolver(ANCH, B, *wb.caps, jacobian=jac, bw_max=T - 1, chain_threshold=1)
    res = s.solve(wb).copy()
    kind = s.last_kernel_kind()
    assert kind in ("tree_wave_kernel", "chain_lm_kernel") and (kind == "tree_wave_kernel" or every == 1)
    tol = 1e-7 if jac == "analytic" else 1e-5
    n_same_it = 0
    for i in range(n_or):
        poses, chi, st = want[i]
        d = np.abs(wb.poses[i] - poses).max()
        assert d < tol, (i, d)
        assert abs(res[i, 0] - chi) <= (1e-6 if jac == "analytic" else 1e-4) * max(1.0, abs(chi)), (i, res[i, 0], chi)
        n_same_it += res[i, 3] == st.outer_iterations
    assert n_same_it >= 0.9 * n_or     # (g2o's Terminate — ten rejected trials or a gain ratio of exactly 0 — is a rounding-edge event at convergence)
    assert np.abs(wb.poses - ref.poses).max() < tol
    assert np.array_equal(res[:, 6], res_g[:, 6])          # pose-to-pose edges that share their pair with another edge
    assert (res[:, 4] != res_g[:, 4]).mean() < (0.05 if T > 10 else 0.3)   # (ten-pose windows converge: their last accept / reject decisions are rounding-level ties)
    # resident API: the same bits
    wb2 = _copy_batch(la, wb); wb2.poses[:] = before
    s.upload(wb2); s.solve_resident(); s.download(wb2)
    assert s.last_kernel_kind() == kind and np.array_equal(wb2.poses, wb.poses) and np.array_equal(wb2.result, res)
    # a host-path solve of ANOTHER batch (another topology, another schedule) while this one is resident must not disturb the resident
    # batch's schedule: the handle keeps one set of host-built tables per path
    other = _forest_batch(la, np.random.default_rng(1), B, T, max(every, 2) if T >= 12 else every, False)
    s.solve(other)
    s.solve_resident(); s.download(wb2)
    assert s.last_kernel_kind() == kind and np.array_equal(wb2.poses, wb.poses) and np.array_equal(wb2.result, res)
    s.close()


@pytest.mark.parametrize("jac", ["analytic", "numeric"])
def test_lane_per_window_tree_kernel_on_the_same_schedule(gpu, jac):
    """tree_lm_kernel — one lane per window walking the host's schedule, the state in an [entry][lane] workspace — serves batches whose
    nodes have several EdgeSE3 to their parent; LOCAMD_TREE=lane selects it for any forest batch: the same answers as tree_wave_kernel
    and the oracle, on a batch with two trees, doubled pairs, priors and lever arms."""
    import localization_amd as la
    from oracle import oracle as O
    from _oracle_window import oracle_solve_instance
    B, T = 70, 24
    rng = np.random.default_rng(31 + len(jac))
    wb = _forest_batch(la, rng, B, T, 4, True)
    # a second EdgeSE3 on one child-parent pair of every window: tree_wave_kernel does not take such batches
    for i in range(B):
        wb.add_se3(i, 3, 5, *_rel(wb, i, 3, 5), np.eye(6) * 2e3, True)
    mode = O.JAC_ANALYTIC if jac == "analytic" else O.JAC_NUMERIC_G2O
    want = [oracle_solve_instance(wb, i, ANCH, jac_mode=mode) for i in range(12)]
    s = la.WindowSolver(ANCH, B, *wb.caps, jacobian=jac, bw_max=T - 1, chain_threshold=1)
    res = s.solve(wb).copy()
    assert s.last_kernel_kind() == "tree_lm_kernel" and (res[:, 7] % 65536 == 2 * T - 2).all()   # (two trees: 2 nv - 2 blocks)
    tol = 1e-7 if jac == "analytic" else 1e-5
    for i in range(12):
        assert np.abs(wb.poses[i] - want[i][0]).max() < tol, (i, np.abs(wb.poses[i] - want[i][0]).max())
    # without the doubled EdgeSE3 both variants apply: lane-per-window (forced) against lane-per-pose
    plain_lane = _forest_batch(la, np.random.default_rng(5), B, T, 4, True)
    plain_wave = _copy_batch(la, plain_lane)
    s.set_option("tree", 2)                                # the lane-per-window variant (was LOCAMD_TREE=lane)
    s.solve(plain_lane)
    s.set_option("tree", -1)
    s.solve(plain_wave)
    assert np.abs(plain_lane.poses - plain_wave.poses).max() < tol
    s.close()


def _rel(wb, i, a, b):
    """relative pose a -> b of window i's current estimates (a measurement the estimates satisfy), as (t, R)"""
    Ra, ta = wb.poses[i, a, :9].reshape(3, 3), wb.poses[i, a, 9:]
    Rb, tb = wb.poses[i, b, :9].reshape(3, 3), wb.poses[i, b, 9:]
    return Ra.T @ (tb - ta), Ra.T @ Rb


def test_tree_kernel_needs_one_shared_forest_topology(gpu):
    import localization_amd as la
    rng = np.random.default_rng(3)
    B, T = 64, 12
    base = _forest_batch(la, rng, B, T, 4, False)
    s = la.WindowSolver(ANCH, B, *base.caps, jacobian="analytic", bw_max=T - 1, chain_threshold=1)

    def kind(mut):
        wb = _copy_batch(la, base)
        mut(wb)
        s.solve(wb)
        return s.last_kernel_kind()

    assert kind(lambda wb: None) == "tree_wave_kernel"
    def other_anchor(wb): wb.r_idx[17, 3, 1] = -1 - 2                        # one window ranges another anchor: not ONE topology
    def shorter(wb): wb.counts[5] = (T - 1, T - 1, 0, T - 2)                  # one window is shorter
    def cycle(wb):                                                            # every window gets a loop closure: not a forest
        for i in range(B): wb.add_range(i, 2, 9, 1.0, 10.0)
    for mut in (other_anchor, shorter, cycle):
        assert kind(mut) == "window_lm_kernel", mut.__name__
    s.close()
    small = la.WindowSolver(ANCH, B, *base.caps, jacobian="analytic", bw_max=T - 1)   # default threshold: 64 windows are a small batch
    small.solve(_copy_batch(la, base))
    assert small.last_kernel_kind() == "window_lm_kernel"
    small.close()


@pytest.mark.parametrize("seed,T,jac", [(1, 40, "analytic"), (2, 64, "analytic"), (3, 33, "numeric"), (4, 64, "numeric")])
def test_tree_wave_kernel_on_random_forests(gpu, seed, T, jac):
    """tree_wave_kernel on forests nobody designed: every pose hangs on a random earlier pose (bushy nodes with a dozen children, nodes with
    many inner children, chains), two or three trees, every pose 1 … 3 anchor ranges, some child-parent pairs a smoothness range next to
    their EdgeSE3, priors on a few poses, EdgeSE3 stored in either direction.  One topology for the whole batch, other measurements and
    estimates per window; against the oracle (12 windows) and the general kernel (all)."""
    import localization_amd as la
    from oracle import oracle as O
    from _oracle_window import oracle_solve_instance
    B = 66
    rng = np.random.default_rng(900 + seed)
    # the topology
    parent = np.full(T, -1)
    roots = sorted(rng.choice(np.arange(1, T), size=int(rng.integers(1, 3)), replace=False).tolist() + [0])
    for k in range(1, T):
        if k in roots:
            continue
        parent[k] = int(rng.integers(0, k)) if rng.random() < 0.7 else int(rng.choice([0, max(0, k - 1), k // 2]))
    # (every pose at least one anchor range, roots three: a pose held by its EdgeSE3 alone leaves directions so weakly observed that, in numeric
    #  mode, the oracle's own two Jacobian modes end 1e-3 … 1e-2 m apart after the fixed 10 iterations — chi2 agreeing to six digits)
    n_anchor = rng.integers(1, 4, T); n_anchor[roots] = 3
    smooth = rng.random(T) < 0.25
    prior = rng.random(T) < 0.15
    flip = rng.random(T) < 0.4
    wb = la.WindowBatch(B, T, 4 * T + 2, T, T)
    for i in range(B):
        tt = np.cumsum(rng.normal(0, 0.08, (T, 3)), axis=0) + np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1.5, 1.5), 1.1])
        tR = Rotation.from_rotvec(np.cumsum(rng.normal(0, 0.03, (T, 3)), axis=0) + rng.normal(0, 0.3, 3))
        et = tt + rng.normal(0, 0.04, (T, 3))
        eR = (tR * Rotation.from_rotvec(rng.normal(0, 0.02, (T, 3)))).as_matrix()
        off = np.array([0.08, -0.02, 0.05])
        for k in range(T): wb.add_pose(i, et[k], eR[k])
        for k in range(T):
            for a in range(int(n_anchor[k])):
                an = (k + a) % 4
                wb.add_range(i, k, an, float(np.float32(np.linalg.norm(tt[k] + tR[k].apply(off) - ANCH[an]) + rng.normal(0, 0.03))), 1 / 0.055 ** 2, off, anchor=True)
            p = int(parent[k])
            if p < 0:
                continue
            Zt = tR[p].inv().apply(tt[k] - tt[p]) + rng.normal(0, 0.01, 3)
            ZR = (tR[p].inv() * tR[k] * Rotation.from_rotvec(rng.normal(0, 0.01, 3))).as_matrix()
            A = rng.normal(size=(6, 6)); info = A @ A.T + 6 * np.eye(6); info *= 6e4 / np.trace(info)
            if flip[k]: wb.add_se3(i, k, p, -ZR.T @ Zt, ZR.T, info, True)
            else: wb.add_se3(i, p, k, Zt, ZR, info, k % 7 != 0)
            if smooth[k]:
                if k % 2: wb.add_range(i, p, k, float(np.linalg.norm(tt[k] - tt[p])), 1 / 0.1 ** 2)
                else: wb.add_range(i, k, p, float(np.linalg.norm(tt[k] - tt[p])), 1 / 0.1 ** 2)
            if prior[k]:
                wb.add_prior(i, k, et[k], (tR[k] * Rotation.from_rotvec(rng.normal(0, 2e-3, 3))).as_matrix(), np.array([0, 0, 0.5, 1, 1, 1.0]) / 4.592449e-06)
    ref = _copy_batch(la, wb)
    wb0 = _copy_batch(la, wb)
    mode = O.JAC_ANALYTIC if jac == "analytic" else O.JAC_NUMERIC_G2O
    want = [oracle_solve_instance(wb, i, ANCH, jac_mode=mode) for i in range(12)]
    g = la.WindowSolver(ANCH, B, *wb.caps, jacobian=jac, bw_max=T - 1, chain_threshold=0)
    g.solve(ref)
    assert g.last_kernel_kind() == "window_lm_kernel"
    g.close()
    s = la.WindowSolver(ANCH, B, *wb.caps, jacobian=jac, bw_max=T - 1, chain_threshold=1)
    res = s.solve(wb).copy()
    assert s.last_kernel_kind() == "tree_wave_kernel"
    s.close()
    if jac == "analytic":
        for i in range(12):
            poses, chi, st = want[i]
            assert np.abs(wb.poses[i] - poses).max() < 1e-7, (i, np.abs(wb.poses[i] - poses).max())
            assert abs(res[i, 0] - chi) <= 1e-6 * max(1.0, abs(chi))
        assert np.abs(wb.poses - ref.poses).max() < 1e-7
    else:
        # Numeric Jacobians on deep random trees started 4 … 8 cm off: after the reference's fixed 10 iterations the iterates are NOT converged, and
        # the 1e-7 relative noise of the difference quotient moves them along weakly observed directions — the oracle's own two Jacobian modes end
        # 1e-3 … 1e-2 m apart on such windows, chi2 agreeing to six digits.  So: chi2 against the oracle tightly, and the poses no further
        # from the numeric oracle than three times what the general kernel or the oracle's other mode are.
        want_a = [oracle_solve_instance(wb0, i, ANCH, jac_mode=O.JAC_ANALYTIC) for i in range(12)]
        for i in range(12):
            poses, chi, st = want[i]
            d = np.abs(wb.poses[i] - poses).max()
            dg = np.abs(ref.poses[i] - poses).max()
            da = np.abs(want_a[i][0] - poses).max()
            assert d < max(1e-5, 3 * dg, 3 * da), (i, d, dg, da)
            assert abs(res[i, 0] - chi) <= 1e-5 * max(1.0, abs(chi)), (i, res[i, 0], chi)
            assert res[i, 3] == st.outer_iterations
    kids = np.bincount(parent[parent >= 0], minlength=T)
    assert kids.max() >= 4     # (the generator really makes bushy nodes)


def test_tree_wave_kernel_on_a_small_handle_reads_the_staging_block_in_place(gpu):
    """A handle of <= 4 windows whose threshold was lowered runs tree_wave_kernel on the page-locked staging block itself (no copies around the
    kernel, as for the node's wave3 / wave6 windows) when no pose has priors or more than two ranges — the kernel reads its inputs once, in its
    prologue; with priors the batch is copied to the device first.  Both against the general kernel."""
    import localization_amd as la
    for rich in (False, True):
        B, T = 2, 24
        wb = _forest_batch(la, np.random.default_rng(77), B, T, 4, rich)
        ref = _copy_batch(la, wb)
        g = la.WindowSolver(ANCH, B, *wb.caps, jacobian="analytic", bw_max=T - 1, chain_threshold=0)
        g.solve(ref)
        assert g.last_kernel_kind() == "window_lm_kernel"
        g.close()
        s = la.WindowSolver(ANCH, B, *wb.caps, jacobian="analytic", bw_max=T - 1, chain_threshold=1)
        s.solve(wb)
        assert s.last_kernel_kind() == "tree_wave_kernel"
        again = _copy_batch(la, ref); again.poses[:] = _forest_batch(la, np.random.default_rng(77), B, T, 4, rich).poses
        s.set_option("zero_copy", 0)
        s.solve(again)
        s.close()
        assert np.abs(wb.poses - ref.poses).max() < 1e-7
        assert np.array_equal(again.poses, wb.poses)          # staged in place or copied first: the same bits
