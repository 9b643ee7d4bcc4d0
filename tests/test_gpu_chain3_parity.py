"""GPU parity of the TRANSLATION-ONLY lane-per-window kernel (chain3_lm_kernel, chain3_kernel.hip) — the exact 3-DoF reduction
of cfg/uwb_only.yaml's window — against the 6-DoF oracle (the g2o restatement with full VertexSE3 poses and 6x6 blocks), against
the 6-DoF lane-per-window kernel and against the wave-per-window kernel on the same batches, and the kernel-selection rules.

Tolerances (fp64 on both sides, the reference's fixed 10 iterations):
  same Jacobian mode, GPU vs oracle: analytic 1e-7 m; numeric (delta = 1e-9) 3e-5 m on these graphs (the near-zero smoothness
    ranges amplify last-bit differences of the summation order by 5e8, as in test_chain_kernel_matches_oracle);
  analytic GPU vs numeric oracle (= the reference's configuration) at the batch size where the host picks these kernels:
    median 1e-6 m, 99 % below 1e-4 m, max 5e-3 m — unconverged iterates of a few windows follow a different, equally valid LM
    accept/reject sequence (the oracle's own two modes differ by the same amounts).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ANCH = np.array([[3, -3, 0.58], [3, 3, 1.97], [-3, 3, 0.54], [-3, -3, 1.76]], dtype=float)


def _copy_batch(la, wb):
    out = la.WindowBatch(wb.B, *wb.caps)
    for name in ("counts", "poses", "r_idx", "r_val", "p_idx", "p_val", "s_idx", "s_val"):
        getattr(out, name)[:] = getattr(wb, name)
    return out


def _translation_only_batch(la, rng, B, T, with_z_prior):
    """Windows in addRangeEdge's creation order, identity rotations, no lever arm; ragged lengths, missing links, doubled pairs."""
    nr_max, np_max = max(2 * T + 2, 4), (T if with_z_prior else 0)
    wb = la.WindowBatch(B, T, nr_max, np_max, 0)
    for i in range(B):
        Ti = T if i % 7 else max(T // 2, 1)
        truth = np.cumsum(rng.normal(0, 0.05, (Ti, 3)), axis=0) + np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1.5, 1.5), 1.1])
        est = truth + rng.normal(0, 0.05, (Ti, 3))
        for k in range(Ti):
            wb.add_pose(i, est[k])
        for k in range(Ti):
            for a in ([int(rng.integers(0, 4))] if Ti > 1 else [0, 1, 2, 3]):
                d = float(np.float32(np.linalg.norm(truth[k] - ANCH[a]) + rng.normal(0, 0.03)))
                wb.add_range(i, k, a, d, 1.0 / 0.055 ** 2, anchor=True)
            if k > 0 and not (i % 5 == 2 and k == 3):        # (some windows miss a link: two independent chains)
                wb.add_range(i, k - 1, k, 0.0, 1.0 / (5.0 / 32 / 3) ** 2)
                if i % 5 == 1 and k == 2:                     # (some have two edges on one pair, the second the other way round)
                    wb.add_range(i, k, k - 1, 0.02, 0.5 / (5.0 / 32 / 3) ** 2)
            if with_z_prior and k % 2 == 0:                   # addLidarEdge: z prior, information only on (2, 2)
                wb.add_prior(i, k, np.array([est[k, 0], est[k, 1], truth[k, 2] + rng.normal(0, 0.02)]), np.eye(3), np.array([0, 0, 1 / 0.05, 0, 0, 0.0]))
    return wb


@pytest.mark.parametrize("T,jac,with_z_prior", [
    (10, "analytic", False),   # cfg/uwb_only.yaml's window
    (10, "numeric", False),    # the reference's Jacobian mode
    (1, "analytic", False),    # a lone pose with four ranges
    (12, "analytic", True),    # + lidar-style z priors (translation-only information): still 3-DoF; (G, y) in LDS, translations in HBM
    (20, "numeric", True),     # everything in the HBM slab
    (40, "analytic", False),   # beyond 32 poses the coupling blocks are always stored in full
])
def test_chain3_kernel_matches_6dof_oracle_and_6dof_kernels(gpu, T, jac, with_z_prior):
    import localization_amd as la
    from oracle import oracle as O
    from _oracle_window import oracle_solve_instance
    B = 70
    rng = np.random.default_rng(1000 + 10 * T + len(jac))
    wb = _translation_only_batch(la, rng, B, T, with_z_prior)
    wb.counts[3, 1:] = 0   # an instance whose poses have no edge at all: comes back untouched
    before = wb.poses.copy()
    mode = O.JAC_ANALYTIC if jac == "analytic" else O.JAC_NUMERIC_G2O
    want = [oracle_solve_instance(wb, i, ANCH, jac_mode=mode) for i in range(B)]
    wave, six = _copy_batch(la, wb), _copy_batch(la, wb)
    general = la.WindowSolver(ANCH, B, *wb.caps, jacobian=jac, chain_threshold=0)
    res_general = general.solve(wave).copy()
    assert general.last_kernel_kind() == "window_lm_kernel"
    general.close()
    chain = la.WindowSolver(ANCH, B, *wb.caps, jacobian=jac, chain_threshold=1)
    chain.set_option("chain3", 0)
    res6 = chain.solve(six).copy()
    assert chain.last_kernel_kind() == "chain_lm_kernel"
    chain.set_option("chain3", 1)
    res = chain.solve(wb).copy()
    assert chain.last_kernel_kind() == "chain3_lm_kernel"
    tol = 1e-7 if jac == "analytic" else 3e-5
    for i in range(B):
        nv = int(wb.counts[i, 0])
        if nv == 0 or wb.counts[i, 1] + wb.counts[i, 2] == 0:
            assert np.array_equal(wb.poses[i], before[i]) and res[i, 4] == 0
            continue
        poses, chi, st = want[i]
        assert np.array_equal(wb.poses[i, :nv, :9], before[i, :nv, :9])     # rotations never move
        assert np.abs(poses[:, :9] - before[i, :nv, :9]).max() < 1e-15     # ... in the 6-DoF oracle either
        d = np.abs(wb.poses[i, :nv] - poses).max()
        assert d < tol, (i, d)
        assert abs(res[i, 0] - chi) <= (1e-6 if jac == "analytic" else 1e-4) * max(1.0, abs(chi)), (i, res[i, 0], chi)
    # the three kernels agree among themselves: every dropped term of the 3-DoF form is an exact zero of the 6-DoF one
    assert np.abs(wb.poses - six.poses).max() < (1e-9 if jac == "analytic" else tol)
    assert np.abs(wb.poses - wave.poses).max() < tol
    assert np.array_equal(res[:, 6], res6[:, 6]) and np.array_equal(res[:, 6], res_general[:, 6])   # edges sharing their pair with another
    if T > 1:   # (a lone well-observed pose converges early: its remaining accept / reject decisions are taken on rounding-level chi differences)
        assert (res[:, 4] != res6[:, 4]).mean() < 0.03 and (res[:, 3] == res6[:, 3]).all()
        assert (res[:, 4] != res_general[:, 4]).mean() < 0.05
    # resident API: the same answer again, and the threshold is looked at per solve
    wb2 = _copy_batch(la, wb); wb2.poses[:] = before
    chain.upload(wb2)
    with pytest.raises(la.LocalizationAmdError):
        chain.download(wb2)                      # nothing solved since the upload
    chain.solve_resident()
    chain.download(wb2)
    assert chain.last_kernel_kind() == "chain3_lm_kernel"
    for i in range(B):
        nv = int(wb.counts[i, 0])
        assert np.array_equal(wb2.poses[i, :nv], wb.poses[i, :nv]), i
    assert np.array_equal(wb2.result, res)
    chain.L.loc_window_set_chain_threshold(chain.h, 0)
    chain.solve_resident()
    assert chain.last_kernel_kind() == "window_lm_kernel"
    import ctypes as C
    rc = chain.L.loc_window_set_anchors(chain.h, 2, ANCH[:2].ctypes.data_as(C.POINTER(C.c_double)))   # the resident batch references 4 anchors
    assert rc == -1 and b"resident batch" in chain.L.loc_last_error()
    chain.close()


def test_chain3_is_taken_only_by_translation_only_batches(gpu):
    import localization_amd as la
    rng = np.random.default_rng(5)
    B, T = 64, 6
    base = _translation_only_batch(la, rng, B, T, True)
    s = la.WindowSolver(ANCH, B, *base.caps, jacobian="analytic", chain_threshold=1)

    def kind(mut):
        wb = _copy_batch(la, base)
        mut(wb)
        s.solve(wb)
        return s.last_kernel_kind()

    assert kind(lambda wb: None) == "chain3_lm_kernel"

    def lever(wb): wb.r_val[17, 0, 2:5] = (0.0, 0.0, 1e-300)
    def turned(wb): wb.poses[40, 2, :9] = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]]).reshape(9)
    def rot_info(wb): wb.p_val[9, 0, 15] = 1e-3
    def rot_meas(wb): wb.p_val[9, 0, :9] = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]]).reshape(9)
    def far_pair(wb): wb.r_idx[11, int(wb.counts[11, 1]) - 1] = (4, 1)       # a moving-moving edge that skips poses: not a chain at all
    for mut in (lever, turned, rot_info, rot_meas):
        assert kind(mut) == "chain_lm_kernel", mut.__name__
    assert kind(far_pair) == "window_lm_kernel"
    s.close()


@pytest.mark.parametrize("kernel", ["wave3_lm_kernel", "chain3_lm_kernel", "chain_lm_kernel"])
def test_lane_per_window_kernels_at_the_batch_size_that_selects_them(gpu, kernel):
    """12 288 ten-pose windows of cfg/uwb_only.yaml's topology with the DEFAULT thresholds, no threshold forced: wave3_lm_kernel (what
    such a batch takes), chain3_lm_kernel (with LOCAMD_WAVE3=0: what it took before, and what windows of more than 64 poses take) and
    chain_lm_kernel (with the 3-DoF kernels switched off): same-mode parity on a 512-window sample, the cross-mode bound (analytic
    kernel vs the numeric oracle = the reference's configuration) on 2 048, and the numeric kernel against the numeric oracle."""
    import sys
    import localization_amd as la
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "perf"))
    import bench_window as bw
    B, n_distinct = 12288, 2048
    small, graphs, anchors, T = bw.build(n_distinct, "uwb_only", seed=99)
    wb = la.WindowBatch(B, *small.caps)
    for name in ("counts", "poses", "r_idx", "r_val", "p_idx", "p_val", "s_idx", "s_val"):
        src = getattr(small, name)
        getattr(wb, name)[:] = np.resize(src, (B,) + src.shape[1:])
    poses0 = wb.poses.copy()
    out = {}
    for jac in ("analytic", "numeric"):
        s = la.WindowSolver(anchors, B, *wb.caps, maximum_iteration=10, bw_max=1, jacobian=jac)
        if kernel == "chain_lm_kernel":
            s.set_option("chain3", 0)
        if kernel == "chain3_lm_kernel":
            s.set_option("wave3", 0)
        wb.poses[:] = poses0
        s.solve(wb)
        assert s.last_kernel_kind() == kernel
        s.close()
        out[jac] = wb.poses[:, :, 9:].copy()
        assert np.array_equal(out[jac][:n_distinct], out[jac][n_distinct:2 * n_distinct])     # repeated windows, other lanes / waves: same bits
    want_num = bw.oracle_time(graphs, anchors, T, 2048)[1]
    want_ana = bw.oracle_time(graphs, anchors, T, 512, analytic=True)[1]
    same = np.abs(out["analytic"][:512] - want_ana).max(axis=(1, 2))
    assert same.max() < 1e-7 and np.median(same) < 1e-9, (same.max(), np.median(same))
    cross = np.abs(out["analytic"][:2048] - want_num).max(axis=(1, 2))
    assert np.median(cross) < 1e-6 and np.quantile(cross, 0.99) < 1e-4 and cross.max() < 5e-3, (np.median(cross), np.quantile(cross, 0.99), cross.max())
    num = np.abs(out["numeric"][:2048] - want_num).max(axis=(1, 2))
    assert np.median(num) < 1e-7 and np.quantile(num, 0.99) < 1e-5 and num.max() < 5e-3, (np.median(num), np.quantile(num, 0.99), num.max())
