"""The HIP paths against the committed golden fixtures directly (tests/golden/scipy_minima.npz: minima of the same robust
objective found by scipy, not by the oracle; tests/golden/bag_example.npz is used by test_gpu_node_parity.py)."""
import os

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "scipy_minima.npz"))


def _polish_matches(i, got, A, d, s):
    """LM and scipy's trust region may settle in different basins of the multi-modal robust cost; then the kernel's answer
    must itself be a minimiser (scipy restarted from it stays put)."""
    from scipy.optimize import least_squares
    def f(p):
        r = (d.astype(float) - np.linalg.norm(p[None] - A, axis=1)) / s.astype(float)
        return np.sign(r) * np.sqrt(np.log1p(r * r))
    pol = least_squares(f, got, xtol=1e-15, ftol=1e-15, gtol=1e-15)
    return np.abs(pol.x - got).max() < 1e-6


@pytest.mark.parametrize("jac", ["analytic", "numeric"])
def test_snapshot_kernel_reaches_scipy_minima(gpu, gold, jac):
    import localization_amd as la
    A = gold["a_anchors"]; N = gold["a_dist"].shape[0]
    solver = la.SnapshotSolver(A, N, maximum_iteration=400, distance_outlier=0.0, jacobian=jac)
    solver.set_positions(gold["a_init"].T.copy())
    pos, chi2, trials = solver.solve(gold["a_dist"].T[None], gold["a_err"].T[None])
    solver.close()
    got = pos[0].T
    same = 0
    for i in range(N):
        if np.abs(got[i] - gold["a_min"][i]).max() < 1e-6:
            same += 1
        else:
            assert _polish_matches(i, got[i], A, gold["a_dist"][i], gold["a_err"][i]), i
    assert same >= int(0.9 * N), same


def test_fusion_kernel_reaches_scipy_minima(gpu, gold):
    """6-DoF golden: lever arm + rotation-only prior with the bag's IMU covariance."""
    import localization_amd as la
    A = gold["b_anchors"]; N = gold["b_dist"].shape[0]; cov = float(gold["b_cov"])
    f = la.FusionSolver(A, N, antenna_offset=gold["b_offset"], maximum_iteration=300, distance_outlier=0.0, jacobian="analytic")
    init = np.zeros((7, N)); init[:3] = gold["b_init_t"].T; init[6] = 1.0
    f.set_poses(init)
    imu = np.zeros((1, N, 8)); imu[0, :, :4] = gold["b_imu_q_xyzw"]; imu[0, :, 4:7] = cov
    pose, chi2, trials = f.solve(gold["b_dist"].T[None], gold["b_err"].T[None], imu)
    f.close()
    assert np.abs(pose[0, :3].T - gold["b_min_t"]).max() < 2e-6
    dq = (Rotation.from_quat(pose[0, 3:7].T).inv() * Rotation.from_quat(gold["b_min_q_xyzw"])).magnitude()
    assert dq.max() < 2e-6


def test_window_kernel_reaches_scipy_minima(gpu, gold):
    """5-pose windows in the reference's topology (one range per pose + zero-range smoothness edges)."""
    import localization_amd as la
    A = gold["c_anchors"]; sig_v = float(gold["c_sigma_v"]); N, T = gold["c_dist"].shape
    anchors = np.concatenate([A, np.zeros((N, 3))])            # one extra fixed vertex per instance: the pose before the window
    wb = la.WindowBatch(N, T, 2 * T + 2, 0, 0)
    for i in range(N):
        anchors[4 + i] = gold["c_prev"][i]
        for k in range(T):
            wb.add_pose(i, gold["c_init"][i, k])
        for k in range(T):
            wb.add_range(i, k, int(gold["c_anchor_idx"][i, k]), float(gold["c_dist"][i, k]), 1.0 / 0.055 ** 2, anchor=True)
            if k == 0:
                wb.add_range(i, 0, 4 + i, 0.0, 1.0 / sig_v ** 2, anchor=True)
            else:
                wb.add_range(i, k - 1, k, 0.0, 1.0 / sig_v ** 2)
        for j, m in enumerate((1, 2)):
            wb.add_range(i, T - 1, m, float(gold["c_extra"][i, j]), 1.0 / 0.055 ** 2, anchor=True)
    solver = la.WindowSolver(anchors, N, T, 2 * T + 2, 0, 0, maximum_iteration=500, jacobian="analytic")
    solver.solve(wb)
    solver.close()
    got = wb.poses[:, :, 9:]
    assert np.abs(got - gold["c_min"]).max() < 5e-5, np.abs(got - gold["c_min"]).max()
