"""dev: wave6_lm_kernel<JAC, SE3> against the general wave-per-window kernel on thousands of random twist windows (random lengths 1 .. 63, an EdgeSE3
per consecutive pair stored either way round — some missing —, missing range links, ranges with zero information, far-off initial estimates, IMU
priors on some windows, random full information matrices): analytic mode, so the two agree to ~1e-7 unless an LM decision flips on rounding."""
import os, sys
import numpy as np
from scipy.spatial.transform import Rotation
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import localization_amd as la

ANCH = np.array([[3, -3, 0.58], [3, 3, 1.97], [-3, 3, 0.54], [-3, -3, 1.76]], dtype=float)


def build(B, Tmax, seed):
    rng = np.random.default_rng(seed)
    wb = la.WindowBatch(B, Tmax, 3 * Tmax + 4, Tmax, max(Tmax, 1))
    for i in range(B):
        T = int(rng.integers(1, Tmax + 1))
        lever = i % 3 == 0
        truth = np.cumsum(rng.normal(0, 0.05, (T, 3)), axis=0) + np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1.5, 1.5), 1.1])
        est = truth + rng.normal(0, 0.05 if i % 11 else 0.4, (T, 3))
        tR = Rotation.from_rotvec(np.cumsum(rng.normal(0, 0.03, (T, 3)), axis=0) + rng.normal(0, 0.3, 3))
        eR = tR * Rotation.from_rotvec(rng.normal(0, 0.02, (T, 3)))
        off = np.array([0.1, 0.0, -0.05]) if lever else np.zeros(3)
        for k in range(T): wb.add_pose(i, est[k], eR[k].as_matrix())
        for k in range(1, T):
            if rng.random() < 0.04: continue
            a, b = (k - 1, k) if rng.random() < 0.5 else (k, k - 1)
            Zt = tR[a].inv().apply(truth[b] - truth[a]) + rng.normal(0, 0.01, 3)
            ZR = (tR[a].inv() * tR[b] * Rotation.from_rotvec(rng.normal(0, 0.01, 3))).as_matrix()
            A = rng.normal(size=(6, 6)); info = A @ A.T + 6 * np.eye(6); info *= rng.choice([1e2, 1e3, 1e4]) / np.trace(info)
            wb.add_se3(i, a, b, Zt, ZR, info, bool(rng.random() < 0.7))
        for k in range(T):
            for a in rng.choice(4, size=int(rng.integers(1, 3)), replace=False):
                p = truth[k] + tR[k].apply(off)
                info = 0.0 if rng.random() < 0.02 else 1.0 / 0.055 ** 2
                wb.add_range(i, k, int(a), float(np.float32(np.linalg.norm(p - ANCH[a]) + rng.normal(0, 0.03))), info, off, anchor=True)
            if k > 0 and rng.random() > 0.05:
                if rng.random() < 0.5: wb.add_range(i, k - 1, k, 0.0, 1.0 / (5.0 / 32 / 3) ** 2)
                else: wb.add_range(i, k, k - 1, 0.0, 1.0 / (5.0 / 32 / 3) ** 2)
            if i % 4 == 1:
                wb.add_prior(i, k, est[k], (tR[k] * Rotation.from_rotvec(rng.normal(0, 2e-3, 3))).as_matrix(), np.array([0, 0, 0, 1, 1, 1.0]) / 4.592449e-06)
    return wb


def copy(wb):
    o = la.WindowBatch(wb.B, *wb.caps)
    for n in ("counts", "poses", "r_idx", "r_val", "p_idx", "p_val", "s_idx", "s_val"): getattr(o, n)[:] = getattr(wb, n)
    return o


worst = None
for B, Tmax in ((4096, 15), (4096, 31), (2048, 63), (4096, 6)):
    wb = build(B, Tmax, 23 + Tmax)
    ref = copy(wb); orig = copy(wb)
    g = la.WindowSolver(ANCH, B, *wb.caps, jacobian="analytic", chain_threshold=0)
    rg = g.solve(ref).copy(); kg = g.last_kernel_kind(); g.close()
    s = la.WindowSolver(ANCH, B, *wb.caps, jacobian="analytic")
    rs = s.solve(wb).copy(); ks = s.last_kernel_kind(); s.close()
    d = np.zeros(B)
    for i in range(B):
        nv = int(wb.counts[i, 0]); d[i] = np.abs(wb.poses[i, :nv] - ref.poses[i, :nv]).max() if nv else 0.0
    same = (rs[:, 4] == rg[:, 4])
    print(f"{ks} vs {kg}: B={B} Tmax={Tmax}: max |dpose| {d.max():.2e}, median {np.median(d):.1e}, > 1e-6: {(d > 1e-6).sum()}; trial counts differ in {(~same).sum()}, "
          f"terminated differs in {(rs[:, 5] != rg[:, 5]).sum()}, shared-edge counts differ in {(rs[:, 6] != rg[:, 6]).sum()}; among windows with equal trial counts max |dpose| {d[same].max():.2e}, "
          f"chi2 rel {(np.abs(rs[same, 0] - rg[same, 0]) / np.maximum(1.0, np.abs(rg[same, 0]))).max():.1e}; non-finite: {(~np.isfinite(wb.poses)).sum()}")
    if Tmax == 31: worst = (wb, ref, orig, d, rs)

sys.path.insert(0, os.path.join(ROOT, "tests"))
from _oracle_window import oracle_solve_instance
wb, ref, orig, d, rs = worst
for i in np.argsort(d)[-5:]:
    nv = int(wb.counts[i, 0])
    poses, chi, st = oracle_solve_instance(orig, int(i), ANCH)
    print(f"window {i}: nv {nv}, wave6<SE3> vs general {d[i]:.2e}; wave6<SE3> vs oracle {np.abs(wb.poses[i, :nv] - poses).max():.2e}; general vs oracle {np.abs(ref.poses[i, :nv] - poses).max():.2e}; "
          f"trials {int(rs[i, 4])} (oracle {st.lm_trials})")
