"""phase cycle stamps of tree_wave_kernel on ONE cfg/uwb_twist.yaml-shaped window (a 15-pose chain: anchor range + smoothness range + twist EdgeSE3 per
pose), the way the node's handle (batch capacity 1) runs it:  LOCALIZATION_AMD_LIB=localization_amd/liblocalization_amd_ttiming.so python tools/dev/probe_tree_chain.py"""
import os, sys
import numpy as np
from scipy.spatial.transform import Rotation
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import localization_amd as la
ANCH = np.array([[3, -3, 0.58], [3, 3, 1.97], [-3, 3, 0.54], [-3, -3, 1.76]], dtype=float)
T, iters = 15, 12
rng = np.random.default_rng(5)
tt = np.cumsum(rng.normal(0, 0.02, (T, 3)), axis=0) + np.array([0.3, -0.2, 1.0])
tR = Rotation.from_rotvec(np.cumsum(rng.normal(0, 0.01, (T, 3)), axis=0))
wb = la.WindowBatch(1, T, 2 * T, 0, T)
for k in range(T): wb.add_pose(0, tt[k] + rng.normal(0, 0.03, 3), (tR[k] * Rotation.from_rotvec(rng.normal(0, 0.01, 3))).as_matrix())
for k in range(T):
    if k > 0:
        wb.add_se3(0, k - 1, k, tR[k - 1].inv().apply(tt[k] - tt[k - 1]) + rng.normal(0, 0.002, 3), (tR[k - 1].inv() * tR[k]).as_matrix(), np.eye(6) * 1e4, True)
        wb.add_range(0, k - 1, k, 0.0, 1 / (1.0 / 60 / 3) ** 2)
    wb.add_range(0, k, k % 4, float(np.float32(np.linalg.norm(tt[k] - ANCH[k % 4]) + rng.normal(0, 0.03))), 1 / 0.055 ** 2, anchor=True)
p0 = wb.poses.copy()
s = la.WindowSolver(ANCH, 1, *wb.caps, maximum_iteration=iters, bw_max=1, jacobian=sys.argv[1] if len(sys.argv) > 1 else "numeric")
for _ in range(3):
    wb.poses[:] = p0
    s.solve(wb)
r = wb.result[0]
print("kernel", s.last_kernel_kind(), f"{s.last_kernel_ms()*1e3:.1f} us", "host timing", s.last_host_timing())
if r[7] > 1e11 and s.last_kernel_kind().startswith("wave6"):   # (make wave6timing)
    trials = int(r[7] // 1e12); total = r[7] - 1e12 * trials
    names = ["set-up", "edges + EdgeSE3 (lin.)", "gather", "(rank-1 factor)", "block Cholesky + x", "apply + trial scoring", "LM update"]
    print(f"cycles {total:.0f}; LM trials {trials}")
    for k in range(7): print(f"  {names[k]:34s} {r[k]:10.0f} cycles  {100*r[k]/total:5.1f} %")
elif r[7] > 1e11:
    trials = int(r[7] // 1e12); total = r[7] - 1e12 * trials
    names = ["linearise: edges", "factor leaves", "leaf-child sums", "factor upper levels", "back-subst", "apply step + loop", "trial scoring", "linearise: hand-over + child sums"]
    v = list(r[:6]) + [r[6] % 1e9, np.floor(r[6] / 1e9)]
    print(f"cycles {total:.0f}; LM trials {trials}")
    for k in range(8): print(f"  {names[k]:34s} {v[k]:10.0f} cycles  {100*v[k]/total:5.1f} %")
else:
    print("result", r)
