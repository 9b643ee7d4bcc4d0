"""phase cycle stamps of window_lm_kernel (the general kernel, `make timing [TIMING_LEVEL=n]`) on ONE cfg/uwb_twist.yaml-shaped window, the way the node
runs it:  LOCALIZATION_AMD_LIB=localization_amd/liblocalization_amd_timing.so python tools/dev/probe_twist_general.py [numeric|analytic]"""
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
print("kernel", s.last_kernel_kind(), f"{s.last_kernel_ms()*1e3:.1f} us")
names = ["set-up (ordering, structure, incidence)", "linearise", "build H, b", "factor phase 1", "factor phase 2", "back-substitution", "update + trial evaluation", "TOTAL"]
for i, nm in enumerate(names): print(f"  {nm:45s} {r[i]:12.0f}  {100 * r[i] / max(r[7], 1):5.1f} %")
