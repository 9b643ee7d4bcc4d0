import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools/dev")
import importlib.util
spec = importlib.util.spec_from_file_location("st", "/root/repo/tools/dev/stress_wave6_se3.py")
src = open("/root/repo/tools/dev/stress_wave6_se3.py").read().split("worst = None")[0]
ns = {"__file__": "/root/repo/tools/dev/stress_wave6_se3.py"}; exec(compile(src, "st", "exec"), ns)
build, copy, la, ANCH = ns["build"], ns["copy"], ns["la"], ns["ANCH"]
B, T = 64, 15
wb = build(B, T, 99)
# adversarial edits
wb.s_val[1, :, 12:] = 0.0                       # zero EdgeSE3 information
wb.r_val[2, :, 1] = 0.0; wb.s_val[2, :, 12:] = 0.0   # nothing at all: H = 0
wb.poses[3, :, 9:] = wb.poses[3, 0, 9:]          # all poses coincide
wb.r_val[4, 0, 0] = np.inf                       # an infinite range
wb.s_val[5, 0, 9] = np.nan                       # a NaN in an EdgeSE3 measurement
wb.r_val[6, :, 0] += 50.0                        # huge residuals
wb.poses[7, :, 9:] += 1e4                        # estimates far away
wb.s_val[8, :, 12:] *= 1e12                      # huge information
wb.s_val[9, :, 12:] *= 1e-12                     # tiny information
ref = copy(wb)
g = la.WindowSolver(ANCH, B, *wb.caps, jacobian=sys.argv[1] if len(sys.argv) > 1 else "analytic", chain_threshold=0); rg = g.solve(ref).copy(); print(g.last_kernel_kind()); g.close()
s = la.WindowSolver(ANCH, B, *wb.caps, jacobian=sys.argv[1] if len(sys.argv) > 1 else "analytic"); rs = s.solve(wb).copy(); print(s.last_kernel_kind()); s.close()
for i in range(12):
    nv = int(wb.counts[i, 0])
    a, b = wb.poses[i, :nv], ref.poses[i, :nv]
    fin_a, fin_b = np.isfinite(a).all(), np.isfinite(b).all()
    d = np.abs(a - b).max() if fin_a and fin_b else float("nan")
    print(i, "finite", fin_a, fin_b, "d %.2e" % d, "iters/trials/term", rs[i, 3:6], rg[i, 3:6], "chi %.6g %.6g" % (rs[i, 0], rg[i, 0]))
