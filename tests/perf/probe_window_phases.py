#!/usr/bin/env python3
"""Diagnostic: cycle shares of the window kernel's phases.  Needs a build with -DLOCAMD_WINDOW_TIMING:
  hipcc ... -DLOCAMD_WINDOW_TIMING -c window_kernel.hip   (link into a separate .so, pass it as argv[1])
Never quote this build's run time (the stamps serialise the phases); read the SHARES."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
so = sys.argv[1]
shape = sys.argv[2] if len(sys.argv) > 2 else "uwb_only"
import localization_amd._lib as _lib
_lib._SO = os.path.abspath(so)
import localization_amd as la
sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))
import bench_window as bw
NB = int(sys.argv[4]) if len(sys.argv) > 4 else 64   # instances (argv[4]): 64 = unloaded GPU, thousands = under load
wb, graphs, anchors, T = bw.build(NB, shape)
bwm = int(sys.argv[3]) if len(sys.argv) > 3 else -1   # pose band (argv[3]); -1 = dense
s = la.WindowSolver(anchors, NB, *wb.caps, maximum_iteration=10, bw_max=bwm)
s.solve(wb)
r = wb.result
fs = np.floor(r[:, 6] / 1e6); ev = (r[:, 6] - fs * 1e6) * 1e3
bd = np.floor(r[:, 7] / 1e6); tot = (r[:, 7] - bd * 1e6) * 1e3
print(f"{shape}: trials {r[:, 4].mean():.1f} iters {r[:, 3].mean():.1f}; cycles per solve: total {tot.mean():.0f}; "
      f"factor+solve {fs.mean():.0f} ({(fs / r[:, 4]).mean():.0f}/trial, {100 * fs.mean() / tot.mean():.0f} %); "
      f"trial errors {ev.mean():.0f} ({(ev / r[:, 4]).mean():.0f}/trial, {100 * ev.mean() / tot.mean():.0f} %); "
      f"linearise+build {bd.mean():.0f} ({(bd / r[:, 3]).mean():.0f}/iteration, {100 * bd.mean() / tot.mean():.0f} %)")
print(f"  inside factor+solve (cycles per solve): (a) band segments {r[:, 0].mean():.0f}, (b) block exchange + 6x6 factor {r[:, 1].mean():.0f}, "
      f"(c) row finish {r[:, 2].mean():.0f}, back-substitution {r[:, 5].mean():.0f}")
