#!/usr/bin/env python3
"""Diagnostic: cycle shares of the window kernel's phases.  Needs the timing build of the library:
    make -C localization_amd/csrc timing          (-> localization_amd/liblocalization_amd_timing.so)
    python tests/perf/probe_window_phases.py SHAPE [BATCH] [--natural]
Never quote this build's run time (the stamps serialise the phases); read the SHARES.  The result record of this build holds
the phase counters instead of chi2 etc. (window_kernel.hip, LOCAMD_WINDOW_TIMING)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
shape = sys.argv[1] if len(sys.argv) > 1 else "uwb_only"
NB = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else 64   # 64 = unloaded GPU, thousands = under load
natural = "--natural" in sys.argv
import localization_amd._lib as _lib
_lib._SO = os.path.join(ROOT, "localization_amd", "liblocalization_amd_timing.so")
import localization_amd as la
sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))
import bench_window as bw
wb, graphs, anchors, T = bw.build(NB, shape)
bwm = 0
for b in range(min(NB, 64)):
    nr, ns = int(wb.counts[b, 1]), int(wb.counts[b, 3])
    r = wb.r_idx[b, :nr]; pp = r[r[:, 1] >= 0]
    if len(pp): bwm = max(bwm, int(np.abs(pp[:, 0] - pp[:, 1]).max()))
    if ns: bwm = max(bwm, int(np.abs(wb.s_idx[b, :ns, 0] - wb.s_idx[b, :ns, 1]).max()))
s = la.WindowSolver(anchors, NB, *wb.caps, maximum_iteration=10, bw_max=bwm, natural_order=natural, jacobian="analytic")
s.solve(wb)
r = wb.result.mean(axis=0)
names = ["set-up (ordering, structure, incidence)", "linearise", "build H, b", "factor phase 1 (SKYLINE: whole sweep)", "factor phase 2",
         "back-substitution", "update + trial evaluation", "TOTAL"]
if "--sub" in sys.argv:   # library built with `make timing TIMING_LEVEL=2`
    names[0], names[1], names[2], names[6] = ("column mode: index + loads", "column mode: updates from earlier columns",
                                              "column mode: 6x6 Cholesky + right-hand side", "column mode: off-diagonal blocks")
if "--sub3" in sys.argv:   # library built with `make timing TIMING_LEVEL=3`
    names[0], names[1], names[2], names[6] = ("back-substitution: index look-ups + y", "back-substitution: blocks of the column",
                                              "back-substitution: triangular solve", "back-substitution: stores + barrier")
if "--sub4" in sys.argv:   # library built with `make timing TIMING_LEVEL=4`
    names[0], names[1], names[2], names[6] = ("factor: column-mode levels", "factor: row-mode levels without dense blocks",
                                              "factor: dense levels, phase 1 (cooperative sums)", "factor: dense levels, phase 2")
if "--sub5" in sys.argv:   # library built with `make timing TIMING_LEVEL=5`
    names[0], names[1], names[2], names[6] = ("set-up: ordering rounds", "set-up: relabelling, structure, levels, dense list",
                                              "set-up: incidence lists + shared pairs", "set-up: staging, edge relabelling")
print(f"{shape} x {NB}{' natural order' if natural else ''}: cycles per solve (lane-0 stamps)")
for i, nm in enumerate(names):
    print(f"  {nm:45s} {r[i]:12.0f}  {100 * r[i] / r[7]:5.1f} %")
