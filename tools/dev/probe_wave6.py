"""phase cycle stamps of wave6_lm_kernel (diagnostic build: make -C localization_amd/csrc wave6timing):
LOCALIZATION_AMD_LIB=localization_amd/liblocalization_amd_timing.so python tools/dev/probe_wave6.py [B] [jacobian]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))
import localization_amd as la
import bench_window as bw
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
wb, graphs, anchors, T = bw.build(B, "uwb_imu", seed=3)
s = la.WindowSolver(anchors, B, *wb.caps, maximum_iteration=10, bw_max=1, jacobian=sys.argv[2] if len(sys.argv) > 2 else "numeric")
for _ in range(3):
    w2 = la.WindowBatch(B, *wb.caps)
    for name in ("counts", "poses", "r_idx", "r_val", "p_idx", "p_val", "s_idx", "s_val"): getattr(w2, name)[:] = getattr(wb, name)
    s.solve(w2)
r = w2.result.mean(axis=0)
trials = int(r[7] // 1e12); total = r[7] - 1e12 * trials
names = ["set-up", "edges (lin.)", "gather", "factor+solves", "recurrences+x", "apply+trial", "LM update"]
print(f"B={B} T={T} kind {s.last_kernel_kind()} kernel {s.last_kernel_ms()*1e3:.1f} us; cycles per window {total:.0f}; LM trials {trials}")
for k in range(7): print(f"  {names[k]:14s} {r[k]:10.0f} cycles  {100*r[k]/total:5.1f} %")
