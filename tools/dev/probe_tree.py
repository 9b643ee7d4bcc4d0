"""phase cycle stamps of tree_wave_kernel (diagnostic build: make -C localization_amd/csrc treetiming):
LOCALIZATION_AMD_LIB=localization_amd/liblocalization_amd_ttiming.so python tools/dev/probe_tree.py [B] [jacobian]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))
import localization_amd as la
import bench_window as bw
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
jac = sys.argv[2] if len(sys.argv) > 2 else "numeric"
wb, graphs, anchors, T = bw.build_pose64(B, np.random.default_rng(7), n_graphs=0)
s = la.WindowSolver(anchors, B, *wb.caps, maximum_iteration=10, bw_max=8, jacobian=jac)
s.upload(wb)
for _ in range(3):
    s.solve_resident()
s.timing_begin(3)
for _ in range(3):
    s.solve_resident()
n, tot, avg = s.timing_end()
s.download(wb)
r = wb.result.mean(axis=0)
trials = int(r[7] // 1e12); total = r[7] - 1e12 * trials
names = ["linearise: edges", "factor leaves", "leaf-child sums", "factor upper levels", "back-subst", "apply step + loop", "trial scoring", "linearise: hand-over + child sums"]
r = list(r[:8]); r.append(0.0)
r[7] = np.floor(wb.result[:, 6] / 1e9).mean(); r[6] = (wb.result[:, 6] % 1e9).mean()
print(f"B={B} {jac} kind {s.last_kernel_kind()} kernel {avg:.3f} ms; cycles per window {total:.0f}; LM trials {trials}")
for k in range(8): print(f"  {names[k]:20s} {r[k]:10.0f} cycles  {100*r[k]/total:5.1f} %")
