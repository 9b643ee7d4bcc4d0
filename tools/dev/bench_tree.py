"""A/B timing of BASELINE config 5 (64-pose key-frame windows) through tree_lm_kernel / the general kernel (dev tool)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))
import localization_amd as la
import bench_window as bw
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
wb, graphs, anchors, T = bw.build_pose64(B, np.random.default_rng(7), n_graphs=0)
for jac in ("analytic", "numeric"):
    s = la.WindowSolver(anchors, B, *wb.caps, maximum_iteration=10, bw_max=8, jacobian=jac)
    s.upload(wb)
    s.solve_resident()
    s.timing_begin(3)
    for _ in range(3): s.solve_resident()
    n, tot, avg = s.timing_end()
    s.download(wb)
    print(f"{jac:9s} {s.last_kernel_kind()} B={B}: {avg:.3f} ms  {B/avg*1e3:.3e} windows/s  mean trials {wb.result[:,4].mean():.2f}", flush=True)
    s.close()
