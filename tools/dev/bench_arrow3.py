"""A/B timing of BASELINE config 4 (anchor self-calibration, 256 + 10 poses) through arrow3 / the general kernel (dev tool)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))
import localization_amd as la
import bench_window as bw
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n_distinct = min(B, 16)
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256   # chain poses (cfg4: 256)
small, graphs, anchors, nv = bw.build_selfcal(n_distinct, np.random.default_rng(11), T=T)
wb = la.WindowBatch(B, *small.caps)
for name in ("counts", "poses", "r_idx", "r_val", "p_idx", "p_val", "s_idx", "s_val"):
    src = getattr(small, name); getattr(wb, name)[:] = np.resize(src, (B,) + src.shape[1:])
for jac in ("analytic", "numeric"):
    s = la.WindowSolver(anchors, B, *wb.caps, maximum_iteration=10, bw_max=nv - 1, jacobian=jac)
    s.upload(wb)
    s.solve_resident()
    s.timing_begin(3)
    for _ in range(3): s.solve_resident()
    n, tot, avg = s.timing_end()
    s.download(wb)
    print(f"{jac:9s} {s.last_kernel_kind()} B={B}: {avg:.3f} ms  {B/avg*1e3:.3e} solves/s  mean trials {wb.result[:,4].mean():.2f}", flush=True)
    s.close()
