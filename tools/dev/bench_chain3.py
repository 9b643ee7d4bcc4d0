"""A/B timing of the lane-per-window kernels on cfg/uwb_only.yaml's window (dev tool): python tools/dev/bench_chain3.py [B] [T]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))
import localization_amd as la
import bench_window as bw
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
if len(sys.argv) > 2: bw.POSES_OVERRIDE = int(sys.argv[2])
small, graphs, anchors, T = bw.build(2048, "uwb_only", seed=1)
wb = la.WindowBatch(B, *small.caps)
for name in ("counts", "poses", "r_idx", "r_val", "p_idx", "p_val", "s_idx", "s_val"):
    src = getattr(small, name); getattr(wb, name)[:] = np.resize(src, (B,) + src.shape[1:])
for jac in ("analytic", "numeric"):
    s = la.WindowSolver(anchors, B, *wb.caps, maximum_iteration=10, bw_max=1, jacobian=jac)
    s.upload(wb)
    s.solve_resident(); s.solve_resident()
    s.timing_begin(5)
    for _ in range(5): s.solve_resident()
    n, tot, avg = s.timing_end()
    print(f"{jac:9s} {s.last_kernel_kind()} LDS={os.environ.get('LOCAMD_CHAIN3_LDS','auto')} B={B} T={T}: {avg:.3f} ms  {B/avg*1e3:.3e} windows/s", flush=True)
    s.close()
