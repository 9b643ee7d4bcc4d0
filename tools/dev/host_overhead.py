"""Host-side cost of loc_window_solve_host at B = 65 536 ten-pose windows (cfg/uwb_only.yaml's window as a batch): validation, structure analysis
(first call: full scan; second call with the same structure and new measurements: one hash pass + the value-dependent checks), staging + kernel."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))
import localization_amd as la
import bench_window as bw
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
small, graphs, anchors, T = bw.build(4096, "uwb_only", seed=3)
wb = la.WindowBatch(B, *small.caps)
for name in ("counts", "poses", "r_idx", "r_val", "p_idx", "p_val", "s_idx", "s_val"):
    src = getattr(small, name); getattr(wb, name)[:] = np.resize(src, (B,) + src.shape[1:])
poses0 = wb.poses.copy()
s = la.WindowSolver(anchors, B, *wb.caps, maximum_iteration=10, bw_max=1, jacobian="numeric")
for label, cache in (("cache on", 1), ("cache off", 0)):
    s.set_option("topology_cache", cache)
    for rep in range(3):
        wb.poses[:] = poses0
        wb.r_val[:, :, 0] += 1e-4          # other measurements, same structure
        t0 = time.perf_counter(); s.solve(wb); dt = (time.perf_counter() - t0) * 1e3
        v, t, r, c = s.last_host_timing()
        print(f"{label} call {rep}: total {dt:7.2f} ms = validate {v:6.2f} + structure {t:6.2f} + stage/launch/copy {r:7.2f} (kernel {s.last_kernel_ms():.2f}); verdict cached: {c}; kernel {s.last_kernel_kind()}")
