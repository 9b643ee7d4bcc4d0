"""phase cycle stamps of arrow3_lm_kernel (diagnostic build -DLOCAMD_ARROW_TIMING): LOCALIZATION_AMD_LIB=..._atiming.so python tools/dev/probe_arrow3.py [B]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))
import localization_amd as la
import bench_window as bw
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
small, graphs, anchors, nv = bw.build_selfcal(min(B, 8), np.random.default_rng(11))
wb = la.WindowBatch(B, *small.caps)
for name in ("counts", "poses", "r_idx", "r_val", "p_idx", "p_val", "s_idx", "s_val"):
    src = getattr(small, name); getattr(wb, name)[:] = np.resize(src, (B,) + src.shape[1:])
s = la.WindowSolver(anchors, B, *wb.caps, maximum_iteration=10, bw_max=nv - 1, jacobian=sys.argv[2] if len(sys.argv) > 2 else "analytic")
s.solve(wb); r = wb.result.mean(axis=0)
names = ["", "linearise", "forward+schur", "dense", "B^T x", "chain z,x", "update", "trial eval"] if os.environ.get("ARROW_LEVEL", "1") == "1" else ["", "lin: tail of the chunk loop", "lin: PK += GZ", "lin: CS barrier", "lin: sums", "lin: CS loads+adds", "lin: CS wave sums", "lin: slot loop"]
tot = r[1:8].sum()
print(f"B={B} kernel {s.last_kernel_ms():.3f} ms; cycles per instance {tot:.0f}")
for k in range(1, 8): print(f"  {names[k]:28s} {r[k]:10.0f} cycles  {100*r[k]/tot:5.1f} %")
