import sys; sys.path.insert(0,"tests/perf"); sys.path.insert(0,".")
import numpy as np, bench_window as bw, localization_amd as la
for shape in ("uwb_only","uwb_imu"):
    B=8192
    wb, graphs, anchors, T = bw.build(B, shape, seed=11)
    outs=[]
    for th in (0,1):
        w2 = la.WindowBatch(B, *wb.caps)
        for name in ("counts","poses","r_idx","r_val","p_idx","p_val","s_idx","s_val"):
            getattr(w2,name)[:] = getattr(wb,name)
        s = la.WindowSolver(anchors, B, *wb.caps, maximum_iteration=10, bw_max=1, chain_threshold=th, jacobian="analytic")
        res = s.solve(w2).copy(); s.close()
        outs.append((w2.poses.copy(), res))
    d = np.abs(outs[0][0]-outs[1][0]).reshape(B,-1).max(axis=1)
    print(shape, "max diff", d.max(), "median", np.median(d), "n>1e-6", int((d>1e-6).sum()), "trial mismatch", int((outs[0][1][:,4]!=outs[1][1][:,4]).sum()), "kernel sig", outs[1][1][0,7], outs[0][1][0,7])
