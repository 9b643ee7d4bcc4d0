"""Test helper: solve one instance of a localization_amd.WindowBatch with the CPU oracle's general graph (the checker).
The arrays are read exactly as the C ABI documents them (include/localization_amd.h), so this is also a check of that layout."""
import numpy as np


def oracle_solve_instance(wb, i, anchors, iterations=10, jac_mode=None):
    """Returns (poses [nv][12] as R(9) t(3), chi2, og_stats)."""
    from oracle import oracle as O
    jac_mode = O.JAC_ANALYTIC if jac_mode is None else jac_mode
    nv, nr, np_, ns = (int(x) for x in wb.counts[i])
    G = O.Graph()
    anchors = np.asarray(anchors, dtype=float).reshape(-1, 3)
    for m, a in enumerate(anchors):
        G.add_vertex(m, a, fixed=True)
    base = 1000
    for k in range(nv):
        G.add_vertex(base + k, wb.poses[i, k, 9:], wb.poses[i, k, :9].reshape(3, 3))
    for e in range(nr):
        v0, v1 = int(wb.r_idx[i, e, 0]), int(wb.r_idx[i, e, 1])
        meas, info = wb.r_val[i, e, 0], wb.r_val[i, e, 1]
        o1 = None if getattr(wb, "r_off1", None) is None else wb.r_off1[i, e].copy()
        G.add_range_edge(base + v0, (-1 - v1) if v1 < 0 else base + v1, meas, info, off0=wb.r_val[i, e, 2:5].copy(), off1=o1)
    for e in range(np_):
        Ri = wb.p_val[i, e, :9].reshape(3, 3); ti = wb.p_val[i, e, 9:12]
        G.add_prior_edge(base + int(wb.p_idx[i, e]), -Ri.T @ ti, Ri.T, np.diag(wb.p_val[i, e, 12:18]))
    for e in range(ns):
        Ri = wb.s_val[i, e, :9].reshape(3, 3); ti = wb.s_val[i, e, 9:12]
        G.add_se3_edge(base + int(wb.s_idx[i, e, 0]), base + int(wb.s_idx[i, e, 1]), -Ri.T @ ti, Ri.T,
                       wb.s_val[i, e, 12:].reshape(6, 6), bool(wb.s_idx[i, e, 2]))
    n, st = G.optimize(iterations, jac_mode)
    out = np.zeros((nv, 12))
    for k in range(nv):
        R, t = G.estimate(base + k)
        out[k, :9] = R.reshape(9); out[k, 9:] = t
    return out, G.chi2(), st
