#!/usr/bin/env python3
"""Secondary measurements of the sliding-window path (not the headline bench): windows/s of the general graph kernel
on synthetic instances in the reference's topology, next to the oracle on one host core.

    python tests/perf/bench_window.py --shape uwb_only   # T = 10, ranges + smoothness        (cfg/uwb_only.yaml shape)
    python tests/perf/bench_window.py --shape uwb_imu    # T = 12, + IMU priors + lever arm   (cfg/uwb_imu.yaml shape)
    python tests/perf/bench_window.py --shape fusion1    # 1 pose, 8 ranges + IMU prior       (BASELINE config 3 shape)
    python tests/perf/bench_window.py --shape selfcal --batch 128   # cfg4 shape: 10 unknown anchors x 256 timesteps per hypothesis
    python tests/perf/bench_window.py --shape pose64 --batch 1024   # 64-pose window, key-frame EdgeSE3 star + one range per pose
                                                               # (BASELINE config 5 shape; matrix in the HBM workspace)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
from scipy.spatial.transform import Rotation

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

ANCH4 = np.array([[3, -3, 0.58], [3, 3, 1.97], [-3, 3, 0.54], [-3, -3, 1.76]], dtype=float)


POSES_OVERRIDE = None   # --poses: window length for the chain shapes (uwb_only / uwb_imu)


def build(B, shape, seed=0):
    import localization_amd as la
    from localization_amd.synthetic import ANCHORS_8
    rng = np.random.default_rng(seed)
    if shape == "pose64":
        return build_pose64(B, rng)
    if shape == "selfcal":
        return build_selfcal(B, rng)
    if shape == "fusion1":
        T, anchors, imu, lever, per_pose = 1, ANCHORS_8, True, True, 8
    elif shape == "uwb_imu":
        T, anchors, imu, lever, per_pose = 12, ANCH4, True, True, 1
    elif shape == "uwb_twist":   # cfg/uwb_twist.yaml: trajectory_length 15, a twist EdgeSE3 between consecutive poses (addTwistEdge)
        T, anchors, imu, lever, per_pose = 15, ANCH4, False, False, 1
    else:
        T, anchors, imu, lever, per_pose = 10, ANCH4, False, False, 1
    if POSES_OVERRIDE and T > 1:
        T = POSES_OVERRIDE
    off = np.array([0.1, 0.0, -0.05]) if lever else np.zeros(3)
    twist = shape == "uwb_twist"
    wb = la.WindowBatch(B, T, max(2 * T, per_pose), T if imu else 0, T if twist else 0)
    graphs = []
    for i in range(B):
        tt = np.cumsum(rng.normal(0, 0.05, (T, 3)), axis=0) + np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1.5, 1.5), 1.1])
        tR = Rotation.from_rotvec(np.cumsum(rng.normal(0, 0.03, (T, 3)), axis=0) + rng.normal(0, 0.3, 3))
        et = tt + rng.normal(0, 0.05, (T, 3)); eR = (tR * Rotation.from_rotvec(rng.normal(0, 0.02, (T, 3)))).as_matrix()
        if shape == "uwb_only":
            # cfg/uwb_only.yaml has no rotation source: Robot::init gives every pose the identity rotation (robot.cpp:47) and, with the
            # default identity antenna offsets (localization.h:170), nothing ever turns it
            eR = np.tile(np.eye(3), (T, 1, 1))
        g = dict(et=et, eR=eR, off=off, ranges=[], smooth=[], priors=[], se3=[])
        for k in range(T):
            wb.add_pose(i, et[k], eR[k])
            if twist and k:
                Zt = tR[k - 1].inv().apply(tt[k] - tt[k - 1]) + rng.normal(0, 0.002, 3)
                ZR = (tR[k - 1].inv() * tR[k] * Rotation.from_rotvec(rng.normal(0, 0.002, 3))).as_matrix()
                wb.add_se3(i, k - 1, k, Zt, ZR, np.eye(6) * 1e4, True); g["se3"].append((k - 1, k, Zt, ZR))
            for a in (range(per_pose) if per_pose > 1 else [int(rng.integers(0, len(anchors)))]):
                d = float(np.float32(np.linalg.norm(tt[k] + tR[k].apply(off) - anchors[a]) + rng.normal(0, 0.03)))
                wb.add_range(i, k, a, d, 1 / 0.055 ** 2, off, anchor=True); g["ranges"].append((k, a, d))
            if k:
                wb.add_range(i, k - 1, k, 0.0, 1 / (5.0 / 32 / 3) ** 2); g["smooth"].append((k - 1, k))
            if imu:
                R = (tR[k] * Rotation.from_rotvec(rng.normal(0, 2e-3, 3))).as_matrix()
                dg = np.array([0, 0, 0, 1, 1, 1.0]) / 4.592449e-06
                wb.add_prior(i, k, et[k], R, dg); g["priors"].append((k, et[k].copy(), R, dg))
        graphs.append(g)
    return wb, graphs, anchors, T


def build_pose64(B, rng, T=64, n_graphs=512):
    """cfg5: per pose one range (round-robin anchor) + one EdgeSE3 from the current key pose (new key
    every 8 poses), Omega = diag(1e4), Cauchy.  Vectorised over the batch (the full B = 16 384 builds in seconds); the
    oracle-side description is kept for the first n_graphs instances."""
    import localization_amd as la
    wb = la.WindowBatch(B, T, T, 0, T)
    tt = np.cumsum(rng.normal(0, 0.05, (B, T, 3)), axis=1) + np.stack([rng.uniform(-1.5, 1.5, B), rng.uniform(-1.5, 1.5, B), np.full(B, 1.1)], 1)[:, None, :]
    rv = np.cumsum(rng.normal(0, 0.03, (B, T, 3)), axis=1) + rng.normal(0, 0.3, (B, 1, 3))
    tR = Rotation.from_rotvec(rv.reshape(-1, 3))
    et = tt + rng.normal(0, 0.05, (B, T, 3))
    eR = (tR * Rotation.from_rotvec(rng.normal(0, 0.02, (B * T, 3)))).as_matrix().reshape(B, T, 3, 3)
    tRm = tR.as_matrix().reshape(B, T, 3, 3)
    k = np.arange(T)
    anch = k % 4
    d = (np.linalg.norm(tt - ANCH4[anch][None], axis=2) + rng.normal(0, 0.03, (B, T))).astype(np.float32).astype(np.float64)
    key = np.where(k < 8, 0, (k // 8) * 8 - 1)
    Rk = tRm[:, key]                                                    # [B, T, 3, 3] rotation of each pose's key
    Zt = np.einsum("btji,btj->bti", Rk, tt - tt[:, key]) + rng.normal(0, 0.01, (B, T, 3))
    ZR = np.einsum("btji,btjk->btik", Rk, tRm) @ Rotation.from_rotvec(rng.normal(0, 0.01, (B * T, 3))).as_matrix().reshape(B, T, 3, 3)
    wb.counts[:] = (T, T, 0, T - 1)
    wb.poses[:, :, :9] = eR.reshape(B, T, 9); wb.poses[:, :, 9:] = et
    wb.r_idx[:, :, 0] = k[None]; wb.r_idx[:, :, 1] = -1 - anch[None]
    wb.r_val[:, :, 0] = d; wb.r_val[:, :, 1] = 1 / 0.055 ** 2; wb.r_val[:, :, 2:] = 0.0
    wb.s_idx[:, : T - 1, 0] = key[1:][None]; wb.s_idx[:, : T - 1, 1] = k[1:][None]; wb.s_idx[:, : T - 1, 2] = 1
    ZRi = np.swapaxes(ZR[:, 1:], -1, -2)                                # inverse measurement
    wb.s_val[:, : T - 1, :9] = ZRi.reshape(B, T - 1, 9)
    wb.s_val[:, : T - 1, 9:12] = -np.einsum("btij,btj->bti", ZRi, Zt[:, 1:])
    wb.s_val[:, : T - 1, 12:] = (np.eye(6) * 1e4).reshape(36)
    graphs = []
    for i in range(min(B, n_graphs)):
        graphs.append(dict(et=et[i], eR=eR[i], off=np.zeros(3), smooth=[], priors=[],
                           ranges=[(int(kk), int(anch[kk]), float(d[i, kk])) for kk in range(T)],
                           se3=[(int(key[kk]), int(kk), Zt[i, kk], ZR[i, kk]) for kk in range(1, T)]))
    return wb, graphs, ANCH4, T


def build_selfcal(B, rng, T=256, A=10):
    """cfg4: anchor self-calibration. Every node moves (localization.cpp:94-98): A unknown anchors, one
    tag with T timesteps, A ranges per timestep, smoothness edges along the tag; each instance is one Monte-Carlo
    hypothesis = a seeded perturbation (sigma 1 m) of the anchor initialisation, held by a weak position prior.
    Tag poses take slots 0..T-1 (time order), anchors the LAST A slots, so the skyline is an arrowhead."""
    import localization_amd as la
    true_anchors = np.column_stack([rng.uniform(-4, 4, A), rng.uniform(-4, 4, A), rng.uniform(0, 3, A)])
    wb = la.WindowBatch(B, T + A, T * A + T, A, 0)
    graphs = []
    tt = np.cumsum(rng.normal(0, 0.05, (T, 3)), axis=0) + np.array([0.0, 0.0, 1.2])
    d_true = np.linalg.norm(tt[:, None, :] - true_anchors[None], axis=2) + rng.normal(0, 0.03, (T, A))
    pinfo = np.array([1.0, 1.0, 1.0, 0, 0, 0])
    for i in range(B):
        hyp = true_anchors + rng.normal(0, 1.0, true_anchors.shape)
        et = tt + rng.normal(0, 0.05, tt.shape)
        g = dict(et=et, hyp=hyp, ranges=[], smooth=[])
        for k in range(T): wb.add_pose(i, et[k])
        for a in range(A):
            wb.add_pose(i, hyp[a]); wb.add_prior(i, T + a, hyp[a], np.eye(3), pinfo)
        for k in range(T):
            for a in range(A):
                d = float(np.float32(d_true[k, a])); wb.add_range(i, k, T + a, d, 1 / 0.055 ** 2); g["ranges"].append((k, a, d))
            if k: wb.add_range(i, k - 1, k, 0.0, 1 / (5.0 / 32 / 3) ** 2); g["smooth"].append((k - 1, k))
        graphs.append(g)
    return wb, graphs, np.zeros((0, 3)), T + A


def oracle_selfcal(g, T, A, iters=10):
    from oracle import oracle as O
    G = O.Graph()
    for k in range(T): G.add_vertex(100 + k, g["et"][k])
    info = np.diag([1.0, 1.0, 1.0, 0, 0, 0])
    for a in range(A):
        G.add_vertex(100 + T + a, g["hyp"][a]); G.add_prior_edge(100 + T + a, g["hyp"][a], np.eye(3), info)
    for (k, a, d) in g["ranges"]: G.add_range_edge(100 + k, 100 + T + a, d, 1 / 0.055 ** 2)
    for (k0, k1) in g["smooth"]: G.add_range_edge(100 + k0, 100 + k1, 0.0, 1 / (5.0 / 32 / 3) ** 2)
    G.optimize(iters, O.JAC_NUMERIC_G2O)
    return np.array([G.estimate(100 + k)[1] for k in range(T + A)])


def oracle_time(graphs, anchors, T, n, iters=10, analytic=False):
    from oracle import oracle as O
    t0 = time.perf_counter()
    out = []
    if graphs and "hyp" in graphs[0]:
        A = len(graphs[0]["hyp"])
        for g in graphs[:n]:
            out.append(oracle_selfcal(g, T - A, A, iters))
        return time.perf_counter() - t0, np.array(out)
    for g in graphs[:n]:
        G = O.Graph()
        for m, a in enumerate(anchors): G.add_vertex(m, a, fixed=True)
        for k in range(T): G.add_vertex(100 + k, g["et"][k], g["eR"][k])
        for (k, a, d) in g["ranges"]: G.add_range_edge(100 + k, a, d, 1 / 0.055 ** 2, off0=g["off"])
        for (k0, k1) in g["smooth"]: G.add_range_edge(100 + k0, 100 + k1, 0.0, 1 / (5.0 / 32 / 3) ** 2)
        for (k, t, R, dg) in g["priors"]: G.add_prior_edge(100 + k, t, R, np.diag(dg))
        for (ki, kj, t, R) in g.get("se3", []): G.add_se3_edge(100 + ki, 100 + kj, t, R, np.eye(6) * 1e4, True)
        G.optimize(iters, O.JAC_ANALYTIC if analytic else O.JAC_NUMERIC_G2O)
        out.append(np.array([G.estimate(100 + k)[1] for k in range(T)]))
    return time.perf_counter() - t0, np.array(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="uwb_only", choices=["uwb_only", "uwb_imu", "uwb_twist", "fusion1", "pose64", "selfcal"])
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cpu-n", type=int, default=256)
    ap.add_argument("--bw", default="auto", help="'auto' = the widest pose-pose coupling in the batch (what the node front-end passes), "
                    "'dense' = nv_max - 1, or a number")
    ap.add_argument("--poses", type=int, default=0, help="window length for the chain shapes (default: the profile's 10 / 12)")
    ap.add_argument("--jacobian", default="analytic", choices=["analytic", "numeric"])
    ap.add_argument("--chain-threshold", type=int, default=-1, help="loc_window_set_chain_threshold (A/B runs: 1 = the lane-per-window chain kernels for every chain batch, 0 = the general kernel)")
    ap.add_argument("--natural", action="store_true", help="windows of <= 64 poses: keep the caller's pose order (no in-kernel minimum-degree ordering)")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-window latency launches (profiling: one kernel shape only)")
    ap.add_argument("--pmc-json", default=None, help="a tests/perf/pmc_summary.py output for THIS shape and batch: its f64 instruction counts "
                    "(wave-instructions per launch, x 64 lanes; fma = 2 flop) give the f64 rate next to the HBM roofline")
    ap.add_argument("--tile", type=int, default=0, help="build this many distinct windows and repeat them cyclically up to --batch (selfcal: building 1 024 "
                    "hypotheses of 2 815 edges in Python takes a minute)")
    ap.add_argument("--cache", default=None, help="npz file: load the generated batch from it if it exists, else build and save "
                    "(profiling runs repeat the same command once per counter pass)")
    a = ap.parse_args()
    import localization_amd as la
    global POSES_OVERRIDE
    POSES_OVERRIDE = a.poses or None
    names = ("counts", "poses", "r_idx", "r_val", "p_idx", "p_val", "s_idx", "s_val")
    if a.cache and os.path.exists(a.cache):
        z = np.load(a.cache)
        wb = la.WindowBatch(a.batch, *[int(v) for v in z["caps"]])
        for nm in names: getattr(wb, nm)[:] = z[nm]
        graphs, anchors, T = [], z["anchors"], int(z["T"])
        a.cpu_n = 0
    else:
        if a.tile and a.tile < a.batch:
            small, graphs, anchors, T = build(a.tile, a.shape)
            wb = la.WindowBatch(a.batch, *small.caps)
            for nm in names:
                src = getattr(small, nm); getattr(wb, nm)[:] = np.resize(src, (a.batch,) + src.shape[1:])
        else:
            wb, graphs, anchors, T = build(a.batch, a.shape)
        if a.cache:
            np.savez(a.cache, caps=np.array(wb.caps), anchors=anchors, T=T, **{nm: getattr(wb, nm) for nm in names})
    if a.bw == "auto":
        bw = 0
        for b in range(a.batch):
            nr, ns = int(wb.counts[b, 1]), int(wb.counts[b, 3])
            r = wb.r_idx[b, :nr]; pp = r[r[:, 1] >= 0]
            if len(pp): bw = max(bw, int(np.abs(pp[:, 0] - pp[:, 1]).max()))
            if ns: bw = max(bw, int(np.abs(wb.s_idx[b, :ns, 0] - wb.s_idx[b, :ns, 1]).max()))
    else:
        bw = -1 if a.bw == "dense" else int(a.bw)
    solver = la.WindowSolver(anchors, a.batch, *wb.caps, maximum_iteration=10, bw_max=bw, jacobian=a.jacobian, natural_order=a.natural,
                             **({} if a.chain_threshold < 0 else {"chain_threshold": a.chain_threshold}))
    poses0 = wb.poses.copy()
    ms = []
    for r in range(a.reps + 1):
        wb.poses[:] = poses0
        t0 = time.perf_counter(); solver.solve(wb); wall = time.perf_counter() - t0
        if r: ms.append((solver.last_kernel_ms(), wall * 1e3))
    k_ms = float(np.median([m[0] for m in ms])); w_ms = float(np.median([m[1] for m in ms]))
    one = la.WindowBatch(1, *wb.caps)
    for name in ("counts", "poses", "r_idx", "r_val", "p_idx", "p_val", "s_idx", "s_val"):
        getattr(one, name)[:] = getattr(wb, name)[:1] if name != "poses" else poses0[:1]
    s1 = la.WindowSolver(anchors, 1, *wb.caps, maximum_iteration=10, bw_max=bw, jacobian=a.jacobian, natural_order=a.natural)
    lat = [float("nan")] if a.no_latency else []
    for r in range(0 if a.no_latency else 20):
        one.poses[:] = poses0[:1]
        t0 = time.perf_counter(); s1.solve(one); lat.append((time.perf_counter() - t0) * 1e3)
    if a.cpu_n > 0:
        cpu_s, cpu_t = oracle_time(graphs, anchors, T, min(a.cpu_n, a.batch))
        diff = float(np.abs(wb.poses[: len(cpu_t), :, 9:] - cpu_t).max())
    else:
        cpu_s, cpu_t, diff = float("nan"), [0], None
    # roofline: algorithmic bytes per window as SURVEY §8(d) counts them for cfg5 (window state read + written, 56 B per pose each
    # way, + one new measurement 232 B + chi2 8 B: 7 408 B at T = 64), priced against 8 TB/s; and, when PMC counts are given, the
    # issued f64 lane-operations against the fp64 vector peak.  The kernel is latency / VALU-issue bound, not bandwidth bound.
    algo_bytes = 2.0 * T * 56 + 240
    ach = algo_bytes * a.batch / (k_ms * 1e-3) / 1e9
    roof = {"bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
            "algorithmic_bytes_per_window": algo_bytes, "kernel": solver.last_kernel_kind(), "kernel_ms_avg": k_ms}
    if a.pmc_json and os.path.exists(a.pmc_json):
        pj = json.load(open(a.pmc_json))
        lane_flop = 64.0 * (pj.get("SQ_INSTS_VALU_ADD_F64", 0) + pj.get("SQ_INSTS_VALU_MUL_F64", 0) + 2 * pj.get("SQ_INSTS_VALU_FMA_F64", 0)
                            + pj.get("SQ_INSTS_VALU_TRANS_F64", 0))
        tf = lane_flop / (k_ms * 1e-3) / 1e12
        roof["valu_f64"] = {"issued_tflops": tf, "peak_tflops": 78.6, "frac": tf / 78.6,
                            "note": "f64 VALU wave-instructions x 64 lanes (PMC; counts idle lanes too: an upper bound on useful flops)"}
        if pj.get("FETCH_SIZE") is not None and pj.get("WRITE_SIZE") is not None:
            roof["traffic"] = (2.0 * pj["FETCH_SIZE"] + pj["WRITE_SIZE"]) * 1024.0   # KB -> B
            roof["traffic_rate_gbs"] = roof["traffic"] / (k_ms * 1e-3) / 1e9
            roof["traffic_note"] = ("HBM-side bytes per launch from PMC: FETCH_SIZE x 2 + WRITE_SIZE, the factor 2 measured for coalesced 8-B and 16-B per "
                                    "lane streams and for scattered 8-B reads alike (profiles/r03_fetch_calibration.json: FETCH_SIZE x 2 = 128-B lines fetched)")
    print(json.dumps({
        "roofline": roof,
        "shape": a.shape, "poses_per_window": T, "unknowns": 6 * T, "batch": a.batch, "bw_max": bw, "lds_bytes_per_instance": solver.lds_bytes,
        "gpu_kernel_ms_per_batch": k_ms, "gpu_windows_per_s_kernel": a.batch / (k_ms * 1e-3),
        "gpu_windows_per_s_incl_pcie": a.batch / (w_ms * 1e-3),
        "gpu_single_window_latency_ms_incl_pcie": float(np.median(lat)), "gpu_single_window_kernel_ms": s1.last_kernel_ms(),
        "cpu_oracle_windows_per_s_1core": len(cpu_t) / cpu_s, "cpu_ms_per_window": cpu_s / len(cpu_t) * 1e3,
        "mean_lm_trials": float(wb.result[:, 4].mean()), "jacobian": a.jacobian, "natural_order": bool(a.natural),
        "elimination_levels": float(wb.result[0, 7] // 65536), "factor_blocks": float(int(wb.result[0, 7]) % 65536),
        "root_supernode_poses": float(round((wb.result[0, 7] % 1.0) * 16)),
        "max_abs_diff_vs_oracle_numeric_m": diff}))


if __name__ == "__main__":
    main()
