#!/usr/bin/env python3
"""Secondary measurement: BASELINE config 3 (8 anchors + IMU prior, 6-DoF, batch 65 536) on the fusion kernel.
Not the headline metric (bench.py is); prints one JSON line with updates/s, the HBM-roofline fraction at SURVEY §8(d)'s
248 B/update, and the oracle's time on one host core over a bounded sample."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--epochs", type=int, default=64)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--cpu-tags", type=int, default=512)
    ap.add_argument("--cpu-epochs", type=int, default=32)
    ap.add_argument("--jacobian", default="analytic", choices=["analytic", "numeric"], help="numeric = the reference's configuration")
    a = ap.parse_args()
    import torch
    import localization_amd as la
    from localization_amd.synthetic import make_fusion_stream
    from oracle import oracle as O
    B, E = a.batch, a.epochs
    dev = torch.device("cuda", 0)
    s = make_fusion_stream(B, E, seed=0)
    dist = torch.from_numpy(la.pack_ranges(s["dist"])).to(dev)
    err = torch.from_numpy(la.pack_ranges(s["err"])).to(dev)
    imu = torch.from_numpy(s["imu"]).to(dev)
    f = la.FusionSolver(s["anchors"], B, antenna_offset=s["offset"], maximum_iteration=10, distance_outlier=3.0, jacobian=a.jacobian)
    out_pose = torch.empty((E, 7, B), dtype=torch.float64, device=dev)
    out_chi2 = torch.empty((E, B), dtype=torch.float64, device=dev)
    out_trials = torch.empty((E, B), dtype=torch.uint8, device=dev)
    ms = []
    for r in range(a.steps + 1):
        f.set_poses(s["init"])                      # replay the same 64 epochs from the same start
        torch.cuda.synchronize()
        f.solve_device(dist, err, imu, out_pose, out_chi2, out_trials)
        torch.cuda.synchronize()
        if r:
            ms.append(f.last_kernel_ms())
    k_ms = float(np.median(ms))
    upd = float(B) * E
    nt, ne = min(a.cpu_tags, B), min(a.cpu_epochs, E)
    t0 = time.perf_counter()
    rp, rc, rt, _ = O.fusion_batch(s["anchors"], s["offset"], s["dist"][:ne, :, :nt], s["err"][:ne, :, :nt], s["imu"][:ne, :nt],
                                   s["init"][:, :nt], iterations=10, gate=3.0, jac_mode=O.JAC_NUMERIC_G2O)
    cpu_s = time.perf_counter() - t0
    g = out_pose[:ne, :, :nt].cpu().numpy()
    e = np.sqrt(((out_pose[-1, :3].cpu().numpy() - s["truth_t"][-1]) ** 2).sum(axis=0))
    print(json.dumps({
        "config": "BASELINE cfg3: 8 anchors + IMU rotation prior + antenna lever arm, 6-DoF, g2o-style LM, 10 iterations",
        "batch": B, "epochs_per_launch": E, "jacobian": a.jacobian, "kernel_ms": k_ms, "updates_per_s": upd / (k_ms * 1e-3),
        "roofline": {"bound": "hbm", "algorithmic_bytes_per_update": 248.0, "achieved_GBps": 248.0 * upd / (k_ms * 1e-3) / 1e9,
                     "peak_GBps": 8000.0, "frac": 248.0 * upd / (k_ms * 1e-3) / 1e9 / 8000.0},
        "mean_lm_trials": float(out_trials.cpu().numpy().mean()),   # (host-side: no torch kernels, so the script runs under rocprofv3 --pmc)
        "median_err_vs_truth_m": float(np.median(e)),
        "cpu_baseline": {"updates_per_s": nt * ne / cpu_s, "cores": 1, "kind": "port",
                         "sample": f"{nt} tags x {ne} epochs, oracle g2o restatement (numeric Jacobians)",
                         "max_abs_diff_vs_gpu": float(np.abs(g - rp).max())}}))
    f.close()


if __name__ == "__main__":
    main()
