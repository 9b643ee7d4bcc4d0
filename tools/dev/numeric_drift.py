#!/usr/bin/env python3
"""Diagnostic: per-solve difference between the node front-end on the GPU and the oracle front-end in both Jacobian modes on the
example bag (first N ranges) — is a numeric-vs-numeric difference drift or a flip of an LM decision?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import localization_amd as la
from oracle import oracle as O
import test_gpu_node_parity as T
bag = np.load(os.path.join(ROOT, "tests", "golden", "bag_example.npz"))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
cfg = dict(trajectory_length=10, maximum_velocity=5.0, distance_outlier=1.0, maximum_iteration=10, minimum_optimize_error=2000.0, publish_range=True)
ev = T._events(bag, False, N)
res = {}
ids = list(bag["anchor_ids"]) + [200]; pos = np.concatenate([bag["anchor_pos"], [[0.0, 0.0, 1.0]]])
for name, obj in (("gpu_num", la.LocalizationNode(ids, pos, jacobian="numeric", **cfg)), ("gpu_ana", la.LocalizationNode(ids, pos, **cfg)),
                  ("ora_num", O.LocalizationOracle(ids, pos, jac_mode=O.JAC_NUMERIC_G2O, **cfg)), ("ora_ana", O.LocalizationOracle(ids, pos, jac_mode=O.JAC_ANALYTIC, **cfg))):
    rt, trials = [], []
    for _, kind, i in ev:
        o = obj.add_range(200, int(bag["uwb_responder"][i]), float(bag["uwb_stamp"][i]), bag["uwb_distance"][i], bag["uwb_distance_err"][i], int(bag["uwb_antenna"][i]), "uwb")
        if o["solved"]: rt.append(o["realtime"]); trials.append(o["lm_trials"])
    res[name] = (np.array(rt), np.array(trials))
def d(a, b): return np.abs(res[a][0][:, 1:4] - res[b][0][:, 1:4]).max(axis=1)
print("solve  gpu_num-ora_num  gpu_ana-ora_ana  ora_num-ora_ana  trials(gpu_num ora_num gpu_ana ora_ana)")
for k in range(len(res["gpu_num"][0])):
    print(f"{k:4d}  {d('gpu_num','ora_num')[k]:.2e}  {d('gpu_ana','ora_ana')[k]:.2e}  {d('ora_num','ora_ana')[k]:.2e}   {res['gpu_num'][1][k]} {res['ora_num'][1][k]} {res['gpu_ana'][1][k]} {res['ora_ana'][1][k]}")
