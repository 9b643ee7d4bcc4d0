#!/usr/bin/env python3
"""PCIe-inclusive rate of the snapshot path (host buffers in, host buffers out through loc_snapshot_solve_host).
Never the headline `value` (bench.py measures HBM-resident inputs); quoted in DESIGN.md §5."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import localization_amd as la
from localization_amd.synthetic import ANCHORS_8, make_snapshot_stream

B, K = 65536, 16
s = make_snapshot_stream(B, K, seed=0)
solver = la.SnapshotSolver(ANCHORS_8, B, maximum_iteration=10, distance_outlier=1.0)
solver.set_positions(s["init"])
d = la.pack_ranges(s["dist"]); e = la.pack_ranges(s["err"])
out_pos = np.empty((K, 3, B)); out_chi2 = np.empty((K, B)); trials = np.empty((K, B), dtype=np.uint8)
fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
ts = []
for r in range(6):
    t0 = time.perf_counter()
    rc = solver.L.loc_snapshot_solve_host(solver.h, K, d.ctypes.data_as(fp), e.ctypes.data_as(fp), out_pos.ctypes.data_as(dp),
                                          out_chi2.ctypes.data_as(dp), trials.ctypes.data_as(C.POINTER(C.c_uint8)))
    assert rc == 0
    ts.append(time.perf_counter() - t0)
t = float(np.median(ts[1:]))
print(json.dumps({"batch": B, "epochs": K, "ms_per_call": t * 1e3, "updates_per_s_pcie_inclusive": B * K / t,
                  "bytes_over_pcie_per_update": 64 + 33, "effective_GBps": (64 + 33) * B * K / t / 1e9,
                  "note": "pageable host memory, synchronous staging (hipMemcpyAsync + stream sync)"}))
