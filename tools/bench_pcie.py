#!/usr/bin/env python3
"""PCIe-inclusive rate of the snapshot path (host buffers in, host buffers out).  Never the headline `value` (bench.py
measures HBM-resident inputs); quoted in DESIGN.md §5.  Three ways through the ABI:
  tiles_sync      loc_snapshot_solve_host: pre-packed tiles, copy-in -> solve -> copy-out one after the other
  kmb_pipe        loc_snapshot_solve_host_kmb: natural [K][M][B] layout, packed on the GPU, three-stream pipeline; pageable
  kmb_pipe_pinned same with page-locked buffers from loc_host_alloc (what lets the copies overlap)"""
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

B = 65536
K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
s = make_snapshot_stream(B, K, seed=0)
solver = la.SnapshotSolver(ANCHORS_8, B, maximum_iteration=10, distance_outlier=1.0, jacobian="analytic")
fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
res = {"batch": B, "epochs": K, "bytes_over_pcie_per_update": 64 + 33}


def timed(fn, reps=5):
    ts = []
    for r in range(reps + 1):
        solver.set_positions(s["init"])
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts[1:]))


d = la.pack_ranges(s["dist"]); e = la.pack_ranges(s["err"])
out = (np.empty((K, 3, B)), np.empty((K, B)), np.empty((K, B), dtype=np.uint8))
t = timed(lambda: la._lib.check(solver.L.loc_snapshot_solve_host(solver.h, K, d.ctypes.data_as(fp), e.ctypes.data_as(fp), out[0].ctypes.data_as(dp),
                                                                out[1].ctypes.data_as(dp), out[2].ctypes.data_as(C.POINTER(C.c_uint8)))))
ref = [o.copy() for o in out]
res["tiles_sync"] = {"ms_per_call": t * 1e3, "updates_per_s": B * K / t, "effective_GBps": 97 * B * K / t / 1e9}
t = timed(lambda: solver.solve_stream(s["dist"], s["err"], out))
assert all(np.array_equal(a, b) for a, b in zip(ref, out))
res["kmb_pipe"] = {"ms_per_call": t * 1e3, "updates_per_s": B * K / t, "effective_GBps": 97 * B * K / t / 1e9}
pd = solver.pinned((K, 8, B), np.float32); pe = solver.pinned((K, 8, B), np.float32)
pd[:] = s["dist"]; pe[:] = s["err"]
pout = (solver.pinned((K, 3, B), np.float64), solver.pinned((K, B), np.float64), solver.pinned((K, B), np.uint8))
t = timed(lambda: solver.solve_stream(pd, pe, pout))
assert all(np.array_equal(a, b) for a, b in zip(ref, pout))
res["kmb_pipe_pinned"] = {"ms_per_call": t * 1e3, "updates_per_s": B * K / t, "effective_GBps": 97 * B * K / t / 1e9}
print(json.dumps(res))
