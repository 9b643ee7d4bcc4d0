"""ORACLE — TEST INFRASTRUCTURE ONLY (see g2o_graph_oracle.h; PARITY UNPINNED).

All-cores driver for the CPU baseline leg of bench.py: a static split of independent tags over
worker PROCESSES, each running the single-threaded oracle on its own contiguous slice.  Processes, not threads: the
oracle builds and frees one small graph per update and glibc's allocator serialises that across threads (measured here:
8 threads = 2.3x one thread, 8 processes = 5.8x).  Workers are spawned (never forked: the parent may hold a HIP
context) and import numpy + the oracle only.
"""
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ready(_):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import oracle as O
    O.lib()
    return os.getpid()


def _solve(part):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import oracle as O
    anchors, d, e, init, kw = part
    t0 = time.perf_counter()
    pos = O.snapshot_batch(anchors, d, e, init, **kw)[0]
    return pos[-1], time.perf_counter() - t0


def snapshot_batch_all_cores(anchors, dist, err, init, workers, **kw):
    """dist/err [K][M][B], init [3][B] -> (final positions [3][B], wall seconds of the solve, workers used).
    The pool is started and warmed (library loaded in every worker) before the clock starts."""
    import numpy as np
    B = dist.shape[2]
    workers = max(1, min(workers, B))
    bounds = np.linspace(0, B, workers + 1).astype(int)
    parts = [(anchors, np.ascontiguousarray(dist[:, :, a:b]), np.ascontiguousarray(err[:, :, a:b]),
              np.ascontiguousarray(init[:, a:b]), kw) for a, b in zip(bounds[:-1], bounds[1:])]
    with mp.get_context("spawn").Pool(workers) as pool:
        pool.map(_ready, range(workers), chunksize=1)
        t0 = time.perf_counter()
        res = pool.map(_solve, parts, chunksize=1)
        wall = time.perf_counter() - t0
    return np.concatenate([r[0] for r in res], axis=1), wall, workers
