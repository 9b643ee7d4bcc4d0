"""N > 1 path on CPU: world_size-2 gloo processes shard a batch by tag, solve their slices independently (here with
the oracle standing in for the device — the product has no CPU path) and all-gather the result slabs; the gathered
batch must equal the single-process result bit for bit (tags are independent: no data-path collective)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from localization_amd.sharding import all_gather_results, all_reduce_scalars, barrier_and_max, shard_array, shard_bounds, shard_window_batch


def test_shard_bounds_cover_batch_exactly():
    for total in [1, 7, 64, 65536, 65537, 100003]:
        for world in [1, 2, 3, 4, 8]:
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b
            assert max(b - a for a, b in spans) == -(-total // world)
    assert shard_bounds(65536, 3, 8) == (24576, 32768)          # 8 192 tags per GPU at G = 8


def test_shard_descriptor_of_the_c_abi():
    """loc_shard_bounds / loc_shard_plan (include/localization_amd.h): the rule every caller shares — SURVEY §8(e)'s per-GPU slices."""
    import localization_amd as la
    from localization_amd.sharding import shard_plan
    plan = shard_plan(16384, 8, 8)                                 # cfg5: 2 048 windows per GPU at G = 8
    assert [(r, d, hi - lo) for r, d, lo, hi in plan] == [(r, r, 2048) for r in range(8)]
    plan = shard_plan(1024, 8, 8)                                  # cfg4: 128 hypotheses per GPU
    assert plan[3] == (3, 3, 384, 512)
    plan = shard_plan(10, 4, 2)                                    # ragged: 3 3 3 1, two GPUs per node
    assert [(d, lo, hi) for _, d, lo, hi in plan] == [(0, 0, 3), (1, 3, 6), (0, 6, 9), (1, 9, 10)]
    assert shard_plan(2, 4, 4)[3][2:] == (2, 2)                    # trailing ranks may be empty
    for bad in [(10, 4, 4), (10, -1, 4), (-1, 0, 4), (10, 0, 0)]:
        with pytest.raises(la.LocalizationAmdError):
            shard_bounds(*bad)
    # the pure-arithmetic statement of the same rule
    for total in [0, 1, 7, 65536, 100003]:
        for world in [1, 2, 3, 8]:
            per = -(-total // world)
            for r in range(world):
                assert shard_bounds(total, r, world) == (min(r * per, total), min(min(r * per, total) + per, total))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, B, K, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from localization_amd.synthetic import make_snapshot_stream
    from oracle import oracle as O
    s = make_snapshot_stream(B, K, seed=123)
    d = shard_array(s["dist"], rank, world); e = shard_array(s["err"], rank, world); init = shard_array(s["init"], rank, world)
    pos, chi2, trials, _ = O.snapshot_batch(s["anchors"], d, e, np.ascontiguousarray(init), iterations=10, gate=1.0,
                                            jac_mode=O.JAC_ANALYTIC)
    full_pos = all_gather_results(torch.from_numpy(pos), B, axis=-1)
    full_chi = all_gather_results(torch.from_numpy(chi2), B, axis=-1)
    tot = all_reduce_scalars([chi2.sum(), float(trials.sum())])
    t = barrier_and_max(0.1 * (rank + 1))
    if rank == 0:
        np.savez(out_path, pos=full_pos.numpy(), chi2=full_chi.numpy(), tot=np.array(tot), t=np.array(t))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [64, 37])
def test_two_rank_shard_equals_single_process(tmp_path, B):
    from localization_amd.synthetic import make_snapshot_stream
    from oracle import oracle as O
    O.build()
    K, world = 3, 2
    out = str(tmp_path / "gathered.npz")
    mp.start_processes(_worker, args=(world, _free_port(), B, K, out), nprocs=world, join=True, start_method="spawn")
    got = np.load(out)
    s = make_snapshot_stream(B, K, seed=123)
    pos, chi2, trials, _ = O.snapshot_batch(s["anchors"], s["dist"], s["err"], s["init"], iterations=10, gate=1.0,
                                            jac_mode=O.JAC_ANALYTIC)
    assert np.array_equal(got["pos"], pos) and np.array_equal(got["chi2"], chi2)
    assert got["tot"][0] == pytest.approx(chi2.sum(), rel=1e-12) and got["tot"][1] == trials.sum()
    assert float(got["t"]) == pytest.approx(0.2)


def _window_worker(rank, world, port, B, out_path):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench_window as bw
    from _oracle_window import oracle_solve_instance
    wb, _, anchors, T = bw.build_pose64(B, np.random.default_rng(5), T=12, n_graphs=0)
    part, lo, hi = shard_window_batch(wb, rank, world)
    poses = np.zeros((hi - lo, T, 12)); chi2 = np.zeros(hi - lo)
    for i in range(hi - lo):
        poses[i], chi2[i], _ = oracle_solve_instance(part, i, anchors)
    full_pose = all_gather_results(torch.from_numpy(poses), B, axis=0)
    full_chi = all_gather_results(torch.from_numpy(chi2), B, axis=0)
    if rank == 0:
        np.savez(out_path, poses=full_pose.numpy(), chi2=full_chi.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [6, 5])
def test_two_rank_window_shard_equals_single_process(tmp_path, B):
    """cfg5 / cfg4 style sharding: a WindowBatch split by instance over two gloo ranks, each rank solving its
    windows independently (the oracle standing in for the device), poses and chi2 all-gathered: bit-identical to one process."""
    from oracle import oracle as O
    O.build()
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))
    import bench_window as bw
    from _oracle_window import oracle_solve_instance
    out = str(tmp_path / "gathered_windows.npz")
    mp.start_processes(_window_worker, args=(2, _free_port(), B, out), nprocs=2, join=True, start_method="spawn")
    got = np.load(out)
    wb, _, anchors, T = bw.build_pose64(B, np.random.default_rng(5), T=12, n_graphs=0)
    for i in range(B):
        p, c, _ = oracle_solve_instance(wb, i, anchors)
        assert np.array_equal(got["poses"][i], p) and got["chi2"][i] == c
    part, lo, hi = shard_window_batch(wb, 1, 2)
    assert (lo, hi) == (-(-B // 2), B) and np.array_equal(part.s_val[: hi - lo], wb.s_val[lo:hi])


def test_bench_refuses_to_run_a_smaller_job_than_asked():
    """`python bench.py --gpus N` without a launcher starts the N ranks itself — and exits non-zero, before touching any GPU, when fewer
    than N devices are visible (here: none), instead of silently benchmarking what is there."""
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--legs", "none"],
                         capture_output=True, text=True, timeout=120, env=env)
    assert run.returncode != 0 and "refusing to run a smaller job" in run.stderr and run.stdout.strip() == ""


class _FakeDist:
    """what bench.py's Dist offers to reporting_collectives, on host tensors (no device in this container)"""

    def __init__(self, rank, world):
        self.torch, self.rank, self.world, self.backend, self.dev = torch, rank, world, "gloo", torch.device("cpu")

    def barrier(self):
        dist.barrier()


def _collectives_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import argparse
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    from localization_amd.synthetic import make_snapshot_stream
    from localization_amd.snapshot import pack_ranges
    B, E, steps = 48, 4, 3
    s = make_snapshot_stream(B, E * steps, seed=100 + rank)
    g = torch.Generator().manual_seed(rank)
    out_pos = torch.from_numpy(np.ascontiguousarray(s["truth"])) + 0.01 * torch.randn(E * steps, 3, B, generator=g, dtype=torch.float64)
    out_chi2 = torch.rand(E * steps, B, generator=g, dtype=torch.float64)
    if rank == 1:
        out_chi2[-1, 3] = float("inf"); out_pos[-1, 0, 5] = float("nan")
    out_trials = torch.full((E * steps, B), 10 + rank, dtype=torch.uint8)
    tiles = torch.from_numpy(pack_ranges(s["dist"]))
    args = argparse.Namespace(warmup=1)
    res = bench.reporting_collectives(_FakeDist(rank, world), args, out_pos, out_chi2, out_trials, tiles, E, steps, B)
    fin = torch.isfinite(out_chi2[E:])
    mine = [float(out_chi2[E:][fin].sum()), float(out_trials[E:].double().sum()), 1.0 if rank == 1 else 0.0]
    allm = [torch.zeros(3, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(allm, torch.tensor(mine, dtype=torch.float64))
    if rank == 0:
        import json
        json.dump({"res": res, "sums": torch.stack(allm).sum(0).tolist()}, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_reporting_collectives_two_ranks_gloo(tmp_path):
    """SURVEY §8(e)(1)-(2) as bench.py runs them after the timed region for N > 1: the all-reduce(SUM) of the four reporting scalars and
    the all-gather of one result slab per rank with its checksum — rehearsed with two gloo ranks on host tensors (on a GPU node the same
    code runs over RCCL on device tensors; unmeasured on hardware)."""
    import json
    out = str(tmp_path / "coll.json")
    mp.start_processes(_collectives_worker, args=(2, _free_port(), out), nprocs=2, join=True, start_method="spawn")
    j = json.load(open(out))
    res, sums = j["res"], j["sums"]
    assert res["ranks"] == 2 and res["backend"].startswith("gloo")
    assert res["scalars"] == ["sum_chi2", "sum_lm_trials", "n_nonfinite_estimates", "n_gated_ranges_last_epoch"]
    v = res["all_reduce"]["values"]
    assert v[0] == pytest.approx(sums[0], rel=1e-12) and v[1] == sums[1] == 48 * 8 * (10 + 11) and v[2] == sums[2] == 1.0
    assert v[3] >= 0 and res["all_reduce"]["bytes"] == 32
    assert res["all_gather"]["checksum_ok"] is True and res["all_gather"]["bytes"] == 2 * 4 * 48 * 8 and res["all_gather"]["slab_shape_per_rank"] == [4, 48]
    assert res["bytes"] == 32 + 3072 and res["ms"] > 0
