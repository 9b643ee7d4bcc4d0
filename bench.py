#!/usr/bin/env python3
"""Headline benchmark: localization updates/s, synthetic 8-anchor UWB, batch = 65 536 tags per GPU (BASELINE cfg 2).

A "step" is one launch of the hot path over one batch of synthetic input: E epochs x B tags, i.e. E*B updates
(one update = gate + M Cauchy range factors + the reference's fixed `maximum_iteration` = 10 LM iterations +
chi2).  All inputs and outputs are resident in HBM before the timed region; every step consumes
fresh epochs of one long seeded random-walk stream (no buffer is re-read between steps).

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver through torch.distributed.run (one rank per GPU, RCCL only for the barrier and the
max-over-ranks of the time: the path has no data-path collective; the batch is sharded by tag, "weak" scaling).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_PER_UPDATE = 120.0  # 8x4 dist + 8x4 err + 3x8 prior read, 3x8 pos + 8 chi2 written
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBS = 6290.0          # same guide: measured copy ceiling (SURVEY.md §8(d) asks for the fraction of both)


def cpu_baseline(anchors, dist_tiles, err_tiles, init, n_tags, n_epochs, gpu_pos, M):
    """The oracle (CPU restatement of the reference's g2o path: 6-DoF vertices, numeric Jacobians, LM) timed on one
    host core over a bounded sample of the same workload: the first n_tags tags x the first n_epochs epochs."""
    import numpy as np
    from localization_amd.snapshot import unpack_ranges
    from oracle import oracle as O
    d = unpack_ranges(dist_tiles[:n_epochs, :, :n_tags, :].cpu().numpy(), M)
    e = unpack_ranges(err_tiles[:n_epochs, :, :n_tags, :].cpu().numpy(), M)
    O.lib()
    t0 = time.perf_counter()
    rp, rc, rt, _ = O.snapshot_batch(anchors, d, e, init[:, :n_tags], iterations=10, gate=1.0,
                                     jac_mode=O.JAC_NUMERIC_G2O, gate_from_epoch=1)
    dt = time.perf_counter() - t0
    out = {"value": n_tags * n_epochs / dt, "unit": "updates/s", "cores": 1, "kind": "port",
           "sample": f"first {n_tags} tags x first {n_epochs} epochs of the benchmark stream "
                     f"({n_tags * n_epochs} updates, {dt:.1f} s): oracle/ g2o restatement, 6-DoF vertices, "
                     "numeric central-difference Jacobians, 10 LM iterations, 1 thread"}
    if gpu_pos is not None:
        g = gpu_pos[:n_epochs, :, :n_tags].cpu().numpy()
        out["max_abs_diff_vs_gpu_m"] = float(np.abs(g - rp).max())
    return out


def host_core_share(cap=16):
    """Cores this process may really use: the affinity mask, cut to the cgroup CPU quota when one is set, and never more
    than `cap` — a GPU box shows the whole host's cores in its affinity mask but gives one GPU's job a 16-core share."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(float(quota) / period)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, cap))


def cpu_baseline_all_cores(anchors, dist_tiles, err_tiles, init, tags_per_core, n_epochs, M):
    """the same oracle with a static split of the tags over every host core this process may use
    (one single-threaded worker process per core, oracle/parallel.py) — the fair throughput baseline.  The reference
    itself is single-threaded (localization_node.cpp:98): `cpu_baseline` (1 core) stays the headline CPU figure."""
    from localization_amd.snapshot import unpack_ranges
    from oracle import oracle as O
    from oracle.parallel import snapshot_batch_all_cores
    cores = host_core_share()
    n_tags = min(tags_per_core * cores, dist_tiles.shape[2])
    d = unpack_ranges(dist_tiles[:n_epochs, :, :n_tags, :].cpu().numpy(), M)
    e = unpack_ranges(err_tiles[:n_epochs, :, :n_tags, :].cpu().numpy(), M)
    _, dt, used = snapshot_batch_all_cores(anchors, d, e, init[:, :n_tags], cores, iterations=10, gate=1.0,
                                           jac_mode=O.JAC_NUMERIC_G2O, gate_from_epoch=1)
    return {"value": n_tags * n_epochs / dt, "unit": "updates/s", "cores": used, "kind": "port",
            "sample": f"first {n_tags} tags x first {n_epochs} epochs ({n_tags * n_epochs} updates, {dt:.1f} s), static split "
                      f"over {used} single-threaded worker processes, same oracle and settings as cpu_baseline"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=65536, help="tags per GPU")
    ap.add_argument("--epochs", type=int, default=128, help="epochs per launch (step)")
    ap.add_argument("--jacobian", default="analytic", choices=["analytic", "numeric"])
    ap.add_argument("--lpi", type=int, default=0, help="lanes per tag (0 = library default)")
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--datagen", default="torch", choices=["torch", "numpy"],
                    help="numpy: generate the stream on the host and copy it (PMC profiling runs: no torch kernels)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the driver's multi-GPU runs); gloo only to rehearse N > 1 on a one-GPU box")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-tags", type=int, default=2048)
    ap.add_argument("--cpu-epochs", type=int, default=96)
    args = ap.parse_args()

    import numpy as np
    import torch
    import localization_amd as la
    from localization_amd.synthetic import ANCHORS_8, make_snapshot_stream_torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    n_dev = torch.cuda.device_count()
    if args.dist_backend == "gloo":
        local_rank = local_rank % max(n_dev, 1)  # rehearsal: ranks may share a device
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    n_gpus = world
    if args.gpus != n_gpus and rank == 0:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    B, E, M = args.batch, args.epochs, 8
    total_steps = args.warmup + args.steps
    # one long stream: every step reads fresh epochs (inputs (W+K)*E*B*64 B resident in HBM)
    if args.datagen == "torch":
        stream = make_snapshot_stream_torch(B, E * total_steps, seed=args.seed + 1000 * rank, device=dev)
    else:
        from localization_amd.synthetic import make_snapshot_stream
        hs = make_snapshot_stream(B, E * total_steps, seed=args.seed + 1000 * rank)
        stream = {"dist_tiles": torch.from_numpy(la.pack_ranges(hs["dist"])).to(dev),
                  "err_tiles": torch.from_numpy(la.pack_ranges(hs["err"])).to(dev),
                  "init": hs["init"], "truth_last": torch.from_numpy(hs["truth"][-1]).to(dev)}
        del hs
    dist_t, err_t = stream["dist_tiles"], stream["err_tiles"]
    solver = la.SnapshotSolver(ANCHORS_8, B, maximum_iteration=10, distance_outlier=1.0, jacobian=args.jacobian,
                               lanes_per_instance=args.lpi, block_threads=args.block, device=local_rank)
    solver.set_positions(stream["init"])
    out_pos = torch.empty((total_steps * E, 3, B), dtype=torch.float64, device=dev)
    out_chi2 = torch.empty((total_steps * E, B), dtype=torch.float64, device=dev)
    out_trials = torch.empty((total_steps * E, B), dtype=torch.uint8, device=dev)

    def step(i):
        sl = slice(i * E, (i + 1) * E)
        solver.solve_device(dist_t[sl], err_t[sl], out_pos[sl], out_chi2[sl], out_trials[sl])

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    solver.timing_begin(args.steps)
    t0 = time.perf_counter()
    for i in range(args.warmup, total_steps):
        step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    n_launch, kern_ms_total, kern_ms_avg = solver.timing_end()
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    updates_per_launch = float(B) * E
    total_updates = updates_per_launch * args.steps * n_gpus
    value = total_updates / elapsed
    achieved_gbs = ALGO_BYTES_PER_UPDATE * updates_per_launch / (kern_ms_avg * 1e-3) / 1e9

    if rank == 0:
        trials_mean = float(out_trials[args.warmup * E:].cpu().numpy().mean())
        err_last = torch.from_numpy(np.sqrt(((out_pos[-1].cpu().numpy() - stream["truth_last"].cpu().numpy()) ** 2).sum(axis=0)))
        # HBM bytes per launch from the committed PMC passes (rocprofv3 cannot run inside this process); they apply
        # to the default launch shape only
        traffic, traffic_src, f64_flop, issue_slots, issue_ceiling = None, None, None, None, None
        prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(prof) and (B, E) == (65536, 128) and args.jacobian == "analytic":
            try:
                with open(prof) as f:
                    pj = json.load(f)
                traffic, traffic_src, f64_flop = pj["hbm_bytes_per_launch"], pj["source"], pj.get("f64_flop_per_launch")
                issue_slots, issue_ceiling = pj.get("issue_lane_slots_per_launch"), pj.get("measured_issue_ceiling_lane_slots_per_s")
            except Exception:
                traffic = None
        res = {
            "metric": "localization updates/sec (8-anchor UWB, batch=65k)",
            "value": value, "unit": "updates/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE cfg2: synthetic 8-anchor UWB, 3-DoF position, Cauchy range factors, "
                                   "g2o-style LM with the reference's fixed 10 iterations, outlier gate 1 m",
                       "batch_per_gpu": B, "epochs_per_step": E, "updates_per_step_per_gpu": int(updates_per_launch),
                       "anchors": M, "lm_iterations": 10, "jacobian": args.jacobian,
                       "lanes_per_tag": solver.lanes_per_instance, "sharding": f"tags split over {n_gpus} GPU(s), no collective"},
            "lm_iterations_per_s": value * 10,
            "mean_lm_trials_per_update": trials_mean,
            "median_err_vs_truth_m": float(err_last.median().item()),
            "frac_err_gt_0p5m": float((err_last > 0.5).double().mean().item()),
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS, "frac_of_measured_copy_ceiling": achieved_gbs / HBM_COPY_GBS,
                         "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_UPDATE * updates_per_launch,
                         "kernel": "snapshot_lm_kernel", "kernel_ms_avg": kern_ms_avg, "launches_timed": n_launch,
                         "algorithmic_bytes_per_update": ALGO_BYTES_PER_UPDATE,
                         "note": "fp64, 10 LM iterations: VALU-bound, see DESIGN.md"},
        }
        if f64_flop:
            tf = f64_flop / (kern_ms_avg * 1e-3) / 1e12
            res["valu_f64"] = {"achieved_tflops": tf, "peak_tflops": 78.6, "frac": tf / 78.6,
                               "note": "f64 add+mul+2*fma+trans lane-ops per launch (PMC, profiles/) / live kernel time; "
                                       "the kernel is VALU-issue bound at one wave per SIMD, not HBM bound"}
        if issue_slots and issue_ceiling:
            rate = issue_slots / (kern_ms_avg * 1e-3)
            res["valu_issue"] = {"achieved_lane_slots_per_s": rate, "measured_ceiling": issue_ceiling, "frac": rate / issue_ceiling,
                                 "note": "(VALU + SALU wave-instructions per launch) x 64 lanes (PMC) / live kernel time, against the "
                                         "measured one-wave-per-SIMD issue ceiling (tools/fp64_probe.hip): what actually bounds this kernel"}
        if not args.no_cpu_baseline and n_gpus == 1:
            res["cpu_baseline"] = cpu_baseline(ANCHORS_8, dist_t, err_t, stream["init"], min(args.cpu_tags, B),
                                               min(args.cpu_epochs, E * total_steps), out_pos, M)
            try:  # a secondary figure: a host that refuses worker processes must not cost the bench its JSON line
                res["cpu_baseline_all_cores"] = cpu_baseline_all_cores(ANCHORS_8, dist_t, err_t, stream["init"], args.cpu_tags // 2,
                                                                       min(args.cpu_epochs, E * total_steps), M)
            except Exception as exc:  # noqa: BLE001
                res["cpu_baseline_all_cores"] = {"value": None, "error": f"{type(exc).__name__}: {exc}"}
        print(json.dumps(res))
    solver.close()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
