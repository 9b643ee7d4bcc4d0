#!/usr/bin/env python3
"""Headline benchmark: localization updates/s, synthetic 8-anchor UWB, batch = 65 536 tags per GPU (BASELINE cfg 2).

A "step" is one launch of the hot path over one batch of synthetic input: E epochs x B tags, i.e. E*B updates
(one update = gate + M Cauchy range factors + the reference's fixed `maximum_iteration` = 10 LM iterations +
chi2).  All inputs and outputs are resident in HBM before the timed region; every step consumes
fresh epochs of one long seeded random-walk stream (no buffer is re-read between steps).

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver through torch.distributed.run (one rank per GPU, RCCL only for the barrier and the
max-over-ranks of the time: the path has no data-path collective; the batch is sharded by tag, "weak" scaling).

Besides the headline (the top-level keys of the ONE JSON line), the line carries driver-timed secondary legs under
"legs": cfg2 in the reference's numeric-Jacobian configuration, cfg3 (fusion kernel), cfg5 (64-pose windows) and cfg4
(anchor self-calibration) — each with its own roofline and, at N = 1, cpu_baseline.  For N > 1 the window legs shard the
SURVEY §8(e) totals (16 384 windows, 1 024 hypotheses) over the ranks ("strong"), cfg2/cfg3 keep 65 536 tags per GPU.
--legs none|all|comma list selects them (default all).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_PER_UPDATE = 120.0  # 8x4 dist + 8x4 err + 3x8 prior read, 3x8 pos + 8 chi2 written
ALGO_BYTES_CFG3 = 248.0
ALGO_BYTES_CFG5 = 7408.0       # SURVEY.md §8(d), cfg5 row (per window)
ALGO_BYTES_CFG4_PER_IT = 33248.0  # SURVEY.md §8(d), cfg4 row (per hypothesis per LM iteration)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBS = 6290.0          # same guide: measured copy ceiling (SURVEY.md §8(d) asks for the fraction of both)
F64_VECTOR_PEAK_TFLOPS = 78.6  # datasheet fp64 vector peak


def kernel_source_hash(names):
    """sha256 over the kernel sources a PMC summary was measured on: when they change, the committed counters are stale."""
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(ROOT, "localization_amd", "csrc", n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


SNAPSHOT_KERNEL_SOURCES = ("snapshot_kernel.hip", "snapshot_kernel.h", "device_math.h")
_WIN_COMMON = ("window_kernel.h", "device_math.h", "numeric_jacobian.h")
# per leg: the sources of the kernel that leg runs (one translation unit per kernel since round 4)
WINDOW_LEG_SOURCES = {"cfg1_windows": ("wave3_kernel.hip",) + _WIN_COMMON,
                      "cfg4": ("arrow3_kernel.hip",) + _WIN_COMMON,
                      "cfg5": ("tree_wave_kernel.hip", "se3_edge_device.h", "window_device.h") + _WIN_COMMON,
                      "twist_windows": ("wave6_kernel.hip", "se3_edge_device.h", "window_device.h") + _WIN_COMMON}


def window_traffic(leg, world):
    """PMC-measured HBM-side bytes per launch of a window leg's single-GPU batch (profiles/window_traffic.json,
    tools/make_window_traffic.py), or None when the kernel sources have changed since or the batch is sharded differently."""
    path = os.path.join(ROOT, "profiles", "window_traffic.json")
    if world != 1 or not os.path.exists(path):
        return None
    ent = json.load(open(path)).get("legs", {}).get(leg)
    if not ent or leg not in WINDOW_LEG_SOURCES:
        return None
    if ent.get("kernel_source_sha256") != kernel_source_hash(WINDOW_LEG_SOURCES[leg]):
        return {"stale": "profiles/window_traffic.json was measured on different sources of this leg's kernel (hash mismatch): traffic withheld"}
    return ent


def cpu_baseline(anchors, dist_tiles, err_tiles, init, n_tags, n_epochs, gpu_pos, M, gpu_jacobian="numeric"):
    """The oracle (CPU restatement of the reference's g2o path: 6-DoF vertices, numeric Jacobians, LM) timed on one
    host core over a bounded sample of the same workload: the first n_tags tags x the first n_epochs epochs.
    Returns (json object, oracle positions [n_epochs][3][n_tags])."""
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
        out["median_abs_diff_vs_gpu_m"] = float(np.median(np.abs(g - rp).max(axis=1)))
        out["frac_updates_diff_gt_1e-5_m"] = float((np.abs(g - rp).max(axis=1) > 1e-5).mean())
        out["diff_note"] = (f"GPU headline (jacobian = {gpu_jacobian}) against the oracle (numeric) on the sample: " +
                            ("the same Jacobian mode on both sides: SURVEY §8(c)'s 1e-5 m holds for all but the fraction reported (updates whose LM accept / reject "
                             "decision the 1e-7 relative noise of the delta = 1e-9 difference quotient flips; two CPU builds of g2o differ the same way)" if gpu_jacobian == "numeric" else
                             "CROSS-mode, fixed 10 LM iterations, iterates not converged"))
    return out, rp


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


def spawn_ranks(n, argv, backend):
    """One child process per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set (what torch.distributed.run would set);
    rank 0's stdout is this process's stdout.  Returns non-zero unless every rank exits 0; a rank that fails takes the others down
    (they would wait for it in the rendezvous forever)."""
    import socket
    import subprocess
    if backend == "nccl":
        import torch   # (counting devices does not initialise the GPU in this process)
        if torch.cuda.device_count() < n:
            print(f"[bench] --gpus {n} but only {torch.cuda.device_count()} device(s) are visible: refusing to run a smaller job", file=sys.stderr)
            return 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rcs = [None] * n
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
        if any(rc not in (None, 0) for rc in rcs):
            for i, p in enumerate(procs):   # (exactly the processes started above)
                if rcs[i] is None:
                    p.kill()
                    rcs[i] = p.wait()
            break
        time.sleep(0.2)
    if any(rcs):
        print(f"[bench] ranks exited with {rcs}: the {n}-GPU run FAILED", file=sys.stderr)
        return 1
    return 0


class Dist:
    """The rank plumbing every leg shares: barrier + synchronize on both sides of a timed region, max over ranks."""

    def __init__(self, args):
        import torch
        self.torch = torch
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.backend = args.dist_backend
        n_dev = torch.cuda.device_count()
        if args.gpus != self.world:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}: refusing to report a different job than the one asked for")
        if self.backend == "gloo":
            self.local_rank = self.local_rank % max(n_dev, 1)  # rehearsal: ranks may share a device
        elif self.local_rank >= n_dev:
            raise SystemExit(f"bench.py: rank {self.rank} needs GPU {self.local_rank} but only {n_dev} device(s) are visible")
        if self.world > 1:
            import torch.distributed as dist
            torch.cuda.set_device(self.local_rank)
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group("gloo")
        self.dev = torch.device("cuda", self.local_rank)
        torch.cuda.set_device(self.dev)

    def ranks_and_devices(self):
        """(ranks that joined, the HIP device each one runs on): an all-gather over the job's own process group."""
        if self.world == 1:
            return 1, [int(self.local_rank)]
        import torch.distributed as dist
        t = self.torch.zeros(self.world, dtype=self.torch.int64, device=self.dev if self.backend == "nccl" else "cpu")
        t[self.rank] = self.local_rank + 1
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        devs = [int(x) - 1 for x in t.tolist()]
        return sum(1 for d in devs if d >= 0), devs

    def barrier(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if self.world == 1:
            return float(x)
        import torch.distributed as dist
        t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(self, step, warmup, steps):
        """warmup untimed calls of step(i), then exactly `steps` timed ones; returns the max-over-ranks wall seconds."""
        for i in range(warmup):
            step(i)
        self.barrier()
        t0 = time.perf_counter()
        for i in range(warmup, warmup + steps):
            step(i)
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0)

    def close(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()


def reporting_collectives(D, args, out_pos, out_chi2, out_trials, dist_tiles, E, total_steps, B):
    """SURVEY §8(e)(1)-(2), run AFTER the timed region (the solve path itself has no collective):
    (2) all-reduce(SUM) of four reporting scalars — sum chi2, sum LM trials, #non-finite estimates, #ranges the outlier gate
        rejected in the last epoch (localization.cpp:306-313, recomputed here from the last two epochs' outputs);
    (1) all-gather of one result slab per rank (the last epoch: [3 position + 1 chi2][B] doubles) as one consumer of the whole batch
        would ask for it; every rank checks that its own slice of the gathered batch is bit-identical to what it sent and that the
        gathered batch's checksum equals the all-reduced sum of the per-rank checksums.
    RCCL (backend nccl) on device tensors on a GPU node; gloo on host tensors in the one-GPU rehearsal.  Every rank returns the object."""
    import numpy as np
    torch = D.torch
    from localization_amd.synthetic import ANCHORS_8
    w0 = args.warmup * E
    chi = out_chi2[w0:]
    fin = torch.isfinite(chi)
    nonfinite = int((~torch.isfinite(out_pos[w0:])).any(dim=1).sum().item())
    anchors = torch.from_numpy(np.asarray(ANCHORS_8, dtype=np.float64)).to(out_pos.device)
    prior = out_pos[-2]                                                     # [3][B]: the estimate the last epoch's gate looked at
    d_last = dist_tiles[total_steps * E - 1].permute(0, 2, 1).reshape(-1, B)[: anchors.shape[0]].double()   # [M][B]
    pred = (prior[None, :, :] - anchors[:, :, None]).norm(dim=1)           # [M][B]
    gated = int(((pred - d_last).abs() > 1.0).sum().item())
    local = [float(torch.where(fin, chi, torch.zeros_like(chi)).sum().item()), float(out_trials[w0:].double().sum().item()), float(nonfinite), float(gated)]
    slab = torch.cat([out_pos[-1], out_chi2[-1][None, :]], dim=0).contiguous()   # [4][B] f64
    csum_local = int(slab.view(torch.int64).sum().item())                   # wrap-around sum of the bit patterns
    out = {"backend": ("rccl (torch.distributed nccl)" if D.backend == "nccl" else "gloo (rehearsal)") if D.world > 1 else "none (1 rank)",
           "ranks": D.world, "scalars": ["sum_chi2", "sum_lm_trials", "n_nonfinite_estimates", "n_gated_ranges_last_epoch"]}
    if D.world == 1:
        out.update({"all_reduce": {"values": local, "bytes": 0, "ms": 0.0}, "all_gather": {"bytes": 0, "ms": 0.0, "checksum_ok": True},
                    "bytes": 0, "ms": 0.0, "note": "one rank: nothing to exchange (the sums are rank 0's own)"})
        return out
    import torch.distributed as dist
    cdev = D.dev if D.backend == "nccl" else "cpu"
    D.barrier()
    t0 = time.perf_counter()
    v = torch.tensor(local, dtype=torch.float64, device=cdev)
    dist.all_reduce(v, op=dist.ReduceOp.SUM)
    if D.backend == "nccl":
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    send = slab if D.backend == "nccl" else slab.cpu()
    parts = [torch.empty_like(send) for _ in range(D.world)]
    dist.all_gather(parts, send)
    if D.backend == "nccl":
        torch.cuda.synchronize()
    t2 = time.perf_counter()
    cs = torch.tensor([csum_local], dtype=torch.int64, device=cdev)
    dist.all_reduce(cs, op=dist.ReduceOp.SUM)
    whole = torch.cat(parts, dim=1)                                         # [4][world * B]
    ok = bool(torch.equal(parts[D.rank].view(torch.int64), send.view(torch.int64))) and int(whole.view(torch.int64).sum().item()) == int(cs.item())
    okt = torch.tensor([1 if ok else 0], dtype=torch.int64, device=cdev)
    dist.all_reduce(okt, op=dist.ReduceOp.MIN)
    nb_red, nb_gat = v.numel() * 8, send.numel() * 8 * D.world
    out.update({"all_reduce": {"values": [float(x) for x in v.tolist()], "rank0_values": local, "bytes": nb_red, "ms": (t1 - t0) * 1e3},
                "all_gather": {"bytes": nb_gat, "slab_shape_per_rank": list(send.shape), "ms": (t2 - t1) * 1e3, "checksum_ok": bool(okt.item() == 1)},
                "bytes": nb_red + nb_gat, "ms": (t2 - t0) * 1e3,
                "note": "first call of each collective on this process group (connection set-up included); reporting only, outside the timed region"})
    if not out["all_gather"]["checksum_ok"]:
        raise SystemExit("bench.py: the all-gathered result slab does not match what the ranks sent")
    return out


def hbm_roofline(kernel, algo_bytes_per_launch, kern_ms_avg, n_launch, unit_note):
    ach = algo_bytes_per_launch / (kern_ms_avg * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "frac_of_measured_copy_ceiling": ach / HBM_COPY_GBS, "traffic": None,
            "algorithmic_bytes_per_launch": algo_bytes_per_launch, "kernel": kernel, "kernel_ms_avg": kern_ms_avg,
            "launches_timed": n_launch, "note": unit_note}


# ---------------------------------------------------------------------------------------------------------------- legs
def leg_cfg2_mode(D, args, stream, B, E, M, jac, oracle_sample=None):
    """cfg2 in the Jacobian mode the headline did NOT run.  With the default headline (numeric = g2o's central differences,
    types_edge_se3range.h:45-74: the reference's configuration) this is the opt-in analytic fast mode."""
    import numpy as np
    import localization_amd as la
    from localization_amd.synthetic import ANCHORS_8
    torch = D.torch
    steps, warmup = max(2, args.steps), 2   # (as many steps as the headline: five 2.6 ms steps were at the mercy of one host hiccup)
    solver = la.SnapshotSolver(ANCHORS_8, B, maximum_iteration=10, distance_outlier=1.0, jacobian=jac, device=D.local_rank)
    solver.set_positions(stream["init"])
    n_ep = stream["dist_tiles"].shape[0]
    out_pos = torch.empty((E, 3, B), dtype=torch.float64, device=D.dev)
    out_chi2 = torch.empty((E, B), dtype=torch.float64, device=D.dev)

    def step(i):
        k = (i * E) % (n_ep - E + 1)
        solver.solve_device(stream["dist_tiles"][k:k + E], stream["err_tiles"][k:k + E], out_pos, out_chi2, None)

    cross = None
    for i in range(warmup):
        step(i)
        if i == 0 and oracle_sample is not None:   # the first epochs from the initial estimates: what the CPU sample solved
            ne, nt, rp = oracle_sample
            ne = min(ne, E)
            torch.cuda.synchronize()
            cross = float(np.abs(out_pos[:ne, :, :nt].cpu().numpy() - rp[:ne]).max())
    D.barrier()
    solver.timing_begin(steps)
    t0 = time.perf_counter()
    for i in range(warmup, warmup + steps):
        step(i)
    D.barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0)
    n_launch, _, kern_ms = solver.timing_end()
    solver.close()
    upd = float(B) * E
    numeric = jac == "numeric"
    res = {"workload": "BASELINE cfg2 with " + ("g2o's numeric range Jacobians (delta = 1e-9): the reference's exact configuration" if numeric else
                                                 "the analytic range Jacobian: the opt-in fast mode (loc_snapshot_params.jacobian = LOC_JAC_ANALYTIC)"),
           "metric": "localization updates/sec", "value": upd * steps * D.world / elapsed, "unit": "updates/s", "steps": steps,
           "ms_per_step": elapsed / steps * 1e3, "scaling": "weak", "dtype": "f64", "batch_per_gpu": B, "epochs_per_step": E, "jacobian": jac,
           "value_from_kernel_time": upd * D.world / (kern_ms * 1e-3),
           "roofline": hbm_roofline(f"snapshot_lm_kernel<.., JAC = {jac}>", ALGO_BYTES_PER_UPDATE * upd, kern_ms, n_launch,
                                    "120 B/update; VALU-issue bound" + (" (six extra IEEE square roots per edge)" if numeric else ""))}
    if cross is not None:
        res["max_abs_diff_vs_cpu_baseline_m"] = cross
        res["diff_note"] = ("this leg's Jacobian mode against the headline's cpu_baseline sample (oracle, numeric Jacobians): a CROSS-mode figure at the "
                            "reference's fixed 10 LM iterations, i.e. iterates that are not converged (SURVEY §8(c)'s 1e-5 m holds at convergence)")
    return res


def leg_cfg3(D, args):
    """BASELINE cfg3: 8 anchors + IMU rotation prior + antenna lever arm, 6-DoF, B = 65 536 tags per GPU (fusion kernel).
    A step = one launch over 64 resident epochs; odd steps replay them in reverse order (the random walk walked back: still a
    continuous trajectory from where the previous step ended), so no state is re-uploaded and `value` is wall time like every other leg."""
    import numpy as np
    import localization_amd as la
    from localization_amd.synthetic import make_fusion_stream
    torch = D.torch
    B, E = args.batch, 64
    steps, warmup = max(2, min(args.steps, 6)), 2
    s = make_fusion_stream(B, E, seed=args.seed + 1000 * D.rank)
    dist = torch.from_numpy(la.pack_ranges(s["dist"])).to(D.dev)
    err = torch.from_numpy(la.pack_ranges(s["err"])).to(D.dev)
    imu = torch.from_numpy(s["imu"]).to(D.dev)
    fwd = (dist, err, imu)
    rev = tuple(x.flip(0).contiguous() for x in fwd)
    out_pose = torch.empty((E, 7, B), dtype=torch.float64, device=D.dev)
    out_chi2 = torch.empty((E, B), dtype=torch.float64, device=D.dev)
    upd = float(B) * E
    modes = {}
    first_pose = None
    for jac in ("analytic", "numeric"):   # (the reference's configuration last: its first-step output is what the CPU sample is compared with)
        f = la.FusionSolver(s["anchors"], B, antenna_offset=s["offset"], maximum_iteration=10, distance_outlier=3.0, device=D.local_rank, jacobian=jac)
        f.set_poses(s["init"])
        def step(i):
            d, e, q = fwd if i % 2 == 0 else rev
            f.solve_device(d, e, q, out_pose, out_chi2, None)

        step(0)
        torch.cuda.synchronize()
        first_pose = out_pose.clone()
        for i in range(1, warmup):
            step(i)
        D.barrier()
        f.timing_begin(steps)
        t0 = time.perf_counter()
        for i in range(warmup, warmup + steps):
            step(i)
        D.barrier()
        elapsed = D.max_over_ranks(time.perf_counter() - t0)
        n_l, _, avg = f.timing_end()
        kern_ms = D.max_over_ranks(avg)   # HIP events around every timed launch, the slowest rank's average
        f.close()
        modes[jac] = {"elapsed": elapsed, "kern_ms": kern_ms, "launches": n_l}
    num, ana = modes["numeric"], modes["analytic"]
    res = {"workload": "BASELINE cfg3: 8 anchors + IMU rotation prior + antenna lever arm, 6-DoF, g2o-style LM, 10 iterations",
           "metric": "localization updates/sec", "value": upd * steps * D.world / num["elapsed"], "unit": "updates/s", "steps": steps,
           "ms_per_step": num["elapsed"] / steps * 1e3, "scaling": "weak", "dtype": "f64", "batch_per_gpu": B,
           "epochs_per_step": E, "jacobian": "numeric",
           "value_note": "wall time of the timed steps (barrier + synchronize on both sides), numeric Jacobians = the reference's configuration",
           "value_fast_mode": upd * steps * D.world / ana["elapsed"],
           "fast_mode": {"jacobian": "analytic", "ms_per_step": ana["elapsed"] / steps * 1e3, "kernel_ms_avg": ana["kern_ms"],
                         "roofline_frac": ALGO_BYTES_CFG3 * upd / (ana["kern_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS},
           "roofline": hbm_roofline("fusion_lm_kernel<JAC = numeric>", ALGO_BYTES_CFG3 * upd, num["kern_ms"], num["launches"], "248 B/update; VALU-issue bound")}
    if D.rank == 0 and D.world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        nt, ne = 512, 32
        t0 = time.perf_counter()
        rp, _, _, _ = O.fusion_batch(s["anchors"], s["offset"], s["dist"][:ne, :, :nt], s["err"][:ne, :, :nt], s["imu"][:ne, :nt],
                                     s["init"][:, :nt], iterations=10, gate=3.0, jac_mode=O.JAC_NUMERIC_G2O)
        dt = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": nt * ne / dt, "unit": "updates/s", "cores": 1, "kind": "port",
                               "sample": f"first {nt} tags x {ne} epochs ({dt:.1f} s): oracle g2o restatement, numeric Jacobians, 1 thread",
                               "max_abs_diff_vs_gpu": float(np.abs(first_pose[:ne, :, :nt].cpu().numpy() - rp).max()),
                               "diff_note": "numeric Jacobians on both sides (pose entries: metres and quaternion components)"}
    return res


def _window_leg(D, args, wb, anchors, bw_max, name, workload, metric, unit, algo_bytes_per_instance, total_instances, oracle_fn, n_cpu,
                flops_per_instance=None, parity_fn=None, n_parity=0, leg=None):
    """Resident window solves: upload once, `steps` launches from the same initial estimates, HIP-event kernel times."""
    import numpy as np
    import localization_amd as la
    steps, warmup = max(2, min(args.steps, 5)), 1
    B = wb.B
    poses0 = wb.poses.copy()

    def run(jac):
        """upload once, `steps` timed resident launches; leaves the solution of this mode in wb"""
        solver = la.WindowSolver(anchors, B, *wb.caps, maximum_iteration=10, bw_max=bw_max, device=D.local_rank, jacobian=jac)
        wb.poses[:] = poses0
        solver.upload(wb)
        for i in range(warmup):
            solver.solve_resident()
        D.barrier()
        solver.timing_begin(steps)
        t0 = time.perf_counter()
        for i in range(steps):
            solver.solve_resident()
        D.barrier()
        el = D.max_over_ranks(time.perf_counter() - t0)
        n_l, _, k_ms = solver.timing_end()
        solver.download(wb)
        kind = solver.last_kernel_kind()
        solver.close()
        return el, n_l, k_ms, kind

    # the analytic fast mode first, then the reference's configuration (g2o's numeric range Jacobians): `value`, and what stays in wb
    el_ana, _, k_ms_ana, kind_ana = run("analytic")
    poses_analytic = wb.poses[:, :, 9:].copy()
    elapsed, n_launch, kern_ms, kind = run("numeric")
    res = {"workload": workload, "metric": metric, "value": float(total_instances) * steps / elapsed, "unit": unit, "steps": steps,
           "ms_per_step": elapsed / steps * 1e3, "scaling": "strong", "dtype": "f64", "instances_per_gpu": B,
           "instances_total": int(total_instances), "jacobian": "numeric", "mean_lm_trials": float(wb.result[:, 4].mean()),
           "jacobian_note": "g2o's central differences, delta = 1e-9 (types_edge_se3range.h:45-74): the reference's configuration, the library default",
           "elimination_levels": float(wb.result[0, 7] // 65536), "factor_blocks": float(int(wb.result[0, 7]) % 65536),
           "root_supernode_poses": float(round((wb.result[0, 7] % 1.0) * 16)),
           "value_fast_mode": float(total_instances) * steps / el_ana,
           "fast_mode": {"jacobian": "analytic (opt-in)", "kernel": kind_ana, "ms_per_step": el_ana / steps * 1e3, "kernel_ms_avg": k_ms_ana,
                         "roofline_frac": algo_bytes_per_instance * B / (k_ms_ana * 1e-3) / 1e9 / HBM_PEAK_GBS},
           "roofline": hbm_roofline(kind, algo_bytes_per_instance * B, kern_ms, n_launch, name)}
    tr = window_traffic(leg, D.world) if leg else None
    if tr and "stale" in tr:
        res["roofline"]["stale"] = tr["stale"]
    elif tr:
        res["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
        res["roofline"]["traffic_unit"] = "bytes per launch"
        res["roofline"]["traffic_source"] = tr["source"]
        res["roofline"]["traffic_calibration"] = tr.get("calibration")
        res["roofline"]["traffic_rate"] = {"achieved": tr["hbm_bytes_per_launch"] / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": tr["hbm_bytes_per_launch"] / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                           "note": "PMC-measured HBM-side bytes of the same launch shape / live kernel time: what the memory system "
                                                   "actually moved (workspace traffic), against the 8 TB/s peak"}
    if flops_per_instance:
        tf = flops_per_instance * B / (kern_ms * 1e-3) / 1e12
        res["valu_f64"] = {"achieved_tflops": tf, "peak_tflops": F64_VECTOR_PEAK_TFLOPS, "frac": tf / F64_VECTOR_PEAK_TFLOPS,
                           "note": "useful f64 flops per instance (estimate, see DESIGN.md) / kernel time"}
    if D.rank == 0 and D.world == 1 and not args.no_cpu_baseline and n_cpu > 0:
        t0 = time.perf_counter()
        want = oracle_fn(n_cpu)
        dt = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": n_cpu / dt, "unit": unit, "cores": 1, "kind": "port",
                               "sample": f"first {n_cpu} instances ({dt:.1f} s): oracle g2o restatement (dense Cholesky, numeric range Jacobians), 1 thread",
                               "max_abs_diff_vs_gpu_m": float(np.abs(wb.poses[:n_cpu, :, 9:] - want).max()),
                               "median_abs_diff_vs_gpu_m": float(np.median(np.abs(wb.poses[:n_cpu, :, 9:] - want).max(axis=(1, 2)))),
                               "max_abs_diff_vs_gpu_fast_mode_m": float(np.abs(poses_analytic[:n_cpu] - want).max()),
                               "median_abs_diff_vs_gpu_fast_mode_m": float(np.median(np.abs(poses_analytic[:n_cpu] - want).max(axis=(1, 2)))),
                               "diff_note": "oracle: g2o's central differences (the reference's configuration). *_vs_gpu_m: the numeric GPU leg = `value` "
                                            "(the same mode on both sides); *_fast_mode_m: the analytic GPU leg against the same oracle run (cross-mode, "
                                            "fixed 10 LM iterations).  Unconverged iterates of a few instances follow different, equally valid LM "
                                            "accept / reject sequences (DESIGN.md §3)"}
        if parity_fn is not None and n_parity > 0:   # the analytic mode on both sides
            same = parity_fn(n_parity)
            res["cpu_baseline"]["max_abs_diff_fast_mode_vs_analytic_oracle_m"] = float(np.abs(poses_analytic[:n_parity] - same).max())
            res["cpu_baseline"]["fast_mode_sample"] = f"first {n_parity} instances, oracle with analytic Jacobians like the GPU's fast mode"
    wb.poses[:] = poses0
    return res


def leg_cfg5(D, args):
    """BASELINE cfg5: 64-pose sliding windows, range + key-frame pose factors, LM with Cauchy kernels, 16 384 windows in total."""
    import numpy as np
    from localization_amd.sharding import shard_bounds
    sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))
    import bench_window as bw
    total = 16384
    lo, hi = shard_bounds(total, D.rank, D.world)
    wb, graphs, anchors, T = bw.build_pose64(max(hi - lo, 1), np.random.default_rng(args.seed + 7 + 1000 * D.rank), n_graphs=256)
    return _window_leg(D, args, wb, anchors, 8, "7408 B/window (SURVEY §8(d)); the kernel is bound by latency and VALU issue, not bytes",
                       "BASELINE cfg5: 64-pose windows (cfg/uwb_pose.yaml topology), one range + one key-frame EdgeSE3 per pose, Cauchy, 10 LM iterations",
                       "window solves/sec", "windows/s", ALGO_BYTES_CFG5, total,
                       lambda n: bw.oracle_time(graphs, anchors, T, n)[1], 256, leg="cfg5")


def leg_cfg1_windows(D, args):
    """BASELINE cfg1's per-message problem as a batch: the reference's own sliding window (cfg/uwb_only.yaml: 10 poses, one anchor
    range per pose + the zero-range smoothness edges), 65 536 windows in total."""
    import numpy as np
    from localization_amd.sharding import shard_bounds
    sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))
    import bench_window as bw
    total = 65536
    lo, hi = shard_bounds(total, D.rank, D.world)
    import localization_amd as la
    n_mine, n_distinct = max(hi - lo, 1), 4096   # 4096 distinct windows, repeated (generating 65 536 in Python takes 20 s)
    small, graphs, anchors, T = bw.build(min(n_mine, n_distinct), "uwb_only", seed=args.seed + 13 + 1000 * D.rank)
    wb = la.WindowBatch(n_mine, *small.caps)
    for name in ("counts", "poses", "r_idx", "r_val", "p_idx", "p_val", "s_idx", "s_val"):
        src = getattr(small, name)
        getattr(wb, name)[:] = np.resize(src, (n_mine,) + src.shape[1:])   # (cyclic repetition along the batch axis)
    return _window_leg(D, args, wb, anchors, 1, "1360 B/window (30 f64 state in + out, 19 edges x (idx, measurement, information)); the lane-per-window kernels are bound by their workspace traffic",
                       "cfg/uwb_only.yaml's sliding window as a batch: 10 poses, 19 range edges (10 to anchors, 9 smoothness), Cauchy, 10 LM iterations",
                       "window solves/sec", "windows/s", 1360.0, total,
                       lambda n: bw.oracle_time(graphs, anchors, T, n)[1], 2048,
                       parity_fn=lambda n: bw.oracle_time(graphs, anchors, T, n, analytic=True)[1], n_parity=512, leg="cfg1_windows")


def leg_twist_windows(D, args):
    """cfg/uwb_twist.yaml's per-message problem as a batch: 15-pose windows with an anchor range and the smoothness edge per pose and a twist
    EdgeSE3 between consecutive poses (Localization::addTwistEdge, localization.cpp:438-459), 65 536 windows in total — what
    wave6_lm_kernel<JAC, SE3> (one wave per window, full 6x6 coupling blocks) does with the node's EdgeSE3 window when there are many."""
    import numpy as np
    from localization_amd.sharding import shard_bounds
    sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))
    import bench_window as bw
    total = 65536
    lo, hi = shard_bounds(total, D.rank, D.world)
    import localization_amd as la
    n_mine, n_distinct = max(hi - lo, 1), 1024
    small, graphs, anchors, T = bw.build(min(n_mine, n_distinct), "uwb_twist", seed=args.seed + 17 + 1000 * D.rank)
    wb = la.WindowBatch(n_mine, *small.caps)
    for name in ("counts", "poses", "r_idx", "r_val", "p_idx", "p_val", "s_idx", "s_val"):
        src = getattr(small, name)
        getattr(wb, name)[:] = np.resize(src, (n_mine,) + src.shape[1:])
    # algorithmic bytes: 15 poses x 56 B in and out, 29 range edges x 24 B, 14 EdgeSE3 x 232 B (SURVEY §8(d)'s measurement size), chi2
    return _window_leg(D, args, wb, anchors, 1, "5 632 B/window (15 poses x 56 B each way, 29 range edges x 24 B, 14 EdgeSE3 x 232 B, chi2)",
                       "cfg/uwb_twist.yaml's sliding window as a batch: 15 poses, 29 range edges, 14 twist EdgeSE3 (Cauchy), 10 LM iterations",
                       "window solves/sec", "windows/s", 5632.0, total,
                       lambda n: bw.oracle_time(graphs, anchors, T, n)[1], 256,
                       parity_fn=lambda n: bw.oracle_time(graphs, anchors, T, n, analytic=True)[1], n_parity=128, leg="twist_windows")


def leg_cfg1_node(D, args):
    """BASELINE cfg1 the way the drop-in runs it (the reference's own CPU-runnable case): the range messages of the example recording
    (tests/golden/bag_example.npz, decoded from bag/data_example.bag by tools/decode_bag.py) through loc_node_add_range with
    cfg/uwb_only.yaml's parameters — one ten-pose window solve per message (Localization::addRangeEdge + solve(),
    localization.cpp:297-376, 164-192).  A latency figure: rank 0 only."""
    if D.rank != 0:
        return {"skipped": "rank 0 only (a latency figure, not a throughput one)"}
    import ctypes as C
    import numpy as np
    import localization_amd as la
    from localization_amd.node import NodeOutput
    bag = np.load(os.path.join(ROOT, "tests", "golden", "bag_example.npz"))
    ids = [int(i) for i in bag["anchor_ids"]] + [200]
    pos = np.concatenate([bag["anchor_pos"], [[0.0, 0.0, 1.0]]])
    n_msg = len(bag["uwb_stamp"])
    resp = [int(x) for x in bag["uwb_responder"]]; st = [float(x) for x in bag["uwb_stamp"]]
    dist = [float(x) for x in bag["uwb_distance"]]; derr = [float(x) for x in bag["uwb_distance_err"]]
    ant = [int(x) for x in bag["uwb_antenna"]]
    kw = dict(trajectory_length=10, maximum_velocity=5.0, distance_outlier=1.0, maximum_iteration=10, minimum_optimize_error=2000.0, publish_range=True)
    out = {"metric": "latency per range message (one sliding-window solve)", "unit": "ms", "higher_is_better": False, "messages": n_msg,
           "workload": "BASELINE cfg1: bag/data_example.bag's /uwb ranges (fixture), cfg/uwb_only.yaml: 10-pose window, 19 range edges, Cauchy, 10 LM iterations"}
    fr = b"uwb"
    track = {}
    for jac in ("numeric", "analytic"):
        node = la.LocalizationNode(ids, pos, jacobian=jac, device=D.local_rank, **kw)
        L, h, o = node.L, node.h, NodeOutput()
        ref = C.byref(o)
        call, parts, xyz = [], [], []
        for i in range(n_msg):
            t0 = time.perf_counter()
            rc = L.loc_node_add_range(h, 200, resp[i], st[i], dist[i], derr[i], ant[i], fr, ref)
            dt = time.perf_counter() - t0
            if rc < 0:
                raise RuntimeError(L.loc_last_error().decode(errors="replace"))
            if o.solved:
                call.append(dt * 1e3); parts.append(node.last_timing()); xyz.append(list(o.realtime[1:4]))
        node.close()
        c, pt = np.array(call[20:]), np.array(parts[20:])
        track[jac] = np.array(xyz)
        out["reference_config" if jac == "numeric" else "analytic"] = {
            "jacobian": jac, "solves": len(call), "ms_per_message_median": float(np.median(c)), "p99": float(np.percentile(c, 99)), "max": float(c.max()),
            "inside_library_median": {"pack_host": float(np.median(pt[:, 0])), "window_solve_call": float(np.median(pt[:, 1])), "of_which_launch_to_completion": float(np.median(pt[:, 2]))}}
    out["value"] = out["reference_config"]["ms_per_message_median"]
    out["value_note"] = ("the loc_node_add_range call of a message that triggers a solve, host buffers in, poses out (PCIe inclusive), in the REFERENCE's configuration "
                         "(numeric Jacobians); the kernel is wave3_lm_kernel (one wave per window) whenever the window is translation-only, as this recording's is; "
                         "of_which_launch_to_completion: host clock around launch + synchronise (the node's handle runs without HIP events around its kernel: "
                         "~4 us per message; the kernel's own duration: profiles/r04_final/cfg1_node_kernel_stats_head.csv)")
    # the same recording with cfg/uwb_imu.yaml's parameters: IMU orientation priors interleaved in recorded order + an antenna lever arm
    # (6-DoF windows of 12 poses: wave6_lm_kernel)
    kw_imu = dict(trajectory_length=12, maximum_velocity=3.0, distance_outlier=3.0, maximum_iteration=10, minimum_optimize_error=1000.0,
                  publish_range=True, publish_imu=False)
    ant_imu = [[0.05, 0.0, -0.02]] * 3
    ev = [(float(t), 0, i) for i, t in enumerate(bag["uwb_rectime"])] + [(float(t), 1, i) for i, t in enumerate(bag["imu_rectime"]) if t <= bag["uwb_rectime"][-1]]
    ev.sort()
    imu_st = [float(x) for x in bag["imu_stamp"]]
    imu_q = np.ascontiguousarray(bag["imu_q_xyzw"], dtype=np.float64)
    imu_cov = np.ascontiguousarray(np.stack([np.diag(c).ravel() for c in bag["imu_orientation_cov_diag"]]), dtype=np.float64)

    def replay_imu(add_range, add_imu):
        call, xyz = [], []
        for _, kind, i in ev:
            if kind == 0:
                t0 = time.perf_counter()
                solved, p = add_range(i)
                dt = time.perf_counter() - t0
                if solved:
                    call.append(dt * 1e3); xyz.append(p)
            else:
                add_imu(i)
        return np.array(call[20:]), np.array(xyz)

    node = la.LocalizationNode(ids, pos, jacobian="numeric", device=D.local_rank, antenna_offsets=ant_imu, **kw_imu)
    L, h, o = node.L, node.h, NodeOutput()
    ref = C.byref(o)
    dp = C.POINTER(C.c_double)
    frame_imu = b"imu_link"

    def g_range(i):
        if L.loc_node_add_range(h, 200, resp[i], st[i], dist[i], derr[i], ant[i], fr, ref) < 0:
            raise RuntimeError(L.loc_last_error().decode(errors="replace"))
        return bool(o.solved), list(o.realtime[1:4])

    def g_imu(i):
        if L.loc_node_add_imu(h, imu_st[i], imu_q[i].ctypes.data_as(dp), imu_cov[i].ctypes.data_as(dp), frame_imu, ref) < 0:
            raise RuntimeError(L.loc_last_error().decode(errors="replace"))
    c_imu, xyz_imu = replay_imu(g_range, g_imu)
    pt = node.last_timing()
    node.close()
    out["uwb_imu"] = {"workload": "cfg/uwb_imu.yaml on the same recording: 12-pose 6-DoF window, IMU orientation priors, antenna lever arm; numeric Jacobians",
                      "solves": int(len(c_imu) + 20), "ms_per_message_median": float(np.median(c_imu)), "p99": float(np.percentile(c_imu, 99)),
                      "last_solve_inside_library": {"pack_host": pt[0], "window_solve_call": pt[1], "of_which_launch_to_completion": pt[2]}}
    if not args.no_cpu_baseline and D.world == 1:   # (the CPU baseline is timed at N = 1 only: at N > 1 the other ranks would wait for it)
        from oracle import oracle as O
        ora_imu = O.LocalizationOracle(ids, pos, jac_mode=O.JAC_NUMERIC_G2O, antenna_offsets=ant_imu, **kw_imu)

        def o_range(i):
            r = ora_imu.add_range(200, resp[i], st[i], dist[i], derr[i], ant[i], "uwb")
            return r["solved"], list(r["realtime"][1:4])
        oc_imu, oxyz_imu = replay_imu(o_range, lambda i: ora_imu.add_imu(imu_st[i], imu_q[i], imu_cov[i], "imu_link"))
        n = min(len(oxyz_imu), len(xyz_imu))
        out["uwb_imu"]["cpu_baseline"] = {"value": float(np.median(oc_imu)), "unit": "ms", "cores": 1, "kind": "port",
                                          "sample": "the same messages through oracle/localization_oracle.c, numeric Jacobians, one thread",
                                          "median_abs_diff_vs_gpu_m": float(np.median(np.abs(oxyz_imu[:n] - xyz_imu[:n]).max(axis=1))),
                                          "max_abs_diff_vs_gpu_m": float(np.abs(oxyz_imu[:n] - xyz_imu[:n]).max())}
        ora = O.LocalizationOracle(ids, pos, jac_mode=O.JAC_NUMERIC_G2O, **kw)
        oc, oxyz = [], []
        for i in range(n_msg):
            t0 = time.perf_counter()
            r = ora.add_range(200, resp[i], st[i], dist[i], derr[i], ant[i], "uwb")
            dt = time.perf_counter() - t0
            if r["solved"]:
                oc.append(dt * 1e3); oxyz.append(r["realtime"][1:4])
        oc, oxyz = np.array(oc[20:]), np.array(oxyz)
        n = min(len(oxyz), len(track["numeric"]))
        out["cpu_baseline"] = {"value": float(np.median(oc)), "unit": "ms", "cores": 1, "kind": "port",
                               "sample": f"the same {n_msg} messages through oracle/localization_oracle.c (g2o restatement, numeric Jacobians), one thread, median per solved message",
                               "solves": int(len(oc) + 20),
                               "median_abs_diff_vs_gpu_m": float(np.median(np.abs(oxyz[:n] - track["numeric"][:n]).max(axis=1))),
                               "max_abs_diff_vs_gpu_m": float(np.abs(oxyz[:n] - track["numeric"][:n]).max()),
                               "max_abs_diff_vs_gpu_analytic_m": float(np.abs(oxyz[:n] - track["analytic"][:n]).max()),
                               "diff_note": "a closed loop over 1 434 solves, each started from the previous estimates: after the first LM accept / reject "
                                            "decision that the 1e-7 noise of the numeric derivative flips, both follow different, equally valid iterate "
                                            "sequences (the oracle's own two Jacobian modes differ by 1e-3 .. 1e-2 m on this stream; "
                                            "tests/test_gpu_node_parity.py: 1e-6 m before the first flip, ATE difference < 1 mm over the bag)"}
    return out


def leg_node_se3(D, args, which):
    """The drop-in node on windows WITH EdgeSE3 factors, per message, in the reference's configuration (numeric Jacobians), rank 0 only:
      node_uwb_twist       cfg/uwb_twist.yaml (T = 15, 12 LM iterations, vmax 1): ranges to four anchors interleaved with twist messages
                           (addTwistEdge: EdgeSE3 between consecutive poses, localization.cpp:438-459, 560-605); a solve per range message;
      node_uwb_pose_T500   cfg/uwb_pose.yaml AT ITS OWN SIZE (trajectory_length 500, cfg/uwb_pose.yaml:3,8): ranges + key-frame pose factors
                           (addPoseEdge, localization.cpp:254-290); the window is filled with publishing off, then the 500-pose solve is timed.
    Synthetic messages (the example recording has neither topic), the oracle front-end timed on the same messages."""
    if D.rank != 0:
        return {"skipped": "rank 0 only (a latency figure, not a throughput one)"}
    import numpy as np
    import localization_amd as la
    anch = np.array([[3, -3, 0.58], [3, 3, 1.97], [-3, 3, 0.54], [-3, -3, 1.76]], dtype=float)   # the example recording's anchors
    ids = [100, 101, 102, 103, 200]
    pos = np.concatenate([anch, [[0.0, 0.0, 1.0]]])
    rng = np.random.default_rng(args.seed + 31)
    if which == "node_uwb_twist":
        cfg = dict(trajectory_length=15, maximum_velocity=1.0, distance_outlier=1.0, maximum_iteration=12, minimum_optimize_error=1000.0,
                   publish_range=True, publish_twist=False)
        msgs, truth, t = [], np.array([0.3, -0.2, 1.0]), 50.0
        vel = np.array([0.25, 0.1, 0.0])
        for step in range(600):
            t += 1.0 / 60.0
            truth = truth + vel / 60.0
            if step % 2 == 0:
                tw = np.concatenate([vel + rng.normal(0, 0.02, 3), rng.normal(0, 0.01, 3)])
                msgs.append(("twist", t, tw, (np.eye(6) * 1e-2).ravel()))
            else:
                a = (step // 2) % 4
                msgs.append(("range", t, ids[a], float(np.float32(np.linalg.norm(truth - anch[a]) + rng.normal(0, 0.03)))))

        def replay(obj):
            lat, xyz = [], []
            for m in msgs:
                if m[0] == "twist":
                    obj.add_twist(m[1], m[2], m[3], "uwb")
                else:
                    t0 = time.perf_counter()
                    o = obj.add_range(200, m[2], m[1], m[3], 0.055, 0, "uwb")
                    dt = time.perf_counter() - t0
                    if o["solved"]:
                        lat.append(dt * 1e3); xyz.append(np.array(o["realtime"][1:4]))
            return np.array(lat[20:]), np.array(xyz)

        node = la.LocalizationNode(ids, pos, jacobian="numeric", device=D.local_rank, **cfg)
        lat, xyz = replay(node)
        pt, kind = node.last_timing(), node.last_kernel_kind()
        node.close()
        out = {"metric": "latency per range message (one sliding-window solve, EdgeSE3 chain)", "unit": "ms", "higher_is_better": False,
               "workload": "cfg/uwb_twist.yaml: 15-pose window, range edges + twist EdgeSE3 between consecutive poses, Cauchy, 12 LM iterations; numeric Jacobians",
               "value": float(np.median(lat)), "p99": float(np.percentile(lat, 99)), "solves": int(len(lat) + 20), "kernel": kind, "jacobian": "numeric",
               "last_solve_inside_library": {"pack_host": pt[0], "window_solve_call": pt[1], "of_which_launch_to_completion": pt[2]}}
        if not args.no_cpu_baseline and D.world == 1:
            from oracle import oracle as O
            ora = O.LocalizationOracle(ids, pos, jac_mode=O.JAC_NUMERIC_G2O, **cfg)
            olat, oxyz = replay(ora)
            n = min(len(xyz), len(oxyz))
            out["cpu_baseline"] = {"value": float(np.median(olat)), "unit": "ms", "cores": 1, "kind": "port",
                                   "sample": "the same 600 messages through oracle/localization_oracle.c (g2o restatement, numeric Jacobians), one thread, median per solved message",
                                   "median_abs_diff_vs_gpu_m": float(np.median(np.abs(xyz[:n] - oxyz[:n]).max(axis=1))),
                                   "max_abs_diff_vs_gpu_m": float(np.abs(xyz[:n] - oxyz[:n]).max())}
        return out
    # ---- cfg/uwb_pose.yaml at trajectory_length 500
    cfg = dict(trajectory_length=500, maximum_velocity=0.5, distance_outlier=1.0, maximum_iteration=10, minimum_optimize_error=1000.0,
               publish_range=False, publish_pose=False)
    cov = (np.eye(6) * 1e-4).ravel()
    msgs, truth, t = [], np.array([0.0, 0.0, 1.0]), 100.0
    for step in range(520):
        t += 0.05
        truth = truth + np.array([0.004, 0.002 * np.sin(step / 20.0), 0.0])
        rel = np.array([0.004 * ((step % 8) + 1), 0.0, 0.0, 0.0, 0.0, 0.0, 1.0]) + np.concatenate([rng.normal(0, 1e-3, 3), np.zeros(4)])
        msgs.append((t, rel, f"key_{step // 8}", ids[step % 4], float(np.float32(np.linalg.norm(truth - anch[step % 4]) + rng.normal(0, 0.02)))))

    def fill(obj):
        t0 = time.perf_counter()
        for (tt, rel, frame, aid, d) in msgs:
            obj.add_pose(tt, rel, cov, frame)
            obj.add_range(200, aid, tt + 0.01, d, 0.055, 0, "uwb")
        return (time.perf_counter() - t0) * 1e3 / (2 * len(msgs))

    node = la.LocalizationNode(ids, pos, jacobian="numeric", device=D.local_rank, **cfg)
    ingest_ms = fill(node)
    lat, parts = [], []
    g = None
    for rep in range(4):
        t0 = time.perf_counter()
        g = node.solve()
        lat.append((time.perf_counter() - t0) * 1e3)
        parts.append(node.last_timing())
    kind = node.last_kernel_kind()
    gpath = node.path(200)
    first = lat[0]
    out = {"metric": "latency of one 500-pose window solve (cfg/uwb_pose.yaml at its own trajectory_length)", "unit": "ms", "higher_is_better": False,
           "workload": "cfg/uwb_pose.yaml: trajectory_length 500 (3 000 unknowns), one anchor range + one key-frame EdgeSE3 per pose (new key every 8), Cauchy, "
                       "10 LM iterations; numeric Jacobians",
           "value": float(np.median(lat)), "solves_timed": len(lat), "first_solve_ms": float(first), "kernel": kind, "jacobian": "numeric",
           "outer_iterations": int(g["outer_iterations"]), "lm_trials": int(g["lm_trials"]), "ingest_ms_per_message_no_solve": ingest_ms,
           "inside_library_median": {"pack_host": float(np.median([p[0] for p in parts])), "window_solve_call": float(np.median([p[1] for p in parts])),
                                     "of_which_launch_to_completion": float(np.median([p[2] for p in parts]))},
           "value_note": "repeated loc_node_solve calls on the filled window (each continues from the previous estimates, as consecutive messages would)"}
    node.close()
    pm = os.path.join(ROOT, "profiles", "r04_final", "node_pose_T500_pmc.json")
    if os.path.exists(pm):   # (rocprofv3 --pmc passes over tools/dev/node_pose_T500.py = this leg without its CPU baseline: tools/profile_r04.sh t500)
        pj = json.load(open(pm))
        out["pmc"] = {"hbm_side_bytes_per_solve": (2.0 * pj["FETCH_SIZE"] + pj["WRITE_SIZE"]) * 1024.0, "valu_wave_instructions_per_solve": pj["SQ_INSTS_VALU"],
                      "vmem_read_instructions_per_solve": pj["SQ_INSTS_VMEM_RD"], "waves": pj["SQ_WAVES"],
                      "waiting_share_of_wave_cycles": pj["SQ_WAIT_ANY"] / pj["SQ_WAVE_CYCLES"], "valu_share_of_wave_cycles": pj["SQ_ACTIVE_INST_VALU"] / pj["SQ_WAVE_CYCLES"],
                      "source": "profiles/r04_final/node_pose_T500_pmc.json, node_pose_T500_kernel_stats_head.csv (round 4: window_lm_kernel<.., 8 waves>, 3.4 ms per launch under rocprofv3)",
                      "note": "ONE workgroup of eight waves on one CU: the solve is a chain of dependent L2 round trips (H lives in the kernel's global workspace at this size), not bandwidth"}
    if not args.no_cpu_baseline and D.world == 1:
        from oracle import oracle as O
        ora = O.LocalizationOracle(ids, pos, jac_mode=O.JAC_NUMERIC_G2O, **cfg)
        fill(ora)
        olat = []
        for rep in range(4):
            t0 = time.perf_counter()
            ora.solve()
            olat.append((time.perf_counter() - t0) * 1e3)
        opath = ora.path(200)
        out["cpu_baseline"] = {"value": float(np.median(olat)), "unit": "ms", "cores": 1, "kind": "port",
                               "sample": "the same 1 040 messages, then 4 solves of the 500-pose window by oracle/ (g2o restatement with a dense-band Cholesky, numeric Jacobians), one thread",
                               "max_abs_diff_vs_gpu_m": float(np.abs(gpath[:, 1:4] - opath[:, 1:4]).max())}
    return out


def leg_cfg4(D, args):
    """BASELINE cfg4: Monte-Carlo anchor self-calibration, 10 unknown anchors x 256 timesteps per hypothesis, 1 024 hypotheses in total
    (strong scaling: the ranks share them; at N = 1 one GPU solves all 1 024, and the 128-hypothesis share of an 8-GPU job is timed too)."""
    import numpy as np
    import localization_amd as la
    from localization_amd.sharding import shard_bounds
    sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))
    import bench_window as bw
    total = 1024
    lo, hi = shard_bounds(total, D.rank, D.world)
    n_mine, n_distinct = max(hi - lo, 1), 32   # 32 distinct hypotheses, repeated (building 1 024 x 2 815 edges in Python takes a minute)
    small, graphs, anchors, nv = bw.build_selfcal(min(n_mine, n_distinct), np.random.default_rng(args.seed + 11))

    def tiled(n):
        wb = la.WindowBatch(n, *small.caps)
        for name in ("counts", "poses", "r_idx", "r_val", "p_idx", "p_val", "s_idx", "s_val"):
            src = getattr(small, name)
            getattr(wb, name)[:] = np.resize(src, (n,) + src.shape[1:])
        return wb

    note = "33 248 B per hypothesis per LM iteration x 10 iterations (SURVEY §8(d))"
    work = "BASELINE cfg4: anchor self-calibration, 256 tag poses + 10 unknown anchors per hypothesis (798 position unknowns, 2815 range edges), 10 LM iterations"
    res = _window_leg(D, args, tiled(n_mine), anchors, nv - 1, note, work, "hypothesis solves/sec", "solves/s", ALGO_BYTES_CFG4_PER_IT * 10, total,
                      lambda n: np.array([bw.oracle_selfcal(g, 256, 10) for g in graphs[:n]]), 8, leg="cfg4")
    res["lm_iterations_per_s"] = res["value"] * 10
    if D.world == 1:   # one GPU's share of the 8-GPU job SURVEY §8(e) describes: 128 hypotheses (half of the chip's CUs get a workgroup)
        args2 = argparse.Namespace(**vars(args)); args2.no_cpu_baseline = True
        share = _window_leg(D, args2, tiled(128), anchors, nv - 1, note, work, "hypothesis solves/sec", "solves/s", ALGO_BYTES_CFG4_PER_IT * 10, 128,
                            lambda n: None, 0)
        res["share_of_8_gpu_job_128_hypotheses"] = {k: share[k] for k in ("value", "ms_per_step", "value_fast_mode")}
        res["share_of_8_gpu_job_128_hypotheses"]["kernel_ms_avg"] = share["roofline"]["kernel_ms_avg"]
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=65536, help="tags per GPU")
    ap.add_argument("--epochs", type=int, default=128, help="epochs per launch (step)")
    ap.add_argument("--jacobian", default="numeric", choices=["analytic", "numeric"],
                    help="numeric = g2o's central differences: the reference's configuration and every library default (the headline); "
                         "analytic = the opt-in fast mode")
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
    ap.add_argument("--legs", default="all", help="secondary legs: all, none, or a comma list of cfg2_analytic,cfg2_numeric,cfg3,cfg5,cfg4,cfg1_windows,cfg1_node,node_uwb_twist,node_uwb_pose_T500")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Launched bare (`python bench.py --gpus N`): start the N ranks ourselves.  This happens BEFORE anything touches the
        # GPU in this process (no torch import yet); the children are ordinary subprocesses, never an exec of this one.
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:], args.dist_backend))

    import numpy as np
    import torch
    import localization_amd as la
    from localization_amd.synthetic import ANCHORS_8, make_snapshot_stream_torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    D = Dist(args)
    world, rank, dev = D.world, D.rank, D.dev
    n_gpus = world
    ranks_joined, rank_devices = D.ranks_and_devices()
    if ranks_joined != args.gpus:
        raise SystemExit(f"bench.py: {ranks_joined} of {args.gpus} ranks joined")

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
                               lanes_per_instance=args.lpi, block_threads=args.block, device=D.local_rank)
    solver.set_positions(stream["init"])
    out_pos = torch.empty((total_steps * E, 3, B), dtype=torch.float64, device=dev)
    out_chi2 = torch.empty((total_steps * E, B), dtype=torch.float64, device=dev)
    out_trials = torch.empty((total_steps * E, B), dtype=torch.uint8, device=dev)

    def step(i):
        sl = slice(i * E, (i + 1) * E)
        solver.solve_device(dist_t[sl], err_t[sl], out_pos[sl], out_chi2[sl], out_trials[sl])

    for i in range(args.warmup):
        step(i)
    D.barrier()
    solver.timing_begin(args.steps)
    t0 = time.perf_counter()
    for i in range(args.warmup, total_steps):
        step(i)
    D.barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0)
    n_launch, kern_ms_total, kern_ms_avg = solver.timing_end()

    updates_per_launch = float(B) * E
    total_updates = updates_per_launch * args.steps * n_gpus
    value = total_updates / elapsed
    achieved_gbs = ALGO_BYTES_PER_UPDATE * updates_per_launch / (kern_ms_avg * 1e-3) / 1e9

    # ---- the reporting collectives, AFTER the timed region (the solve path has none) -----------------------
    coll = reporting_collectives(D, args, out_pos, out_chi2, out_trials, dist_t, E, total_steps, B)

    res = None
    oracle_sample = None
    if rank == 0:
        trials_mean = float(out_trials[args.warmup * E:].cpu().numpy().mean())
        err_last = torch.from_numpy(np.sqrt(((out_pos[-1].cpu().numpy() - stream["truth_last"].cpu().numpy()) ** 2).sum(axis=0)))
        # HBM bytes per launch from the committed PMC passes (rocprofv3 cannot run inside this process); they apply to the
        # default launch shape AND to the kernel source they were measured on: a hash of the sources is stored with them
        pm, stale, ceiling = None, None, None
        prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(prof) and (B, E) == (65536, 128):
            try:
                with open(prof) as f:
                    pj = json.load(f)
                if pj.get("kernel_source_sha256") == kernel_source_hash(SNAPSHOT_KERNEL_SOURCES):
                    pm, ceiling = pj.get("modes", {}).get(args.jacobian), pj.get("measured_issue_ceiling_lane_slots_per_s")
                else:
                    stale = ("stale: profiles/hbm_traffic.json was measured on a different snapshot_kernel.hip (source hash mismatch); "
                             "PMC-derived fields are withheld until the counters are re-collected")
            except Exception:
                pm = None
        numeric = args.jacobian == "numeric"
        res = {
            "metric": "localization updates/sec (8-anchor UWB, batch=65k)",
            "value": value, "unit": "updates/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic", "ranks_joined": ranks_joined, "rank_devices": rank_devices,
            "config": {"workload": "BASELINE cfg2: synthetic 8-anchor UWB, 3-DoF position, Cauchy range factors, "
                                   "g2o-style LM with the reference's fixed 10 iterations, outlier gate 1 m",
                       "batch_per_gpu": B, "epochs_per_step": E, "updates_per_step_per_gpu": int(updates_per_launch),
                       "anchors": M, "lm_iterations": 10, "jacobian": args.jacobian,
                       "jacobian_note": ("g2o's central differences, delta = 1e-9 (EdgeSE3Range has no linearizeOplus, types_edge_se3range.h:45-74): "
                                         "the reference's configuration and the default of every library entry point") if numeric else
                                        "the opt-in analytic fast mode (NOT the library default)",
                       "lanes_per_tag": solver.lanes_per_instance, "sharding": f"tags split over {n_gpus} GPU(s), no collective"},
            "lm_iterations_per_s": value * 10,
            "mean_lm_trials_per_update": trials_mean,
            "median_err_vs_truth_m": float(err_last.median().item()),
            "frac_err_gt_0p5m": float((err_last > 0.5).double().mean().item()),
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS, "frac_of_measured_copy_ceiling": achieved_gbs / HBM_COPY_GBS,
                         "traffic": pm["hbm_bytes_per_launch"] if pm else None, "traffic_unit": "bytes per launch",
                         "traffic_source": pm["source"] if pm else None,
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_UPDATE * updates_per_launch,
                         "kernel": f"snapshot_lm_kernel<.., JAC = {args.jacobian}>", "kernel_ms_avg": kern_ms_avg, "launches_timed": n_launch,
                         "algorithmic_bytes_per_update": ALGO_BYTES_PER_UPDATE,
                         "binding_resource": "valu_issue",
                         "note": "north_star names the HBM roofline, so `bound`/`frac` price the algorithmic bytes against 8 TB/s; "
                                 "at fp64 and the reference's 10 LM iterations the kernel is VALU-issue bound (see valu_issue / "
                                 "valu_f64 in this object and DESIGN.md), the HBM fraction is small by construction"},
        }
        if stale:
            res["roofline"]["stale"] = stale
        if pm and pm.get("f64_flop_per_launch"):
            tf = pm["f64_flop_per_launch"] / (kern_ms_avg * 1e-3) / 1e12
            res["roofline"]["valu_f64"] = {"achieved_tflops": tf, "peak_tflops": F64_VECTOR_PEAK_TFLOPS, "frac": tf / F64_VECTOR_PEAK_TFLOPS,
                                           "note": "f64 add+mul+2*fma+trans lane-ops per launch (PMC, profiles/) / live kernel time"}
        if pm and pm.get("issue_lane_slots_per_launch") and ceiling:
            rate = pm["issue_lane_slots_per_launch"] / (kern_ms_avg * 1e-3)
            res["roofline"]["valu_issue"] = {"achieved_lane_slots_per_s": rate, "measured_ceiling": ceiling, "frac": rate / ceiling,
                                             "valu_wave_insts_per_launch": pm.get("valu_wave_insts_per_launch"),
                                             "note": "(VALU + SALU wave-instructions per launch) x 64 lanes (PMC) / live kernel time, against the "
                                                     "measured one-wave-per-SIMD issue ceiling (tools/fp64_probe.hip): what actually bounds this kernel"}
        if coll is not None:
            res["collectives"] = coll
        if not args.no_cpu_baseline and n_gpus == 1:
            n_t, n_e = min(args.cpu_tags, B), min(args.cpu_epochs, E * total_steps)
            res["cpu_baseline"], rp = cpu_baseline(ANCHORS_8, dist_t, err_t, stream["init"], n_t, n_e, out_pos, M, args.jacobian)
            oracle_sample = (n_e, n_t, rp)
            try:  # a secondary figure: a host that refuses worker processes must not cost the bench its JSON line
                res["cpu_baseline_all_cores"] = cpu_baseline_all_cores(ANCHORS_8, dist_t, err_t, stream["init"], args.cpu_tags // 2,
                                                                       min(args.cpu_epochs, E * total_steps), M)
            except Exception as exc:  # noqa: BLE001
                res["cpu_baseline_all_cores"] = {"value": None, "error": f"{type(exc).__name__}: {exc}"}
    solver.close()
    del out_pos, out_chi2, out_trials

    # ---- secondary legs (every rank takes part: they carry their own barriers) ------------------------------------------------
    other = "analytic" if args.jacobian == "numeric" else "numeric"
    all_legs = ["cfg2_" + other, "cfg3", "cfg5", "cfg4", "cfg1_windows", "cfg1_node", "node_uwb_twist", "node_uwb_pose_T500", "twist_windows"]
    want = [] if args.legs == "none" else (all_legs if args.legs == "all" else args.legs.split(","))
    legs = {}
    for name in want:
        try:
            if name in ("cfg2_numeric", "cfg2_analytic"):
                out = leg_cfg2_mode(D, args, stream, B, E, M, name[5:], oracle_sample if name[5:] != args.jacobian else None)
            elif name == "cfg3":
                out = leg_cfg3(D, args)
            elif name == "cfg5":
                out = leg_cfg5(D, args)
            elif name == "cfg4":
                out = leg_cfg4(D, args)
            elif name == "cfg1_windows":
                out = leg_cfg1_windows(D, args)
            elif name == "twist_windows":
                out = leg_twist_windows(D, args)
            elif name == "cfg1_node":
                out = leg_cfg1_node(D, args)
            elif name in ("node_uwb_twist", "node_uwb_pose_T500"):
                out = leg_node_se3(D, args, name)
            else:
                out = {"error": "unknown leg"}
        except Exception as exc:  # noqa: BLE001 — a secondary leg must not cost the bench its headline line
            if world > 1:
                raise   # (with several ranks a one-sided failure would deadlock the next barrier: fail loudly instead)
            out = {"error": f"{type(exc).__name__}: {exc}"}
        out["n_gpus"] = n_gpus
        legs[name] = out
        torch.cuda.empty_cache()
    if rank == 0:
        if legs:
            res["legs"] = legs
        oth = legs.get("cfg2_" + other, {})
        key = "value_fast_mode" if other == "analytic" else "value_reference_config"
        res[key] = oth.get("value")
        res["other_jacobian_mode"] = {"jacobian": other, "value": oth.get("value"), "ms_per_step": oth.get("ms_per_step"),
                                      "roofline_frac": (oth.get("roofline") or {}).get("frac"),
                                      "note": "`value` is the numeric-Jacobian mode every library entry point defaults to (the reference's configuration); "
                                              "this is the opt-in analytic mode (loc_snapshot_params.jacobian)" if other == "analytic" else
                                              "`value` was run with --jacobian analytic (opt-in fast mode); this is the library default"} if oth else None
        print(json.dumps(res))
    D.close()


if __name__ == "__main__":
    main()
