"""Multi-GPU sharding of the batch: tags are independent least-squares problems, so the batch is
split into contiguous slices, one per rank / GPU; the anchor table and config are replicated.  There is NO collective
on the solve path.  torch.distributed (backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in CPU tests) is used
only for (1) the barrier / max-over-ranks around a timed region, (2) an optional all-gather of the per-rank result
slabs when one consumer wants the whole batch, (3) an all-reduce of a few reporting scalars.
"""
import ctypes as C

import numpy as np

from ._lib import check, lib


class Shard(C.Structure):
    """include/localization_amd.h: loc_shard"""
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("device", C.c_int32), ("reserved", C.c_int32),
                ("lo", C.c_int64), ("hi", C.c_int64)]


def shard_bounds(total, rank, world):
    """Contiguous slice [lo, hi) of `total` tags owned by `rank`: ceil-divided, last ranks may be shorter/empty.
    The rule lives in the C ABI (loc_shard_bounds) so that C++ callers, this harness and bench.py cannot disagree."""
    lo, hi = C.c_int64(), C.c_int64()
    check(lib().loc_shard_bounds(int(total), int(rank), int(world), C.byref(lo), C.byref(hi)))
    return int(lo.value), int(hi.value)


def shard_plan(total, world, devices_per_node):
    """loc_shard_plan: the whole job's descriptor table, [(rank, device, lo, hi)] for rank 0 .. world - 1."""
    arr = (Shard * int(world))()
    check(lib().loc_shard_plan(int(total), int(world), int(devices_per_node), arr))
    return [(s.rank, s.device, int(s.lo), int(s.hi)) for s in arr]


def shard_array(x, rank, world, axis=-1):
    lo, hi = shard_bounds(x.shape[axis], rank, world)
    idx = [slice(None)] * x.ndim
    idx[axis] = slice(lo, hi)
    return x[tuple(idx)]


def shard_window_batch(wb, rank, world):
    """The contiguous slice of a WindowBatch's instances owned by `rank` (windows / hypotheses are independent problems:
    SURVEY §8(e) — cfg5: 2 048 windows per GPU at G = 8, cfg4: 128 hypotheses per GPU).  Returns (shard, lo, hi)."""
    from .window import WindowBatch
    lo, hi = shard_bounds(wb.B, rank, world)
    part = WindowBatch(max(hi - lo, 1), *wb.caps)
    part.B = hi - lo
    for name in ("counts", "poses", "r_idx", "r_val", "p_idx", "p_val", "s_idx", "s_val", "result"):
        getattr(part, name)[: hi - lo] = getattr(wb, name)[lo:hi]
    return part, lo, hi


def barrier_and_max(elapsed_s, device=None):
    """barrier, then the max of `elapsed_s` over ranks (the whole job is as slow as its slowest rank)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(elapsed_s)
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_gather_results(local, total, axis=-1):
    """All-gather per-rank result slabs (torch tensors, sharded along `axis` by shard_bounds) into the full batch.
    Slabs are padded to the common shard size for the collective and trimmed afterwards."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    per = -(-int(total) // world)
    x = local.movedim(axis, 0).contiguous()
    pad = per - x.shape[0]
    if pad > 0:
        x = torch.cat([x, x.new_zeros((pad,) + tuple(x.shape[1:]))], dim=0)
    out = [torch.empty_like(x) for _ in range(world)]
    dist.all_gather(out, x)
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(total, r, world)
        parts.append(out[r][: hi - lo])
    return torch.cat(parts, dim=0).movedim(0, axis)


def all_reduce_scalars(values, device=None):
    """Sum a handful of reporting scalars (sum chi2, #published, #gated ...) over ranks."""
    import torch
    import torch.distributed as dist
    v = torch.tensor([float(x) for x in values], dtype=torch.float64, device=device or "cpu")
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
    return [float(x) for x in v.tolist()]
