"""Multi-GPU layout of a batch: one process per GPU, contiguous shards, no data exchange.

Items are independent, so the path shards trivially: rank k owns items
[lo_k, hi_k) of every SoA array.  The only collective is the all-reduce (sum) of the 4-counter
tally -- 32 bytes over RCCL/xGMI (backend "nccl" on ROCm), or gloo in CPU tests.
"""
from __future__ import annotations


def shard_bounds(n: int, rank: int, world: int):
    """Contiguous block of ceil(n / world) items per rank (the last ranks may get fewer or none)."""
    if world < 1 or not (0 <= rank < world) or n < 0:
        raise ValueError("bad shard request")
    per = -(-n // world)
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def shard(arrays, rank: int, world: int):
    """Slice every (n, width) array of a batch to this rank's block."""
    n = len(arrays[0])
    lo, hi = shard_bounds(n, rank, world)
    return [a[lo:hi] for a in arrays], (lo, hi)


def allreduce_tally(tally, group=None):
    """Sum the per-rank tallies in place (torch tensor of 4 integers); no-op without a process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(tally, op=dist.ReduceOp.SUM, group=group)
    return tally


def verify_sharded(verify_fn, arrays, rank: int, world: int, group=None):
    """verify_fn(*local_arrays) -> (status, tally) on this rank's shard; returns
    (local status, (lo, hi), global tally)."""
    local, bounds = shard(arrays, rank, world)
    status, tally = verify_fn(*local)
    return status, bounds, allreduce_tally(tally, group)
