"""Replica sharding across the GPUs of one node.

Replicas are fully independent cities (no shared mutable state), so the hot path shards with NO
data-path collective: rank g of G owns the contiguous replica range returned by ``shard`` and a
private copy of the static tables.  The single collective is a ``sum`` all-reduce of the int64[8]
aggregate counters (RCCL over xGMI on GPUs - backend "nccl" - or gloo on CPU in the tests);
64 bytes per call, latency-bound, once per episode.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

ALLREDUCE_CALLS = 0      # collectives issued by this process (reported by bench.py)
RAW_COUNTER_NAMES = ("orders", "rejects", "wait_sum", "matched_value", "evals", "arrivals", "dispatches", "dispatch_cost")


def shard(total_replicas: int, world_size: int, rank: int) -> Tuple[int, int]:
    """(first_replica, count) of ``rank``; the first ``total % world`` ranks take one extra."""
    if not (0 <= rank < world_size):
        raise Exception("rank %d outside world of %d" % (rank, world_size))
    base, extra = divmod(total_replicas, world_size)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def allreduce_counters(tensor, group=None):
    """In-place sum of an int64[8] counter tensor over all ranks (no-op without a process group; with a group the
    collective is issued even for a world of one, so a single-rank launch exercises the same RCCL path)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        if tensor.is_cuda and dist.get_backend(group) != "nccl":
            host = tensor.cpu()                  # gloo (tests: two ranks on one GPU) reduces host memory
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            tensor.copy_(host)
        else:
            dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)
        global ALLREDUCE_CALLS
        ALLREDUCE_CALLS += 1
    return tensor


def max_over_ranks(value: float, device=None) -> float:
    """The slowest rank's time: what the job's throughput is quoted on."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return value
    if device is None and dist.get_backend() == "nccl":
        device = "cuda"               # RCCL reduces device tensors only
    if dist.get_backend() != "nccl":
        device = None
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
