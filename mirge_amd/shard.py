"""Multi-GPU plumbing: reads shard across ranks (one process per GPU), every rank collapses and
annotates its own shard, and the count vectors are summed with ONE all-reduce.

Annotation is a pure function of the sequence and every table number is a sum over reads, so the
all-reduce alone gives exact count vectors.  Two things are sums over UNIQUES instead: trimmed_uniq
(miRge.pl:468) and the per-round readsProcessed/readsAligned (miRge.pl:410-413); a sequence present in
two shards would be counted (and annotated) in both.  merge_uniques() is the optional exchange step that
makes them exact: one all-to-all of (key, sample, count) records by key owner between collapse and
cascade, after which every sequence lives on exactly one rank.
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


def allreduce_counts(ctx, dist=None, device=None, always=False):
    """The one collective of a sharded run: sum this context's flat count buffer (count vectors, category totals,
    length histogram, trimmed_uniq, per-round readsProcessed / readsAligned) over all ranks and install the sums.
    On GPUs the buffer is packed, reduced (RCCL) and read back from device memory; under gloo (CPU rehearsal) it goes
    through the host.  A one-rank group has nothing to sum and returns at once, unless `always` (the one-GPU test of
    the RCCL calls themselves)."""
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not always):
        return
    import torch
    if device is not None and str(device).startswith("cuda") and dist.get_backend() == "nccl":
        t = torch.empty(ctx.counts_flat_len(), dtype=torch.int64, device=device)
        ctx.counts_flat_get_dev(t.data_ptr())
        dist.all_reduce(t)
        torch.cuda.synchronize(device)
        ctx.counts_flat_set_dev(t.data_ptr())
    else:
        ctx.set_counts_flat(allreduce_flat(ctx.counts_flat(), dist, None))


REC16, REC48 = 24, 56          # bytes per record (include/mirge_hip.h: mirge_shard_export)


def merge_uniques(ctx, dist, device):
    """All-to-all of the local unique tables by key owner.  ctx: capi.Context after its
    collapse calls; afterwards it holds exactly the sequences this rank owns, with counts summed over all
    ranks, ready for annotate().  Two collectives: the record counts, then the records (16-byte and
    48-byte classes in one buffer each).  Verbatim uniques (reads no key holds: a host-side table) follow in an
    all-gather of byte blobs -- only when some rank has any."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    on_gpu = device is not None and str(device).startswith("cuda")
    # gloo (the one-GPU rehearsal of the N > 1 path) has no all-to-all over device buffers: the exchange is then staged
    # through host tensors; under RCCL the records travel GPU to GPU
    wire = device if (not on_gpu or dist.get_backend() == "nccl") else None
    n16, n48 = ctx.shard_export_sizes(world)
    send16 = torch.empty(max(int(n16.sum()) * REC16, 8), dtype=torch.uint8, device=device)
    send48 = torch.empty(max(int(n48.sum()) * REC48, 8), dtype=torch.uint8, device=device)
    ctx.shard_export(world, send16.data_ptr(), send48.data_ptr())
    blob = ctx.verb_export()
    mine = torch.tensor(np.stack([n16, n48, np.full(world, len(blob))], axis=1).astype(np.int64), device=wire)   # [world, 3]: what I send to each rank (+ my blob size)
    theirs = torch.empty_like(mine)                                                           # what each rank sends to me
    dist.all_to_all_single(theirs, mine)
    r = theirs.cpu().numpy()
    r16, r48, rblob = r[:, 0], r[:, 1], r[:, 2]
    if wire is not device:
        send16, send48 = send16.cpu(), send48.cpu()
    recv16 = torch.empty(max(int(r16.sum()) * REC16, 8), dtype=torch.uint8, device=wire)
    recv48 = torch.empty(max(int(r48.sum()) * REC48, 8), dtype=torch.uint8, device=wire)
    dist.all_to_all_single(recv16[:int(r16.sum()) * REC16], send16[:int(n16.sum()) * REC16],
                           output_split_sizes=[int(x) * REC16 for x in r16], input_split_sizes=[int(x) * REC16 for x in n16])
    dist.all_to_all_single(recv48[:int(r48.sum()) * REC48], send48[:int(n48.sum()) * REC48],
                           output_split_sizes=[int(x) * REC48 for x in r48], input_split_sizes=[int(x) * REC48 for x in n48])
    if wire is not device:
        recv16, recv48 = recv16.to(device), recv48.to(device)
    if on_gpu:
        torch.cuda.synchronize(device)
    ctx.shard_import(recv16.data_ptr(), int(r16.sum()), recv48.data_ptr(), int(r48.sum()))
    if int(rblob.sum()):                     # every rank sees the same sizes: all of them take this branch, or none
        cap = int(rblob.max())
        pad = torch.zeros(cap, dtype=torch.uint8)
        pad[:len(blob)] = torch.from_numpy(np.ascontiguousarray(blob))
        pad = pad.to(wire) if wire is not None else pad
        got = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(got, pad)
        ctx.verb_clear()
        for src in range(world):
            ctx.verb_import(got[src][:int(rblob[src])].cpu().numpy(), rank, world)
    return int(r16.sum()) + int(r48.sum())
