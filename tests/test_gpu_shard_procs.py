"""The N > 1 path with real processes on the GPU: two ranks (fresh child processes, both on cuda:0, gloo between them
as on a one-GPU box -- RCCL refuses two ranks on one device) shard one read set, merge their unique tables
(shard.merge_uniques: export -> all-to-all -> import), annotate what they own, and sum the flat count buffer
(shard.allreduce_counts).  Rank 0 compares the union of the per-rank tables, the summed count vectors and the summed
per-round statistics with the one-rank run of the same reads."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, pickle, sys, numpy as np
sys.path.insert(0, os.environ["MIRGE_ROOT"])
import torch, torch.distributed as dist
from mirge_amd import capi, shard, synth
from tests import util
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
libs, _ = synth.make_libraries(synth.Profile.small(), 61)
S = 2
samples = [synth.make_reads(libs, 30000, 62 + s, zipf_perm_seed=9 + s) for s in range(S)]
rng = np.random.default_rng(5)
for seqs, lens in samples:                                  # some reads of the 48-byte key class
    for i in rng.integers(0, len(lens), size=150):
        seqs[i, int(rng.integers(0, lens[i]))] = ord("N")
samples = [util.with_verbatim(libs, seqs, lens, seed=s, n=120) for s, (seqs, lens) in enumerate(samples)]    # + reads no key holds (a host-side table)
with capi.Context(None, S, 0) as ctx:
    util.load_libs_into(ctx, libs)
    for s, rd in enumerate(samples):
        lo, hi = shard.shard_bounds(len(rd), rank, world)
        ctx.collapse_seqs(rd[lo:hi], s)
    shard.merge_uniques(ctx, dist, dev)
    ctx.annotate()
    mine = util.gpu_result_dict(ctx)
    shard.allreduce_counts(ctx, dist, dev)
    flat = ctx.counts_flat()
    stats = [ctx.round_stats(r + 1)[:2] for r in range(5)]
parts = [None] * world
dist.all_gather_object(parts, mine)
if rank == 0:
    pickle.dump({"parts": parts, "flat": flat, "stats": stats}, open(os.environ["MIRGE_OUT"], "wb"))
dist.destroy_process_group()
'''


def test_two_processes_one_gpu_merge_and_allreduce(tmp_path):
    import pickle
    from mirge_amd import capi, synth
    out = str(tmp_path / "res.pkl")
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    env = dict(os.environ, MIRGE_ROOT=util.ROOT, MIRGE_OUT=out, MASTER_ADDR="127.0.0.1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", "29541", str(w)], env=env, timeout=300)
    res = pickle.load(open(out, "rb"))          # written by this test's own child a moment ago
    libs, _ = synth.make_libraries(synth.Profile.small(), 61)
    samples = [synth.make_reads(libs, 30000, 62 + s, zipf_perm_seed=9 + s) for s in range(2)]
    rng = np.random.default_rng(5)
    for seqs, lens in samples:
        for i in rng.integers(0, len(lens), size=150):
            seqs[i, int(rng.integers(0, lens[i]))] = ord("N")
    samples = [util.with_verbatim(libs, seqs, lens, seed=s, n=120) for s, (seqs, lens) in enumerate(samples)]
    with capi.Context(None, 2, 0) as one:
        util.load_libs_into(one, libs)
        for s, rd in enumerate(samples):
            one.collapse_seqs(rd, s)
        one.annotate()
        want = util.gpu_result_dict(one)
        want_flat = one.counts_flat()
        want_stats = [one.round_stats(r + 1)[:2] for r in range(5)]
    got = {}
    for part in res["parts"]:
        assert not (set(part) & set(got))                   # every sequence lives on exactly one rank
        got.update(part)
    assert got == want
    assert sum(1 for k in want if not util.fits_key(k)) > 100
    assert np.array_equal(res["flat"], want_flat)           # count vectors, trimmed_uniq and round statistics: all sums now
    assert [tuple(int(x) for x in r) for r in res["stats"]] == [tuple(int(x) for x in r) for r in want_stats]
