"""N > 1 path on the CPU: two gloo ranks each run the hot path (oracle stand-in for the device) on their
shard and all-reduce the flat count vectors; the result equals the unsharded run."""
import os
import subprocess
import sys

import numpy as np

from mirge_amd import shard
from tests import util

WORKER = r'''
import os, sys, numpy as np
sys.path.insert(0, os.environ["MIRGE_ROOT"])
import torch.distributed as dist
from mirge_amd import shard, synth
from oracle import oracle as orc
from tests import util
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
libs, _ = synth.make_libraries(synth.Profile.tiny(), 55)
seqs, lens = synth.make_reads(libs, 6000, 56)
lo, hi = shard.shard_bounds(len(lens), rank, world)
res = orc.run_path(util.oracle_libs(libs, orc), [(seqs[lo:hi], lens[lo:hi])], use_index=True, maxlen=124)
flat = np.concatenate([res[k].ravel() for k in ("quant", "iscan", "cat", "lenhist", "tuniq")]).astype(np.uint64)
tot = shard.allreduce_flat(flat, dist)
if rank == 0:
    np.save(os.environ["MIRGE_OUT"], tot)
dist.destroy_process_group()
'''


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 100, 12345):
        for w in (1, 2, 3, 8):
            b = [shard.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
    assert sorted(sum((shard.samples_of_rank(48, r, 8) for r in range(8)), [])) == list(range(48))


def test_two_rank_allreduce_equals_single(tmp_path, oracle):
    from mirge_amd import synth
    out = str(tmp_path / "tot.npy")
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    env = dict(os.environ, MIRGE_ROOT=util.ROOT, MIRGE_OUT=out, MASTER_ADDR="127.0.0.1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", "29533", str(w)], env=env, timeout=300)
    tot = np.load(out)
    libs, _ = synth.make_libraries(synth.Profile.tiny(), 55)
    seqs, lens = synth.make_reads(libs, 6000, 56)
    res = oracle.run_path(util.oracle_libs(libs, oracle), [(seqs, lens)], use_index=True, maxlen=124)
    one = np.concatenate([res[k].ravel() for k in ("quant", "iscan", "cat", "lenhist", "tuniq")]).astype(np.uint64)
    assert np.array_equal(tot[:-1], one[:-1])        # everything but trimmed_uniq is additive over shards
    assert tot[-1] >= one[-1]
