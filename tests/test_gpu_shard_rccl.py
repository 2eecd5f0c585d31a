"""The RCCL calls of the sharded path, executed for real: ONE rank on cuda:0 under backend "nccl" (RCCL refuses two
ranks on one device, and the box has one GPU), in a fresh child process.  With one rank every collective is a copy to
oneself, so this says nothing about xGMI -- what it pins is that the calls shard.py makes are ones RCCL accepts:
all_to_all_single of int64 sizes and of uint8 records with uneven split lists on device buffers, the all_gather of the
verbatim blobs, the in-place all_reduce of the flat uint64 count buffer viewed as int64, and that tables, count vectors
and round statistics come out of them unchanged."""
import os
import subprocess
import sys

import pytest

from tests import util

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys, numpy as np
sys.path.insert(0, os.environ["MIRGE_ROOT"])
import torch, torch.distributed as dist
from mirge_amd import capi, shard, synth
from tests import util
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29547", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
libs, _ = synth.make_libraries(synth.Profile.small(), 71)
S = 2
samples = [synth.make_reads(libs, 40000, 72 + s, zipf_perm_seed=3 + s) for s in range(S)]
rng = np.random.default_rng(6)
for seqs, lens in samples:
    for i in rng.integers(0, len(lens), size=150):
        seqs[i, int(rng.integers(0, lens[i]))] = ord("N")
samples = [util.with_verbatim(libs, seqs, lens, seed=s, n=100) for s, (seqs, lens) in enumerate(samples)]

def run(sharded):
    with capi.Context(None, S, 0) as ctx:
        util.load_libs_into(ctx, libs)
        for s, rd in enumerate(samples):
            ctx.collapse_seqs(rd, s)
        moved = shard.merge_uniques(ctx, dist, dev) if sharded else 0
        ctx.annotate()
        if sharded:
            shard.allreduce_counts(ctx, dist, dev, always=True)
        return util.gpu_result_dict(ctx), ctx.counts_flat(), [ctx.round_stats(r + 1)[:2] for r in range(5)], moved

want, want_flat, want_stats, _ = run(False)
got, flat, stats, moved = run(True)
assert moved > 10000, moved                                  # every record went through the all-to-all
assert got == want
assert sum(1 for k in want if not util.fits_key(k)) > 50
assert np.array_equal(flat, want_flat)
assert [tuple(int(x) for x in r) for r in stats] == [tuple(int(x) for x in r) for r in want_stats]
dist.destroy_process_group()
print("rccl one-rank ok", moved)
'''


def test_rccl_calls_with_one_rank(tmp_path):
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    env = dict(os.environ, MIRGE_ROOT=util.ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, str(w)], env=env, timeout=300, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "rccl one-rank ok" in out.stdout
