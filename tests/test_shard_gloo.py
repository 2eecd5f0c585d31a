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


MERGE_WORKER = r'''
import ctypes, os, sys, zlib, numpy as np
sys.path.insert(0, os.environ["MIRGE_ROOT"])
import torch, torch.distributed as dist
from mirge_amd import shard, synth
from oracle import oracle as orc
from tests import util
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
libs, _ = synth.make_libraries(synth.Profile.tiny(), 57)
S = 2
samples = [util.with_verbatim(libs, *synth.make_reads(libs, 4000, 58 + s), seed=s) for s in range(S)]
all_reads = sorted({r for rd in samples for r in rd if util.fits_key(r)})
ident = {r: i for i, r in enumerate(all_reads)}          # the double's 16-byte key: (sequence id, 0)

class TableDouble:
    """capi.Context's exchange surface on host memory: lets shard.merge_uniques run under gloo without a GPU."""
    def __init__(self):
        self.tab = {}
        self.verb = {}                                       # reads no key holds: (string, sample) -> count
    def add(self, reads, sample):
        for r in reads:
            if util.fits_key(r):
                self.tab[(ident[r], sample)] = self.tab.get((ident[r], sample), 0) + 1
            else:
                self.verb[(r, sample)] = self.verb.get((r, sample), 0) + 1
    def verb_export(self):
        out = b"".join(np.array([len(r), s, c], dtype=np.uint32).tobytes() + r.encode() for (r, s), c in sorted(self.verb.items()))
        return np.frombuffer(out, dtype=np.uint8)
    def verb_clear(self):
        self.verb = {}
    def verb_import(self, blob, rank, w):
        b = bytes(blob); p = 0
        while p < len(b):
            n, s, c = np.frombuffer(b[p:p + 12], dtype=np.uint32); p += 12
            r = b[p:p + int(n)].decode(); p += int(n)
            if zlib.crc32(r.encode()) % w == rank:
                self.verb[(r, int(s))] = self.verb.get((r, int(s)), 0) + int(c)
    def _owner(self, sid, w):
        return zlib.crc32(all_reads[sid].encode()) % w
    def shard_export_sizes(self, w):
        n = np.zeros(w, dtype=np.uint64)
        for (sid, s) in self.tab:
            n[self._owner(sid, w)] += 1
        return n, np.zeros(w, dtype=np.uint64)
    def shard_export(self, w, p16, p48):
        rec = sorted((self._owner(sid, w), sid, s, c) for (sid, s), c in self.tab.items())
        a = np.array([[sid, 0, s | (c << 32)] for _, sid, s, c in rec], dtype=np.uint64).reshape(-1, 3)
        ctypes.memmove(p16, a.ctypes.data, a.nbytes)
    def shard_import(self, p16, n16, p48, n48):
        assert n48 == 0
        a = np.zeros((n16, 3), dtype=np.uint64)
        ctypes.memmove(a.ctypes.data, p16, a.nbytes)
        self.tab = {}
        for sid, _, tail in a:
            k = (int(sid), int(tail) & 0xFFFFFFFF)
            self.tab[k] = self.tab.get(k, 0) + (int(tail) >> 32)

t = TableDouble()
for s, rd in enumerate(samples):
    lo, hi = shard.shard_bounds(len(rd), rank, world)
    t.add(rd[lo:hi], s)
shard.merge_uniques(t, dist, None)
held = torch.tensor([len(t.verb)]); dist.all_reduce(held)
assert int(held) == len({(r, s) for s, rd in enumerate(samples) for r in rd if not util.fits_key(r)})     # every verbatim entry on exactly one rank
owned = {sid for sid, _ in t.tab}
per_sample = [[all_reads[sid] for (sid, s), c in sorted(t.tab.items()) if s == smp for _ in range(c)] +
              [r for (r, s), c in sorted(t.verb.items()) if s == smp for _ in range(c)] for smp in range(S)]
res = orc.run_path(util.oracle_libs(libs, orc), per_sample, use_index=True, maxlen=1000)
flat = np.concatenate([res[k].ravel() for k in ("quant", "iscan", "cat", "lenhist", "tuniq", "stats")]).astype(np.uint64)
tot = shard.allreduce_flat(flat, dist)
own = torch.zeros(len(all_reads), dtype=torch.int64); own[sorted(owned)] = 1
dist.all_reduce(own)
if rank == 0:
    np.save(os.environ["MIRGE_OUT"], tot)
    assert int(own.min()) == 1 and int(own.max()) == 1      # every sequence on exactly one rank
dist.destroy_process_group()
'''


def test_two_rank_unique_merge_makes_everything_additive(tmp_path, oracle):
    """shard.merge_uniques under gloo (host-memory table double): after the all-to-all by key owner, trimmed_uniq and
    the per-round unique counts are sums over ranks too."""
    from mirge_amd import synth
    out = str(tmp_path / "tot.npy")
    w = tmp_path / "worker.py"
    w.write_text(MERGE_WORKER)
    env = dict(os.environ, MIRGE_ROOT=util.ROOT, MIRGE_OUT=out, MASTER_ADDR="127.0.0.1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", "29534", str(w)], env=env, timeout=300)
    tot = np.load(out)
    libs, _ = synth.make_libraries(synth.Profile.tiny(), 57)
    samples = [util.with_verbatim(libs, *synth.make_reads(libs, 4000, 58 + s), seed=s) for s in range(2)]
    res = oracle.run_path(util.oracle_libs(libs, oracle), samples, use_index=True, maxlen=1000)
    one = np.concatenate([res[k].ravel() for k in ("quant", "iscan", "cat", "lenhist", "tuniq", "stats")]).astype(np.uint64)
    assert np.array_equal(tot, one)
