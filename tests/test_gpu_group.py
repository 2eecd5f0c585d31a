"""Multi-GPU inside the library (include/mirge_hip.h: mirge_group_*) and the batch ingest (mirge_collapse_fastq_batch),
through the C ABI on the GPU.

* one rank through RCCL from C (real ncclSend / ncclRecv / ncclAllReduce calls, every one a copy to itself -- the box
  has one GPU): tables byte-identical to the golden ones the unmodified reference wrote;
* 2 and 3 logical ranks on ONE device with the exchanges staged through host memory (MIRGE_GROUP_HOST_STAGED: the
  same group code, RCCL refuses two ranks on a device): union of the ranks' per-unique tables, count vectors and round
  statistics equal to the one-context run and to the oracle, reads no key holds included;
* a batch of .gz / plain sample files, inflated by several host threads, equal to the files collapsed one by one.
The 8-GPU exchange itself has never run (no multi-GPU box on this pool): the call pattern is what is checked here."""
import gzip
import os

import numpy as np
import pytest

from mirge_amd import capi, synth
from tests import util

pytestmark = pytest.mark.gpu


def _tables_equal_golden(outdir, case):
    exp = case["expected"]
    for f in ("miR.Counts.csv", "miR.RPM.csv"):
        assert open(os.path.join(outdir, f)).read() == open(os.path.join(exp, f)).read(), f
    for f in ("mapped.csv", "unmapped.csv"):
        assert util.rows(os.path.join(outdir, f)) == util.rows(os.path.join(exp, f)), f


@pytest.mark.parametrize("name", ["toy", "synth_tiny", "synth_verbatim"])
def test_one_rank_through_rccl_equals_golden(tmp_path, name):
    case = util.golden_case(name)
    S = len(case["fastqs"])
    with capi.Group(case["species"], S, 1) as g:                  # RCCL: ncclCommInitAll over device 0
        tot, kept, _ = g.collapse_fastq_batch(case["fastqs"], list(range(S)))
        g.annotate()
        out = str(tmp_path / "out")
        g.write_tables(os.path.join(case["species"], "merges.csv"), out, case["names"])
        _tables_equal_golden(out, case)
        st = case["stats"]
        assert g.n_unique() == st["nUnique"]
        for r in range(5):
            assert g.round_stats(r + 1)[1] == st["annotStats"][r]["readsAligned"]
        assert [int(x) for x in tot] == [q["trimmedReads"] for q in st["quantStats"]]
        assert g.timing()["merge_ms"] > 0


@pytest.mark.parametrize("world", [2, 3])
def test_logical_ranks_host_staged_equal_one_context(oracle, tmp_path, world):
    libs, merges = synth.make_libraries(synth.Profile.small(), 5200 + world)
    S = 3
    samples = [synth.make_reads(libs, 30000, 5210 + s, zipf_perm_seed=40 + s) for s in range(S)]
    for seqs, lens in samples:                                     # reads with N (48-byte class) in every sample
        rng = np.random.default_rng(len(lens))
        for i in rng.integers(0, len(lens), size=200):
            seqs[i, int(rng.integers(0, lens[i]))] = ord("N")
    with capi.Context(None, S, 0) as ctx:
        util.load_libs_into(ctx, libs)
        for s, (seqs, lens) in enumerate(samples):
            ctx.collapse_seqs(seqs, s, lens)
        ctx.annotate()
        want = util.gpu_result_dict(ctx)
        wcv = ctx.counts()
        wstats = [ctx.round_stats(r + 1)[:2] for r in range(5)]
    with capi.Group(None, S, world, devices=[0] * world, flags=capi.GROUP_HOST_STAGED) as g:
        for i, nm in enumerate(synth.LIB_NAMES):
            g.load_library(i, libs[nm].names, libs[nm].text, libs[nm].offs)
        for s, (seqs, lens) in enumerate(samples):
            g.collapse_seqs(seqs, s, lens)                          # contiguous chunks of every sample
        g.annotate()
        got = {}
        for r in range(world):
            part = util.gpu_result_dict(g.ctx(r))
            assert not (set(part) & set(got)), "a sequence lives on two ranks after the merge"
            got.update(part)
            assert len(part) > len(want) // (world * 2), "the owner hash leaves a rank nearly empty"
        cv = g.counts()
        stats = [g.round_stats(r + 1)[:2] for r in range(5)]
        assert g.n_unique() == len(want)
    assert got == want
    for k in ("quant", "iscan", "cat", "tuniq", "lenhist"):
        assert np.array_equal(cv[k], wcv[k]), k
    assert stats == wstats
    res = oracle.run_path(util.oracle_libs(libs, oracle), samples, use_index=True, nthreads=8, maxlen=capi.MAX_READ_LEN)
    assert got == util.oracle_result_dict(res)


def test_logical_ranks_whole_sample_files_and_verbatim_reads(tmp_path):
    """Whole sample files per rank (file i -> rank i mod W), .gz among them, reads no key holds among them; the group's
    tables against the golden ones of the reference."""
    case = util.golden_case("synth_verbatim")
    S = len(case["fastqs"])
    paths = []
    for i, p in enumerate(case["fastqs"]):
        if i % 2 == 0:
            q = str(tmp_path / (os.path.basename(p) + ".gz"))
            with open(p, "rb") as f, gzip.open(q, "wb") as o:
                o.write(f.read())
            paths.append(q)
        else:
            paths.append(p)
    with capi.Group(case["species"], S, 2, devices=[0, 0], flags=capi.GROUP_HOST_STAGED) as g:
        g.collapse_fastq_batch(paths, list(range(S)))
        g.annotate()
        out = str(tmp_path / "out")
        g.write_tables(os.path.join(case["species"], "merges.csv"), out, case["names"])
        _tables_equal_golden(out, case)
        assert not os.path.exists(os.path.join(out, ".rank1"))


def test_batch_of_gz_files_equals_one_by_one(tmp_path):
    libs, _ = synth.make_libraries(synth.Profile.tiny(), 5300)
    S = 6
    paths, n_each = [], []
    for s in range(S):
        seqs, lens = synth.make_reads(libs, 5000 + 700 * s, 5310 + s)
        p = str(tmp_path / ("s%d.fastq" % s))
        synth.write_fastq_fast(p, seqs, lens)
        n_each.append(len(lens))
        if s % 3 != 2:                                              # four .gz files (one of them two gzip members), two plain
            data = open(p, "rb").read()
            with open(p + ".gz", "wb") as o:
                if s == 1:
                    cut = data.index(b"\n@", len(data) // 2) + 1
                    o.write(gzip.compress(data[:cut]) + gzip.compress(data[cut:]))
                else:
                    o.write(gzip.compress(data))
            p += ".gz"
        paths.append(p)
    with capi.Context(None, S, 0) as ctx:
        util.load_libs_into(ctx, libs)
        for s, p in enumerate(paths):
            assert ctx.collapse_fastq(p, s) == n_each[s]
        ctx.annotate()
        want = util.gpu_result_dict(ctx); wcv = ctx.counts()
        ctx.reset()
        tot, kept, _ = ctx.collapse_fastq_batch(paths, list(range(S)))
        assert [int(x) for x in tot] == n_each
        ctx.annotate()
        assert util.gpu_result_dict(ctx) == want
        cv = ctx.counts()
        for k in ("quant", "iscan", "cat", "tuniq", "lenhist"):
            assert np.array_equal(cv[k], wcv[k]), k
        # a file that does not exist: the call fails loudly, names it, and leaves the context usable
        with pytest.raises(capi.MirgeError) as e:
            ctx.collapse_fastq_batch(paths[:2] + [str(tmp_path / "nope.fastq.gz")], [0, 1, 2])
        assert "nope.fastq.gz" in str(e.value)
        ctx.reset()
        ctx.collapse_fastq_batch(paths, list(range(S)))
        ctx.annotate()
        assert util.gpu_result_dict(ctx) == want


def test_group_refuses_what_it_cannot_do():
    with pytest.raises(capi.MirgeError) as e:
        capi.Group(None, 1, 2, devices=[0, 0])                      # two ranks on one device through RCCL
    assert "one device" in str(e.value)
    with pytest.raises(capi.MirgeError):
        capi.Group(None, 1, 0)
