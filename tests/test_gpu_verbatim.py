"""Reads no 2-bit key holds -- longer than 124 nt, lower / mixed case, IUPAC codes, '.', other printable characters --
through every ingest path of the library, against the CPU oracle: collapsed by exact string (miRge.pl:306), matched as
bowtie's parser would see them (a/c/g/t = the base, anything else = N: hypothesis H4, parity with bowtie unpinned),
counted and exported like every other unique.  The golden case synth_verbatim (tables written by the unmodified
miRge.pl) runs in test_gpu_parity / test_gpu_tools with the other golden cases."""
import numpy as np
import pytest

from mirge_amd import capi, synth
from tests import util

pytestmark = pytest.mark.gpu


def fastq_of(reads):
    return "".join("@r%d\n%s\n+\n%s\n" % (i, r, "I" * len(r)) for i, r in enumerate(reads)).encode()


@pytest.fixture(scope="module")
def setup(oracle):
    libs, _ = synth.make_libraries(synth.Profile.small(), 301)
    samples = [util.with_verbatim(libs, *synth.make_reads(libs, 6000, 302 + s), seed=s, n=150) for s in range(2)]
    res = oracle.run_path(util.oracle_libs(libs, oracle), samples, use_index=True, nthreads=8, maxlen=capi.MAX_READ_LEN)
    return libs, samples, res


def check(ctx, res):
    got, exp = util.gpu_result_dict(ctx), util.oracle_result_dict(res)
    assert got.keys() == exp.keys()
    bad = [(k, got[k], exp[k]) for k in exp if got[k] != exp[k]]
    assert not bad, bad[:5]
    cv = ctx.counts()
    for k in ("quant", "iscan", "cat", "tuniq", "lenhist"):
        assert np.array_equal(cv[k], res[k]), k
    for r in range(5):
        assert ctx.round_stats(r + 1)[:2] == (int(res["stats"][r, 0]), int(res["stats"][r, 1])), r
    n_odd = sum(1 for k in exp if not util.fits_key(k))
    assert n_odd > 100 and sum(1 for k, v in exp.items() if not util.fits_key(k) and v[0] > 0) > 30
    assert any(len(k) > 124 and v[0] > 0 for k, v in exp.items())


@pytest.mark.parametrize("path", ["seqs", "fastq_mem", "fastq_file"])
def test_verbatim_reads_equal_oracle(setup, tmp_path, path):
    libs, samples, res = setup
    with capi.Context(None, 2, 0) as ctx:
        util.load_libs_into(ctx, libs)
        for s, rd in enumerate(samples):
            if path == "seqs":
                ctx.collapse_seqs(rd, s)
            elif path == "fastq_mem":
                assert ctx.collapse_fastq_mem(fastq_of(rd), s) == len(rd)
            else:
                p = tmp_path / ("s%d.fastq" % s); p.write_bytes(fastq_of(rd))
                assert ctx.collapse_fastq(str(p), s) == len(rd)
        ctx.annotate()
        check(ctx, res)
        # a second run on the same context starts from nothing
        ctx.reset()
        ctx.collapse_seqs(samples[0][:50], 0)
        ctx.annotate()
        assert ctx.n_unique() == len(set(samples[0][:50]))


def test_raw_path_takes_long_and_odd_lines(setup, oracle):
    """--adapter none on raw reads: quality trimming (nothing to trim at 'I'), the >= 16 nt filter, then the same table."""
    libs, samples, _ = setup
    kept = [[r for r in rd if len(r) >= 16] for rd in samples]
    res = oracle.run_path(util.oracle_libs(libs, oracle), kept, use_index=True, nthreads=8, maxlen=capi.MAX_READ_LEN)
    with capi.Context(None, 2, 0) as ctx:
        util.load_libs_into(ctx, libs)
        for s, rd in enumerate(samples):
            assert ctx.collapse_raw_fastq_mem(fastq_of(rd), s, "none", 0, 0)[:2] == (len(rd), len(kept[s]))
        ctx.annotate()
        check(ctx, res)


def test_150nt_raw_reads_with_adapter(oracle):
    """The case ADVICE names: 150-nt raw small-RNA reads with the adapter still on.  Trimmed on the device to the
    inserts; the table equals the one the restatement of trim_file.py + cutadapt gives (parity with cutadapt unpinned)."""
    import collections
    from oracle import trim as otrim
    from tests.test_trim_cpu import raw_case
    data = raw_case(41, n=20000, read_len=150, odd=0.05)
    want, total, _, reported = otrim.trim_fastq(data, synth.ILLUMINA_ADAPTER, 0, 0)
    with capi.Context(None, 1, 0) as ctx:
        assert ctx.collapse_raw_fastq_mem(data, 0, "illumina", 0, 0) == (total, len(want), reported)
        got = {k: v[3][0] for k, v in util.gpu_result_dict(ctx).items()}
    assert got == dict(collections.Counter(x.decode() for x in want))
    assert any(not util.fits_key(k) for k in got)


def test_refused_reads():
    with capi.Context(None, 1, 0) as ctx:
        for bad, code in [("ACGT ACGT", capi.MIRGE_E_FORMAT), ("ACGT\tA", capi.MIRGE_E_FORMAT), ("ACGT\r", capi.MIRGE_E_FORMAT),
                          ("A" * (capi.MAX_READ_LEN + 1), capi.MIRGE_E_LIMIT)]:
            for how in ("seqs", "fastq"):
                ctx.reset()
                with pytest.raises(capi.MirgeError) as e:
                    if how == "seqs":
                        ctx.collapse_seqs(["ACGTACGT", bad], 0)
                    else:
                        ctx.collapse_fastq_mem(fastq_of(["ACGTACGT", bad]), 0)
                assert e.value.code == code, (bad, how)
        ctx.reset()
        ctx.collapse_seqs(["A" * capi.MAX_READ_LEN, "acgu", "ACGT"], 0)        # the longest read the library takes
        assert ctx.n_unique() == 3


@pytest.mark.parametrize("path", ["seqs", "fastq_mem"])
def test_large_batches_of_verbatim_reads_threaded_equals_serial(monkeypatch, path):
    """Tens of thousands of reads no key holds per call (150-nt reads, --adapter none): the host table takes a batch with
    all the host's threads, a shard of the table each (mirge_hip.hip: Verb::add_batch).  Uniques, their numbering, the
    (unique, sample) entries and the counts must be those of the serial loop (MIRGE_HOST_THREADS=1) and of a plain Counter:
    duplicates inside a batch, across the two calls of a sample and across samples."""
    import collections
    rng = np.random.default_rng(77)
    base = ["".join("ACGTacgtN."[x] for x in rng.integers(0, 10, size=int(rng.integers(125, 260)))) for _ in range(9000)]
    def draw(n, lo, hi):
        return [base[int(i)] for i in np.minimum(rng.zipf(1.3, size=n) - 1 + lo, hi - 1)]
    calls = [(0, draw(30000, 0, 6000)), (0, draw(20000, 3000, 9000)), (1, draw(25000, 0, 9000))]      # (sample, reads)
    want = [collections.Counter(), collections.Counter()]
    for s, rd in calls:
        want[s].update(rd)
    exports = []
    for nthr in ("1", "16"):
        monkeypatch.setenv("MIRGE_HOST_THREADS", nthr)
        with capi.Context(None, 2, 0) as ctx:
            for s, rd in calls:
                if path == "seqs":
                    ctx.collapse_seqs(rd, s)
                else:
                    assert ctx.collapse_fastq_mem(fastq_of(rd), s) == len(rd)
            useqs, ulens, _, _, _ = ctx.export_uniques()
            uid, smp, cnt = ctx.export_counts()
            order = [useqs[u, :ulens[u]].tobytes().decode() for u in range(len(ulens))]
            exports.append((order, uid.tolist(), smp.tolist(), cnt.tolist()))
            assert ctx.n_unique() == len(set(want[0]) | set(want[1]))
            per = [collections.Counter(), collections.Counter()]
            for u, s, c in zip(uid.tolist(), smp.tolist(), cnt.tolist()):
                assert order[u] not in per[s], "a (unique, sample) pair has two entries"
                per[s][order[u]] = c
            assert per == want
    # (from FASTQ text the device lists the reads no key holds in the order its threads met them: the numbering then varies
    # from run to run whatever the host does, only the table's content is fixed)
    if path == "seqs":
        assert exports[0] == exports[1], "the threaded batch numbers uniques or entries differently from the serial loop"
