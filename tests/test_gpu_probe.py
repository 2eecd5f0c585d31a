"""The probe path (kernels.h: k_probe_round -- both-strand slot tables, gapped seed tables, hand-over to the generic
kernel) on the GPU against the CPU oracle.  The libraries of the other GPU tests are too small to get slot tables
(they are built from kb = 11 up), so these cases force them onto small libraries and sweep the index k-mer length:
kb = 9..12 with reads of 16..2*kb-1 seed nt exercises the gapped tables, kb = 6 the plain two-piece lookups, and
reads shorter than a k-mer the hand-over."""
import numpy as np
import pytest

from mirge_amd import capi, synth
from tests import util
from tests.test_gpu_parity import _compare_with_oracle

pytestmark = pytest.mark.gpu


# lds = "1": libraries whose text fits the LDS take the compact tables and k_lds_round (off by default), the others the probe kernel
@pytest.mark.parametrize("lds", ["0", "1"])
@pytest.mark.parametrize("kb,bits", [(6, "4194304"), (9, "4194304"), (10, "0"), (12, "4194304"), (12, "0")])
def test_default_cascade_through_slot_tables(oracle, monkeypatch, kb, bits, lds):
    monkeypatch.setenv("MIRGE_LDS", lds)
    monkeypatch.setenv("MIRGE_SLOTS_MIN_KB", "4")
    monkeypatch.setenv("MIRGE_KB", str(kb))
    monkeypatch.setenv("MIRGE_BITS2_MAX", bits)           # 0: slots read without asking the occupancy bits
    libs, _ = synth.make_libraries(synth.Profile.small(), 700 + kb)
    samples = [synth.make_reads(libs, 50000, 710 + kb + s, zipf_perm_seed=5 + s) for s in range(2)]
    seqs, lens = samples[0]
    rng = np.random.default_rng(kb)
    for i in rng.integers(0, len(lens), size=500):         # reads cut below a k-mer / a gapped window: the hand-over
        lens[i] = int(rng.integers(1, 16))
    res = _compare_with_oracle(oracle, libs, samples)
    for r in range(1, 6):
        assert (res["ann_round"] == r).sum() > 0, r


@pytest.mark.parametrize("lds", ["0", "1"])
@pytest.mark.parametrize("seed,kb", [(11, 7), (12, 10), (13, 12), (14, 11)])
def test_random_cascades_through_slot_tables(oracle, monkeypatch, seed, kb, lds):
    """As test_gpu_parity.py::test_random_cascades_vs_oracle, with every library carrying slot tables: rounds of one or
    two pieces run on the probe path, the others (and the 48-byte key class) on the generic kernel."""
    monkeypatch.setenv("MIRGE_LDS", lds)
    monkeypatch.setenv("MIRGE_SLOTS_MIN_KB", "4")
    monkeypatch.setenv("MIRGE_KB", str(kb))
    rng = np.random.default_rng(2000 + seed)
    libs, _ = synth.make_libraries(synth.Profile.small(), 400 + seed)
    samples = [synth.make_reads(libs, 20000, 410 + seed)]
    seqs, lens = samples[0]
    for i in rng.integers(0, len(lens), size=300):
        seqs[i, int(rng.integers(0, lens[i]))] = ord("N")
    rounds = []
    for _ in range(int(rng.integers(3, 9))):
        len_lt = int(rng.choice([0, 0, 22, 26, 30])); len_gt = int(rng.choice([0, 0, 18, 25]))
        rounds.append((int(rng.integers(0, 4)), len_lt, len_gt, int(rng.integers(0, 3)), int(rng.integers(0, 3)),
                       int(rng.choice([12, 16, 20, 24, 28, 30])), int(rng.integers(0, 3)), int(rng.integers(0, 4))))
    with capi.Context(None, 1, 0) as ctx:
        util.load_libs_into(ctx, libs)
        ctx.set_rounds(rounds)
        ctx.collapse_seqs(seqs, 0, lens)
        ctx.annotate()
        got = util.gpu_result_dict(ctx)
        cv = ctx.counts()
        stats = [ctx.round_stats(r + 1)[:2] for r in range(len(rounds))]
    res = oracle.run_path(util.oracle_libs(libs, oracle), samples, rounds=rounds, use_index=True, nthreads=8, maxlen=capi.MAX_READ_LEN)
    assert got == util.oracle_result_dict(res), rounds
    for k in ("quant", "iscan", "cat", "tuniq", "lenhist"):
        assert np.array_equal(cv[k], res[k]), (k, rounds)
    assert stats == [(int(a), int(b)) for a, b in res["stats"]][:len(rounds)], rounds
    assert sum(b for _, b in stats) > 1000


@pytest.mark.parametrize("lds", ["0", "1"])
@pytest.mark.parametrize("seed", [21, 22, 23])
def test_cascades_that_mix_probe_and_generic_rounds(oracle, monkeypatch, seed, lds):
    """Libraries with and without slot tables in one cascade: a probe round leaves split survivor lists (kernels.h:
    ListOut) and a list of the generic kernel's leftovers, a round whose library has no slot tables reads both with the
    generic kernel and leaves one dense list, which the next probe round reads -- every hand-over between the two list
    layouts, against the oracle."""
    import math
    libs, _ = synth.make_libraries(synth.Profile.small(), 800 + seed)
    kbs = {name: min(14, max(4, int(round(math.log2(max(len(l.text) + l.n, 16)) / 2.0)) + 1)) for name, l in libs.items()}
    cut = sorted(kbs.values())[2]                            # the two largest libraries get slot tables, the others none
    assert min(kbs.values()) < cut <= max(kbs.values()), kbs
    monkeypatch.setenv("MIRGE_SLOTS_MIN_KB", str(cut))
    monkeypatch.setenv("MIRGE_LDS", lds)
    rng = np.random.default_rng(3000 + seed)
    samples = [synth.make_reads(libs, 30000, 810 + seed)]
    seqs, lens = samples[0]
    for i in rng.integers(0, len(lens), size=300):
        seqs[i, int(rng.integers(0, lens[i]))] = ord("N")
    for i in rng.integers(0, len(lens), size=300):           # reads below a k-mer: handed from the probe to the generic kernel
        lens[i] = int(rng.integers(4, 16))
    order = list(rng.permutation(4)) + list(rng.permutation(4))
    rounds = [(int(lib), 0, 0, int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.choice([12, 16, 20, 28])),
               int(rng.integers(0, 3)), int(rng.integers(0, 3))) for lib in order[:int(rng.integers(5, 9))]]
    with capi.Context(None, 1, 0) as ctx:
        util.load_libs_into(ctx, libs)
        ctx.set_rounds(rounds)
        ctx.collapse_seqs(seqs, 0, lens)
        ctx.annotate()
        got = util.gpu_result_dict(ctx)
        cv = ctx.counts()
        stats = [ctx.round_stats(r + 1)[:2] for r in range(len(rounds))]
    res = oracle.run_path(util.oracle_libs(libs, oracle), samples, rounds=rounds, use_index=True, nthreads=8, maxlen=capi.MAX_READ_LEN)
    assert got == util.oracle_result_dict(res), rounds
    for k in ("quant", "iscan", "cat", "tuniq", "lenhist"):
        assert np.array_equal(cv[k], res[k]), (k, rounds)
    assert stats == [(int(a), int(b)) for a, b in res["stats"]][:len(rounds)], rounds


@pytest.mark.parametrize("kb,lds", [(4, "0"), (7, "0"), (11, "0"), (7, "1")])
def test_repeat_rich_libraries(oracle, monkeypatch, kb, lds):
    """Libraries with repeated k-mers (synth.Profile.small_repeats: tRNA-like families of 10-50 near-identical members,
    low-complexity stretches, mRNA isoforms sharing exons) instead of i.i.d. bases: bucket lists of tens to thousands of
    entries, most hits with several equally good references.  kb = 4 puts > 4096 entries into most buckets of the larger
    libraries, so those reads are handed to the generic kernel; every assignment and every ambiguity flag against the
    oracle."""
    monkeypatch.setenv("MIRGE_LDS", lds)
    monkeypatch.setenv("MIRGE_SLOTS_MIN_KB", "4")
    monkeypatch.setenv("MIRGE_KB", str(kb))
    libs, _ = synth.make_libraries(synth.Profile.small_repeats(), 900 + kb)
    samples = [synth.make_reads(libs, 30000, 910 + kb)]
    S = 1
    with capi.Context(None, S, 0) as ctx:
        util.load_libs_into(ctx, libs)
        ctx.collapse_seqs(samples[0][0], 0, samples[0][1])
        ctx.annotate()
        got = util.gpu_result_dict(ctx)
        cv = ctx.counts()
        stats = [ctx.round_stats(r + 1)[:2] for r in range(5)]
        handed = [ctx.round_handed_on(r + 1) for r in range(5)]
    res = oracle.run_path(util.oracle_libs(libs, oracle), samples, use_index=True, nthreads=8, maxlen=capi.MAX_READ_LEN)
    assert got == util.oracle_result_dict(res)
    for k in ("quant", "iscan", "cat", "tuniq", "lenhist"):
        assert np.array_equal(cv[k], res[k]), k
    assert stats == [(int(a), int(b)) for a, b in res["stats"]]
    amb = sum(1 for v in got.values() if v[0] > 0 and v[2])
    assert amb > 1000, amb                                           # ties are the rule here, not the exception
    if kb == 4:
        assert sum(handed) > 1000, handed                            # over-long buckets: the hand-over to the generic kernel ran
