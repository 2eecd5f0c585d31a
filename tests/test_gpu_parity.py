"""GPU parity tests: the HIP path, called through the C ABI, against (a) the golden tables the
unmodified reference miRge.pl wrote and (b) the CPU oracle on seeded inputs.  Integer work:
everything is compared bit-exact."""
import numpy as np
import pytest

from mirge_amd import capi, synth
from mirge_amd.driver import MirgeRun
from tests import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", util.golden_cases())
def test_golden_tables(tmp_path, name):
    """FASTQ -> collapse -> cascade -> count vectors on the device, merge/filter/tables on the host:
    miR.Counts.csv and miR.RPM.csv byte-identical, mapped/unmapped.csv identical as row sets, and the
    report's quantStats / annotStats numbers equal to the reference run's.  (The reference ran with the oracle's
    bowtie stand-in as its bowtie: parity with bowtie 1.1.1's own arithmetic -- tie order H1, read-N cost, characters --
    is unpinned)"""
    case = util.golden_case(name)
    run = MirgeRun(case["species"], case["fastqs"], case["names"])
    out = str(tmp_path / "out")
    run.run(out)
    util.check_run_against_golden(run, case, out)
    run.ctx.close()


def _compare_with_oracle(oracle, libs, samples, use_index=True, nthreads=8):
    S = len(samples)
    with capi.Context(None, S, 0) as ctx:
        util.load_libs_into(ctx, libs)
        for s, (seqs, lens) in enumerate(samples):
            ctx.collapse_seqs(seqs, s, lens)
        ctx.annotate()
        got = util.gpu_result_dict(ctx)
        cv = ctx.counts()
        stats = [ctx.round_stats(r + 1)[:2] for r in range(5)]
    res = oracle.run_path(util.oracle_libs(libs, oracle), samples, use_index=use_index, nthreads=nthreads,
                          maxlen=capi.MAX_READ_LEN)
    exp = util.oracle_result_dict(res)
    assert len(got) == len(exp)
    assert got == exp
    for k in ("quant", "iscan", "cat", "tuniq", "lenhist"):
        assert np.array_equal(cv[k], res[k]), k
    assert stats == [(int(a), int(b)) for a, b in res["stats"]]
    return res


def test_tiny_vs_oracle_bruteforce(oracle):
    libs, _ = synth.make_libraries(synth.Profile.tiny(), 101)
    samples = [synth.make_reads(libs, 3000, 200 + s) for s in range(3)]
    res = _compare_with_oracle(oracle, libs, samples, use_index=False)
    assert (res["ann_round"] > 0).sum() > 500


def test_small_multisample_vs_oracle(oracle):
    libs, _ = synth.make_libraries(synth.Profile.small(), 102)
    samples = [synth.make_reads(libs, 40000, 300 + s, zipf_perm_seed=900 + s) for s in range(4)]
    res = _compare_with_oracle(oracle, libs, samples)
    for r in range(1, 6):
        assert (res["ann_round"] == r).sum() > 0, r


def test_isomir_rich_vs_oracle(oracle):
    """BASELINE.json configs[3]: isomiR-rich reads (5'/3' variants, 2 mismatches)."""
    libs, _ = synth.make_libraries(synth.Profile.small(), 103)
    samples = [synth.make_reads(libs, 60000, 400, synth.Mixture.isomir_rich())]
    res = _compare_with_oracle(oracle, libs, samples)
    assert (res["ann_round"] == 5).sum() > 1000


def test_medium_vs_oracle_indexed(oracle):
    libs, _ = synth.make_libraries(synth.Profile.medium(), 104)
    samples = [synth.make_reads(libs, 300000, 500)]
    _compare_with_oracle(oracle, libs, samples)


def test_edge_cases(oracle):
    """Empty sample, empty read, every length 1..124, reads with N, low-complexity reads, duplicates
    across samples, a read present only in a later sample."""
    rng = np.random.default_rng(7)
    libs, _ = synth.make_libraries(synth.Profile.tiny(), 105)
    mir = libs["mirna"]; mr = libs["mrna"]
    rs = lambda n: "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))
    s0 = [rs(L) for L in range(1, 125)] + [""] + ["A" * 30, "T" * 20, "ACACACACACACACACACAC"]
    s0 += [mir.seq(0), mir.seq(0), mir.seq(1)[:20], "N" * 20, "ACGTN" * 5]
    long_m = mr.seq(0)
    s0 += [long_m[10:10 + L] for L in (56, 57, 64, 65, 100, 124)]
    s0 += [long_m[200:260][:30] + "N" + long_m[231:260]]
    s1 = []
    s2 = [mir.seq(0), rs(40), long_m[10:67], mir.seq(2)] * 3
    as_arrays = lambda L: L
    S = 3
    with capi.Context(None, S, 0) as ctx:
        util.load_libs_into(ctx, libs)
        for s, reads in enumerate([s0, s1, s2]):
            ctx.collapse_seqs(as_arrays(reads), s)
        ctx.annotate()
        got = util.gpu_result_dict(ctx)
        cv = ctx.counts()
        # the single-unique accessor agrees with the bulk export
        for i in range(0, ctx.n_unique(), 7):
            seq, refs, amb, cnt = ctx.unique(i)
            r, ref, a, c = got[seq]
            assert tuple(cnt) == c
            assert [x for x in refs if x >= 0] == ([ref] if r else [])
    res = oracle.run_path(util.oracle_libs(libs, oracle), [s0, s1, s2], use_index=False, maxlen=capi.MAX_READ_LEN)
    assert got == util.oracle_result_dict(res)
    for k in ("quant", "iscan", "cat", "tuniq", "lenhist"):
        assert np.array_equal(cv[k], res[k]), k
    assert "" in got and got[""][0] == 0


def test_errors_fail_loudly():
    libs, _ = synth.make_libraries(synth.Profile.tiny(), 106)
    with capi.Context(None, 1, 0) as ctx:
        util.load_libs_into(ctx, libs)
        with pytest.raises(capi.MirgeError) as e:
            ctx.collapse_seqs(["ACGT", "AC GU"], 0)
        assert e.value.code == -4
        with pytest.raises(capi.MirgeError) as e:
            ctx.collapse_seqs(["A" * (capi.MAX_READ_LEN + 1)], 0)
        assert e.value.code == -5
        with pytest.raises(capi.MirgeError):
            ctx.collapse_seqs(["ACGT"], 3)
        with pytest.raises(capi.MirgeError):
            ctx.counts()
    with capi.Context(None, 1, 0) as ctx:
        ctx.collapse_seqs(["ACGT"], 0)
        with pytest.raises(capi.MirgeError):
            ctx.annotate()                      # libraries not loaded
    with pytest.raises(capi.MirgeError):
        capi.Context("/nonexistent/species", 1, 0)


def test_validate_keys_names_the_malformed_key():
    """mirge_validate_keys_dev: keys a caller built itself, against the formats of keys.h."""
    import torch
    def key16(seq, length=None):
        w = [0, 0]
        for i, ch in enumerate(seq):
            w[i >> 5] |= "ACGT".index(ch) << (2 * (i & 31))
        w[1] |= (len(seq) if length is None else length) << 56
        return [x - (1 << 64) if x >= (1 << 63) else x for x in w]
    good = [key16("ACGTACGTACGTACGTAC"), key16("T" * 56), key16("")]
    with capi.Context(None, 1, 0) as ctx:
        k = torch.tensor(good, dtype=torch.int64, device="cuda")
        ctx.validate_keys_dev(k.data_ptr(), len(good), None, 0)
        for bad, what in ((key16("ACGT", 57), "length"), (key16("ACGTACGT", 4), "bits above")):
            k = torch.tensor(good + [bad], dtype=torch.int64, device="cuda")
            with pytest.raises(capi.MirgeError) as e:
                ctx.validate_keys_dev(k.data_ptr(), len(good) + 1, None, 0)
            assert e.value.code == capi.MIRGE_E_FORMAT and "key 3" in str(e.value), what
        # 48-byte keys: N mask beyond the length, and a non-zero code under an N
        base = [0, 0, 0, 5 << 56, 0, 0]                                               # "AAAAA"
        for bad in ([0, 0, 0, 5 << 56, 1 << 7, 0], [3 << 2, 0, 0, 5 << 56, 1 << 1, 0]):
            k = torch.tensor([base, bad], dtype=torch.int64, device="cuda")
            with pytest.raises(capi.MirgeError) as e:
                ctx.validate_keys_dev(None, 0, k.data_ptr(), 2)
            assert e.value.code == capi.MIRGE_E_FORMAT and "key 1" in str(e.value)
        ok = torch.tensor([base, [0, 0, 0, 5 << 56, 1 << 1, 0]], dtype=torch.int64, device="cuda")
        ctx.validate_keys_dev(None, 0, ok.data_ptr(), 2)


def test_device_resident_path_and_reset(oracle):
    """The configuration bench.py times: ASCII matrix resident in HBM -> mirge_pack_dev ->
    mirge_collapse_packed_dev -> annotate, twice through mirge_reset (second run smaller)."""
    import torch
    libs, _ = synth.make_libraries(synth.Profile.small(), 107)
    olibs = util.oracle_libs(libs, oracle)
    with capi.Context(None, 1, 0) as ctx:
        util.load_libs_into(ctx, libs)
        for n, seed in ((50000, 600), (12000, 601)):
            seqs, lens = synth.make_reads(libs, n, seed)
            d_seqs = torch.from_numpy(seqs).cuda(); d_lens = torch.from_numpy(lens).cuda()
            k16 = torch.empty((n, 2), dtype=torch.int64, device="cuda"); k48 = torch.empty((n, 6), dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            n16, n48 = ctx.pack_dev(d_seqs.data_ptr(), seqs.shape[1], d_lens.data_ptr(), n, k16.data_ptr(), k48.data_ptr())
            assert n16 + n48 == n and n48 == int(((seqs == ord("N")).any(axis=1)).sum())
            ctx.validate_keys_dev(k16.data_ptr(), n16, k48.data_ptr(), n48)          # what the library packs passes its own check
            ctx.reset()
            ctx.collapse_packed_dev(k16.data_ptr(), n16, k48.data_ptr(), n48, 0)
            ctx.annotate()
            got = util.gpu_result_dict(ctx)
            cv = ctx.counts()
            res = oracle.run_path(olibs, [(seqs, lens)], use_index=True, nthreads=8, maxlen=capi.MAX_READ_LEN)
            assert got == util.oracle_result_dict(res)
            for k in ("quant", "iscan", "cat", "tuniq", "lenhist"):
                assert np.array_equal(cv[k], res[k]), k
            # same answer from the fused call (the cascade queued behind the collapse, unique count left on the device)
            ctx.reset()
            ctx.collapse_annotate_packed_dev(k16.data_ptr(), n16, k48.data_ptr(), n48, 0)
            assert util.gpu_result_dict(ctx) == got
            cv2 = ctx.counts()
            for k in ("quant", "iscan", "cat", "tuniq", "lenhist"):
                assert np.array_equal(cv2[k], res[k]), k
            assert [ctx.round_stats(r + 1)[:2] for r in range(5)] == [tuple(int(x) for x in res["stats"][r]) for r in range(5)]
            # same answer from the ASCII-resident entry point
            ctx.reset()
            ctx.collapse_ascii_dev(d_seqs.data_ptr(), seqs.shape[1], d_lens.data_ptr(), n, 0)
            ctx.annotate()
            assert util.gpu_result_dict(ctx) == got


def test_partition_invariance(oracle):
    """Shards of one read set (as bench.py --gpus N deals them) sum to the unsharded count vectors."""
    libs, _ = synth.make_libraries(synth.Profile.small(), 108)
    seqs, lens = synth.make_reads(libs, 80000, 700)
    flats = []
    for nshard in (1, 2, 4):
        tot = None
        for r in range(nshard):
            with capi.Context(None, 1, 0) as ctx:
                util.load_libs_into(ctx, libs)
                ctx.collapse_seqs(seqs[r::nshard], 0, lens[r::nshard])
                ctx.annotate()
                f = ctx.counts_flat()
            # trimmedUniq and the per-round processed / aligned counts (the tail of the buffer) count UNIQUES per shard:
            # not additive without the unique-table merge, left out here
            tail = 1 + 2 * capi.MAX_ROUNDS
            tot = f[:-tail].copy() if tot is None else tot + f[:-tail]
        flats.append(tot)
    assert np.array_equal(flats[0], flats[1]) and np.array_equal(flats[0], flats[2])


def test_policy_cases_on_gpu():
    """The by-construction known answers of tests/test_policy_known_answers.py, on the device."""
    from tests.test_policy_known_answers import build_cases
    lib, cases = build_cases()
    names = ["ref%d" % i for i in range(len(lib))]
    with capi.Context(None, 1, 0) as ctx:
        for l in range(4):
            ctx.load_library(l, names, seqs=lib)
        for rnd in sorted({c[2] for c in cases}):
            sub = [c for c in cases if c[2] == rnd]
            ctx.set_rounds([(0,) + tuple(rnd[1:])])
            ctx.reset()
            ctx.collapse_seqs([c[1] for c in sub], 0)
            ctx.annotate()
            got = util.gpu_result_dict(ctx)
            for what, read, _, ref in sub:
                assert got[read][1] == ref, what


def test_48_sample_batch(oracle):
    """BASELINE.json configs[4] in small: 48 samples share one unique table, per-sample count vectors."""
    libs, _ = synth.make_libraries(synth.Profile.small(), 109)
    samples = [synth.make_reads(libs, 3000, 800 + s, zipf_perm_seed=50 + s % 5) for s in range(48)]
    res = _compare_with_oracle(oracle, libs, samples)
    assert res["counts"].shape[1] == 48 and (res["counts"] > 0).sum(axis=1).max() >= 24


@pytest.mark.parametrize("world", [2, 3])
def test_unique_table_merge_across_logical_shards(world):
    """SURVEY.md 8e acceptance: identical results for 1 and N shards, including the numbers that are sums over
    uniques (trimmed_uniq, per-round processed/aligned).  N logical ranks on one device; the all-to-all of
    mirge_amd/shard.py is done by hand (slices of the export buffers), the kernels are the real ones."""
    import torch
    from mirge_amd import shard
    libs, _ = synth.make_libraries(synth.Profile.small(), 107)
    S = 2
    samples = [synth.make_reads(libs, 30000, 700 + s, zipf_perm_seed=70 + s) for s in range(S)]
    # a few reads with N / longer than 56 nt so the 48-byte class travels too
    rng = np.random.default_rng(3)
    for seqs, lens in samples:
        for i in rng.integers(0, len(lens), size=200):
            seqs[i, int(rng.integers(0, lens[i]))] = ord("N")
    with capi.Context(None, S, 0) as one:
        util.load_libs_into(one, libs)
        for s, (seqs, lens) in enumerate(samples):
            one.collapse_seqs(seqs, s, lens)
        one.annotate()
        want = util.gpu_result_dict(one)
        want_flat = one.counts_flat().copy()
        want_stats = [one.round_stats(r + 1)[:2] for r in range(5)]
    ctxs = [capi.Context(None, S, 0) for _ in range(world)]
    try:
        sends = []
        for r, c in enumerate(ctxs):
            util.load_libs_into(c, libs)
            for s, (seqs, lens) in enumerate(samples):
                lo, hi = shard.shard_bounds(len(lens), r, world)
                c.collapse_seqs(seqs[lo:hi], s, lens[lo:hi])
            n16, n48 = c.shard_export_sizes(world)
            b16 = torch.empty(max(int(n16.sum()) * shard.REC16, 8), dtype=torch.uint8, device="cuda")
            b48 = torch.empty(max(int(n48.sum()) * shard.REC48, 8), dtype=torch.uint8, device="cuda")
            c.shard_export(world, b16.data_ptr(), b48.data_ptr())
            o16 = np.concatenate([[0], np.cumsum(n16)]).astype(np.int64) * shard.REC16
            o48 = np.concatenate([[0], np.cumsum(n48)]).astype(np.int64) * shard.REC48
            sends.append((b16, o16, b48, o48))
            assert int(n48.sum()) > 0
        got, flat, stats = {}, np.zeros_like(want_flat), np.zeros((5, 2), dtype=np.int64)
        for d, c in enumerate(ctxs):
            r16 = torch.cat([b16[o16[d]:o16[d + 1]] for b16, o16, _, _ in sends])
            r48 = torch.cat([b48[o48[d]:o48[d + 1]] for _, _, b48, o48 in sends])
            torch.cuda.synchronize()
            c.shard_import(r16.data_ptr() if r16.numel() else 0, r16.numel() // shard.REC16,
                           r48.data_ptr() if r48.numel() else 0, r48.numel() // shard.REC48)
            c.annotate()
            part = util.gpu_result_dict(c)
            assert not (set(part) & set(got))                 # every sequence lives on exactly one rank
            got.update(part)
            flat += c.counts_flat()
            stats += np.array([c.round_stats(r + 1)[:2] for r in range(5)], dtype=np.int64)
        assert got == want
        assert np.array_equal(flat, want_flat)                # trimmed_uniq included
        assert [tuple(int(x) for x in row) for row in stats] == [tuple(int(x) for x in row) for row in want_stats]
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("name", util.golden_cases())
def test_golden_tables_from_ebwt_only_species_dir(tmp_path, name):
    """A seqLibs directory that holds only bowtie indexes (no FASTA): the libraries come out of <lib>.{1,3,4}.ebwt
    (mirge_amd/csrc/ebwt.h; layout unvalidated against a real bowtie-build, written here by tests/ebwt_writer.py) and
    the run reproduces the reference's tables as the FASTA-backed run does."""
    import os
    import shutil
    from tests.ebwt_writer import write_ebwt
    case = util.golden_case(name)
    sp = tmp_path / "species"
    sp.mkdir()
    for lib in ("mirna", "hairpin", "other", "mrna"):
        names, seqs = [], []
        for line in open(os.path.join(case["species"], lib + ".fa")).read().split("\n"):
            if line.startswith(">"):
                names.append(line[1:]); seqs.append("")
            elif names:
                seqs[-1] += line.strip()
        write_ebwt(str(sp / lib), names, seqs)
    if os.path.exists(os.path.join(case["species"], "merges.csv")):
        shutil.copy(os.path.join(case["species"], "merges.csv"), sp / "merges.csv")
    run = MirgeRun(str(sp), case["fastqs"], case["names"])
    out = str(tmp_path / "out")
    run.run(out)
    util.check_run_against_golden(run, case, out)
    run.ctx.close()


@pytest.mark.parametrize("name", util.golden_cases())
def test_native_tables_equal_reference(tmp_path, name):
    """mirge_write_tables (summarize bookkeeping + miRNAmerge + filter + writeDataToCSV in native code) against the
    tables the unmodified reference wrote: Counts/RPM byte-identical, mapped/unmapped as row sets (bowtie's own
    arithmetic aside: parity unpinned)."""
    import os
    case = util.golden_case(name)
    S = len(case["fastqs"])
    with capi.Context(case["species"], S, 0) as ctx:
        for i, f in enumerate(case["fastqs"]):
            ctx.collapse_fastq(f, i)
        ctx.annotate()
        out = str(tmp_path / "out")
        tot, uq = ctx.write_tables(os.path.join(case["species"], "merges.csv"), out, case["names"])
    exp = case["expected"]
    for f in ("miR.Counts.csv", "miR.RPM.csv"):
        assert open(os.path.join(out, f)).read() == open(os.path.join(exp, f)).read(), f
    for f in ("mapped.csv", "unmapped.csv"):
        assert util.rows(os.path.join(out, f)) == util.rows(os.path.join(exp, f)), f
    tline = open(os.path.join(exp, "miR.Counts.csv")).read().split("\n")[1].split(",")[1:]
    assert [str(int(t)) if u else "" for t, u in zip(tot, uq)] == tline


def test_large_mirna_library_counts_through_global_atomics(oracle):
    """More miRNA names than the count kernel keeps in LDS (CNT_LDS_M = 3584): quant / iscan then go straight to
    global atomics."""
    prof = synth.Profile(n_mirna=4200, n_snp=10, n_family=6, n_hairpin=300, n_other=400, other_median=150.0,
                         n_mrna=200, mrna_median=1500.0, mrna_cap=8000)
    libs, _ = synth.make_libraries(prof, 909)
    assert libs["mirna"].n > 3584
    samples = [synth.make_reads(libs, 60000, 910 + s) for s in range(2)]
    _compare_with_oracle(oracle, libs, samples)


def test_counts_with_lds_columns_on_a_small_input(oracle, monkeypatch):
    """The count kernel keeps the quant / iscan columns in LDS only from 2^18 entries up (below that the 2 x M x 8 B of
    dynamic LDS per block cost more than they save): MIRGE_CNT_LDS_MIN = 0 puts that form on a test-sized input."""
    monkeypatch.setenv("MIRGE_CNT_LDS_MIN", "0")
    libs, _ = synth.make_libraries(synth.Profile.small(), 921)
    samples = [synth.make_reads(libs, 40000, 922 + s) for s in range(3)]
    _compare_with_oracle(oracle, libs, samples)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_cascades_vs_oracle(oracle, seed):
    """Cascades other than miRge's five rounds (the list of rounds is data): random libraries per round,
    length filters, -5/-3 trimming, seed lengths 5..30, 0..3 seed mismatches, 0..3 mismatches overall (1..4 pieces,
    cut from the seed or from the whole read), up to 8 rounds -- the device cascade against the oracle's."""
    rng = np.random.default_rng(1000 + seed)
    libs, _ = synth.make_libraries(synth.Profile.small(), 300 + seed)
    samples = [synth.make_reads(libs, 20000, 310 + seed)]
    seqs, lens = samples[0]
    for i in rng.integers(0, len(lens), size=300):                       # some reads with N: the 48-byte class
        seqs[i, int(rng.integers(0, lens[i]))] = ord("N")
    rounds = []
    for _ in range(int(rng.integers(2, 9))):
        len_lt = int(rng.choice([0, 0, 22, 26, 30])); len_gt = int(rng.choice([0, 0, 18, 25]))
        rounds.append((int(rng.integers(0, 4)), len_lt, len_gt, int(rng.integers(0, 4)), int(rng.integers(0, 4)),
                       int(rng.choice([5, 8, 12, 15, 20, 28, 30])), int(rng.integers(0, 4)), int(rng.integers(0, 4))))
    S = 1
    with capi.Context(None, S, 0) as ctx:
        util.load_libs_into(ctx, libs)
        ctx.set_rounds(rounds)
        ctx.collapse_seqs(seqs, 0, lens)
        ctx.annotate()
        got = util.gpu_result_dict(ctx)
        cv = ctx.counts()
        stats = [ctx.round_stats(r + 1)[:2] for r in range(len(rounds))]
    res = oracle.run_path(util.oracle_libs(libs, oracle), samples, rounds=rounds, use_index=True, nthreads=8, maxlen=capi.MAX_READ_LEN)
    assert got == util.oracle_result_dict(res), rounds
    for k in ("quant", "iscan", "cat", "tuniq", "lenhist"):       # category = library of the annotating round, quant only from miRNA rounds
        assert np.array_equal(cv[k], res[k]), (k, rounds)
    assert stats == [(int(a), int(b)) for a, b in res["stats"]][:len(rounds)], rounds
    assert sum(b for _, b in stats) > 1000


def test_flat_count_buffer_on_the_device(oracle):
    """mirge_counts_flat_get_dev / _set_dev (the all-reduce buffer left in device memory) against the host forms."""
    import torch
    libs, _ = synth.make_libraries(synth.Profile.tiny(), 131)
    seqs, lens = synth.make_reads(libs, 5000, 132)
    with capi.Context(None, 2, 0) as ctx:
        util.load_libs_into(ctx, libs)
        ctx.collapse_seqs(seqs[:3000], 0, lens[:3000]); ctx.collapse_seqs(seqs[3000:], 1, lens[3000:])
        ctx.annotate()
        host = ctx.counts_flat()
        t = torch.full((ctx.counts_flat_len(),), -1, dtype=torch.int64, device="cuda")
        ctx.counts_flat_get_dev(t.data_ptr())
        assert np.array_equal(t.cpu().numpy().view(np.uint64), host)
        assert host[-2 * capi.MAX_ROUNDS:].sum() > 0                    # the round statistics are in it
        stats = [ctx.round_stats(r + 1)[:2] for r in range(5)]
        t *= 3
        ctx.counts_flat_set_dev(t.data_ptr())
        assert np.array_equal(ctx.counts_flat(), host * 3)
        assert [ctx.round_stats(r + 1)[:2] for r in range(5)] == [(3 * a, 3 * b) for a, b in stats]
