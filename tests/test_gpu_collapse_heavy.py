"""Collapse with heavy hitters across two samples (the cross-sample table path): 70 % of the reads are one of seven
sequences, the rest distinct."""
import collections

import numpy as np
import pytest

from mirge_amd import capi
from tests import util

pytestmark = pytest.mark.gpu


def _table(ctx):
    return {k: tuple(int(x) for x in v[3]) for k, v in util.gpu_result_dict(ctx).items()}


def _random_reads(rng, n, lo=16, hi=40):
    lens = rng.integers(lo, hi + 1, size=n).astype(np.uint8)
    seqs = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, 48))]
    seqs[np.arange(48)[None, :] >= lens[:, None]] = 0
    return np.ascontiguousarray(seqs), lens


def test_heavy_hitters_and_two_samples():
    rng = np.random.default_rng(3)
    seqs, lens = _random_reads(rng, 400000)
    hot = rng.integers(0, 7, size=400000)
    pick = rng.random(400000) < 0.7
    seqs[pick] = seqs[hot[pick]]; lens[pick] = lens[hot[pick]]
    want = [collections.Counter(seqs[i, :lens[i]].tobytes().decode() for i in range(a, b)) for a, b in ((0, 250000), (250000, 400000))]
    with capi.Context(None, 2, 0) as ctx:
        ctx.collapse_seqs(seqs[:250000], 0, lens[:250000])
        ctx.collapse_seqs(seqs[250000:], 1, lens[250000:])
        got = _table(ctx)
    assert got == {k: (want[0].get(k, 0), want[1].get(k, 0)) for k in set(want[0]) | set(want[1])}


@pytest.mark.timeout(120)
def test_more_uniques_than_a_25_bit_hash_can_spread():
    """45M reads, ~39M distinct: the claim table takes its slot from 32 hash bits.  (With the 25 bits of `key_hash >> 7`
    a table of more than 2^25 uniques degenerated into probe chains millions of slots long: a 100M-read shard hung.)"""
    import torch
    N = 45_000_000
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    idx = torch.randint(0, 150_000_000, (N,), device="cuda", generator=g, dtype=torch.int64)
    k0 = (idx * 0x9E3779B97F4A7C15) & ((1 << 40) - 1)                 # 20 nt per pool member (odd multiplier: a bijection mod 2^40)
    k16 = torch.stack([k0, torch.full_like(k0, 20 << 56)], dim=1).contiguous()
    k48 = torch.zeros((1, 6), dtype=torch.int64, device="cuda")
    want = int(torch.unique(idx).numel())
    assert want > (1 << 25)
    torch.cuda.synchronize()
    with capi.Context(None, 1, 0) as ctx:
        ctx.collapse_packed_dev(k16.data_ptr(), N, k48.data_ptr(), 0, 0)
        assert ctx.n_unique() == want
        uid, smp, cnt = ctx.export_counts()
        assert len(cnt) == want and int(cnt.sum()) == N


@pytest.mark.parametrize("c1", [None, "300"])
def test_partitioned_collapse_equals_table_form(oracle, monkeypatch, c1):
    """The partitioned collapse (kernels.h: k_part_pass / k_part_fold; taken from 2^20 keys per call) forced onto a small
    multi-sample input, against the table form of the same input: same sequences, same per-sample counts; with c1 set,
    a level-1 region overruns and the call must start over in table form by itself."""
    import numpy as np
    from mirge_amd import synth
    libs, _ = synth.make_libraries(synth.Profile.small(), 77)
    samples = [synth.make_reads(libs, 120000, 780 + s, zipf_perm_seed=3 + s) for s in range(3)]

    def run():
        with capi.Context(None, len(samples), 0) as ctx:
            for s, (seqs, lens) in enumerate(samples):
                ctx.collapse_seqs(seqs, s, lens)
            useqs, ulens, _, _, _ = ctx.export_uniques()
            uid, smp, cnt = ctx.export_counts()
            names = [useqs[u, :ulens[u]].tobytes() for u in range(len(ulens))]
            return {(names[int(u)], int(s)): int(c) for u, s, c in zip(uid, smp, cnt)}, ctx.kernel_times()

    monkeypatch.setenv("MIRGE_PART_MIN", "1000000000")
    want, kt = run()
    assert "collapse_local16" in kt and "collapse_part16" not in kt
    monkeypatch.setenv("MIRGE_PART_MIN", "1")
    if c1:
        monkeypatch.setenv("MIRGE_PART_C1", c1)
    got, kt = run()
    assert ("collapse_local16" in kt) == bool(c1)          # the overrun case ends in table form
    assert got == want
    assert sum(got.values()) == sum(len(l) for _, l in samples)


@pytest.mark.parametrize("c1", [None, "300"])
def test_fused_collapse_annotate_equals_the_two_calls(monkeypatch, c1):
    """mirge_collapse_annotate_packed_dev (the cascade queued behind the collapse, the unique count read on the device)
    against collapse + annotate as two calls: two samples, the partitioned collapse forced onto them; with c1 set its
    regions overrun, which the fused call learns only after the cascade has run -- it must settle the collapse in table
    form and run the cascade again by itself."""
    import numpy as np
    import torch
    from mirge_amd import synth
    from tests import util
    libs, _ = synth.make_libraries(synth.Profile.small(), 91)
    samples = [synth.make_reads(libs, 90000, 910 + s, zipf_perm_seed=5 + s) for s in range(2)]
    monkeypatch.setenv("MIRGE_PART_MIN", "1")

    def run(fused):
        with capi.Context(None, 2, 0) as ctx:
            util.load_libs_into(ctx, libs)
            keep = []
            for s, (seqs, lens) in enumerate(samples):
                n = len(lens)
                d_seqs = torch.from_numpy(seqs).cuda(); d_lens = torch.from_numpy(lens).cuda()
                k16 = torch.empty((n, 2), dtype=torch.int64, device="cuda"); k48 = torch.empty((n, 6), dtype=torch.int64, device="cuda")
                torch.cuda.synchronize()
                n16, n48 = ctx.pack_dev(d_seqs.data_ptr(), seqs.shape[1], d_lens.data_ptr(), n, k16.data_ptr(), k48.data_ptr())
                keep.append((k16, k48))
                if fused:
                    ctx.collapse_annotate_packed_dev(k16.data_ptr(), n16, k48.data_ptr(), n48, s)
                else:
                    ctx.collapse_packed_dev(k16.data_ptr(), n16, k48.data_ptr(), n48, s)
            if not fused:
                ctx.annotate()
            cv = ctx.counts()
            return util.gpu_result_dict(ctx), {k: cv[k].copy() for k in cv}, [ctx.round_stats(r + 1)[:2] for r in range(5)], ctx.kernel_times()

    want, wcv, wst, _ = run(False)
    if c1:
        monkeypatch.setenv("MIRGE_PART_C1", c1)
    got, gcv, gst, kt = run(True)
    assert ("collapse_local16" in kt) == bool(c1)
    assert got == want and gst == wst
    for k in wcv:
        assert np.array_equal(gcv[k], wcv[k]), k


@pytest.mark.parametrize("part_min", ["1", "1000000000"])
def test_fused_call_edge_sizes(monkeypatch, part_min):
    """The fused call on inputs of 0, 1, 2 and 300 reads (empty key streams included), first and second sample, in the
    partitioned and in the table form: the same table and count vectors as the two calls."""
    import numpy as np
    import torch
    from mirge_amd import synth
    from tests import util
    monkeypatch.setenv("MIRGE_PART_MIN", part_min)
    libs, _ = synth.make_libraries(synth.Profile.tiny(), 93)
    seqs, lens = synth.make_reads(libs, 300, 931)
    seqs[7, 3] = ord("N")                                          # one read of the 48-byte class
    d_seqs = torch.from_numpy(seqs).cuda(); d_lens = torch.from_numpy(lens).cuda()
    k16 = torch.empty((300, 2), dtype=torch.int64, device="cuda"); k48 = torch.empty((300, 6), dtype=torch.int64, device="cuda")
    with capi.Context(None, 2, 0) as ctx:
        util.load_libs_into(ctx, libs)
        for n0, n1 in ((0, 0), (1, 0), (0, 2), (300, 1), (9, 300)):
            res = []
            for fused in (False, True):
                ctx.reset()
                for s, n in enumerate((n0, n1)):
                    torch.cuda.synchronize()
                    n16, n48 = ctx.pack_dev(d_seqs.data_ptr(), seqs.shape[1], d_lens.data_ptr(), n, k16.data_ptr(), k48.data_ptr()) if n else (0, 0)
                    a, b = (k16.data_ptr() if n16 else None), (k48.data_ptr() if n48 else None)
                    if fused:
                        ctx.collapse_annotate_packed_dev(a, n16, b, n48, s)
                    else:
                        ctx.collapse_packed_dev(a, n16, b, n48, s)
                if not fused:
                    ctx.annotate()
                cv = ctx.counts()
                res.append((util.gpu_result_dict(ctx), {k: cv[k].copy() for k in cv}, [ctx.round_stats(r + 1)[:2] for r in range(5)]))
            assert res[0][0] == res[1][0] and res[0][2] == res[1][2], (n0, n1)
            for k in res[0][1]:
                assert np.array_equal(res[0][1][k], res[1][1][k]), (n0, n1, k)
            assert sum(sum(v[3]) for v in res[1][0].values()) == n0 + n1
