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
