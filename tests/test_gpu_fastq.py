"""FASTQ ingest on the GPU (mirge_collapse_fastq / _mem / _dev) against the reference's read loop restated in
Python (miRge.pl:299-307).  Same cases as tests/test_fastq_cpu.py, through the C ABI."""
import collections
import os

import numpy as np
import pytest

from mirge_amd import capi
from tests.test_fastq_cpu import perl_records, random_fastq
from tests.util import gpu_result_dict

pytestmark = pytest.mark.gpu


def table(ctx):
    return {k: v[3][0] for k, v in gpu_result_dict(ctx).items()}


def want(data):
    return dict(collections.Counter(r.decode() for r in perl_records(data)))


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(None, 1, 0)
    yield c
    c.close()


@pytest.mark.parametrize("seed,n,maxlen", [(1, 700, 60), (2, 5000, 124), (3, 200000, 47)])
def test_mem_file_dev_agree_with_perl_loop(ctx, tmp_path, seed, n, maxlen):
    import torch
    rng = np.random.default_rng(seed)
    if n < 10000:
        data = random_fastq(rng, n, maxlen=maxlen)
    else:                                  # many duplicates, several 16 KB segments, ~8 MB
        pool = [bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(l)).tolist()) for l in rng.integers(16, maxlen + 1, size=3000)]
        data = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, pool[j], b"I" * len(pool[j])) for i, j in enumerate(rng.integers(0, len(pool), size=n)))
    expect = want(data)
    ctx.reset()
    assert ctx.collapse_fastq_mem(data, 0) == len(perl_records(data))
    assert table(ctx) == expect
    p = tmp_path / "x.fastq"
    p.write_bytes(data)
    ctx.reset()
    assert ctx.collapse_fastq(str(p), 0) == len(perl_records(data))
    assert table(ctx) == expect
    d = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    # poison what follows the text inside the allocation: nothing at or beyond `size` may be interpreted
    buf = torch.full((len(data) + 4096,), 10, dtype=torch.uint8, device="cuda")
    buf[:len(data)] = d
    torch.cuda.synchronize()
    ctx.reset()
    assert ctx.collapse_fastq_dev(buf.data_ptr(), len(data), 0) == len(perl_records(data))
    assert table(ctx) == expect


def test_edge_files(ctx):
    cases = [b"", b"@h", b"@h\n", b"@h\nACGT", b"\n\n\n\n", b"\n\n\n\n\n", b"@h\nAC\n+\nII\n@g\n\n+\n\n", b"@a\nAC\n+\nII\n0",
             b"@a\n" + b"A" * 124 + b"\n+\n\n", b"@a\n" + b"ACGT" * 14 + b"\n+\n\n@b\n" + b"ACGT" * 14 + b"A\n+\n\n@c\nNACGN\n+\n\n"]
    for data in cases:
        ctx.reset()
        assert ctx.collapse_fastq_mem(data, 0) == len(perl_records(data)), data
        assert table(ctx) == want(data), data


def test_every_truncation_point(ctx):
    rng = np.random.default_rng(7)
    data = random_fastq(rng, 5, maxlen=30, with_n=False, tricky_quals=False)
    for cut in range(len(data) + 1):
        ctx.reset()
        ctx.collapse_fastq_mem(data[:cut], 0)
        assert table(ctx) == want(data[:cut]), cut


def test_errors(ctx, tmp_path):
    for data, code in [(b"@a\n" + b"A" * 1001 + b"\n+\n\n", capi.MIRGE_E_LIMIT), (b"@a\n" + b"A" * 3000 + b"\n+\n\n", capi.MIRGE_E_LIMIT),
                       (b"@a\nACGT\r\n+\nIIII\r\n", capi.MIRGE_E_FORMAT), (b"@a\nac gt\n+\nIIIII\n", capi.MIRGE_E_FORMAT)]:
        ctx.reset()
        with pytest.raises(capi.MirgeError) as e:
            ctx.collapse_fastq_mem(data, 0)
        assert e.value.code == code
    with pytest.raises(capi.MirgeError) as e:
        ctx.collapse_fastq(str(tmp_path / "missing.fastq"), 0)
    assert e.value.code == capi.MIRGE_E_IO


def test_two_samples_from_files(ctx, tmp_path):
    rng = np.random.default_rng(5)
    a, b = random_fastq(rng, 3000, maxlen=30), random_fastq(rng, 2000, maxlen=30)
    (tmp_path / "a.fq").write_bytes(a); (tmp_path / "b.fq").write_bytes(b)
    ctx.reset(2)
    ctx.collapse_fastq(str(tmp_path / "a.fq"), 0)
    ctx.collapse_fastq(str(tmp_path / "b.fq"), 1)
    got = {k: v[3] for k, v in gpu_result_dict(ctx).items()}
    wa, wb = want(a), want(b)
    assert got == {k: (wa.get(k, 0), wb.get(k, 0)) for k in set(wa) | set(wb)}
    ctx.reset(1)


def test_multi_window_path(ctx, monkeypatch):
    """Files larger than one 2 GiB window are counted / indexed window by window with the newline ordinal carried
    over; MIRGE_FQ_WINDOW shrinks the window so a 1 MB file crosses many of them."""
    rng = np.random.default_rng(9)
    data = random_fastq(rng, 12000, maxlen=60)
    assert len(data) > 20 * 32768
    monkeypatch.setenv("MIRGE_FQ_WINDOW", "32768")
    ctx.reset()
    assert ctx.collapse_fastq_mem(data, 0) == len(perl_records(data))
    assert table(ctx) == want(data)


def test_gzip_files(ctx, tmp_path):
    """fastq.gz (miRge.pl:1092; cutadapt's reader inflates inside trim_file.py:123): single member, several members
    back to back (the trailer then understates the size: the device buffer grows), a file larger than the 16 MB ring
    slots, empty content, and a truncated stream (loud failure)."""
    import gzip
    rng = np.random.default_rng(11)
    small = random_fastq(rng, 4000, maxlen=60)
    pool = [bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(l)).tolist()) for l in rng.integers(16, 48, size=2000)]
    big = b"".join(b"@read%d\n%s\n+\n%s\n" % (i, pool[j], b"I" * len(pool[j])) for i, j in enumerate(rng.integers(0, len(pool), size=450000)))
    assert len(big) > (32 << 20)          # more than two 16 MB ring slots
    cases = {"one.fastq.gz": (gzip.compress(small, 6), small),
             "multi.fastq.gz": (gzip.compress(small[:1000], 1) + gzip.compress(small[1000:], 9) + gzip.compress(b""), small),
             "big.fastq.gz": (gzip.compress(big, 1), big),
             "empty.fastq.gz": (gzip.compress(b""), b"")}
    for name, (blob, data) in cases.items():
        p = tmp_path / name
        p.write_bytes(blob)
        ctx.reset()
        assert ctx.collapse_fastq(str(p), 0) == len(perl_records(data)), name
        assert table(ctx) == want(data), name
    p = tmp_path / "cut.fastq.gz"
    p.write_bytes(cases["one.fastq.gz"][0][:-200])
    ctx.reset()
    with pytest.raises(capi.MirgeError) as e:
        ctx.collapse_fastq(str(p), 0)
    assert e.value.code == capi.MIRGE_E_IO
    # raw reads (trimming path) from a gzip file give what the plain file gives
    from mirge_amd import synth
    libs, _ = synth.make_libraries(synth.Profile.tiny(), synth.BASE_SEED + 3)
    seqs, lens = synth.make_reads(libs, 20000, synth.BASE_SEED + 5)
    plain = tmp_path / "raw.fastq"
    synth.write_raw_fastq_fast(str(plain), seqs, lens, synth.BASE_SEED + 6)
    (tmp_path / "raw.fastq.gz").write_bytes(gzip.compress(plain.read_bytes(), 1))
    ctx.reset()
    a = ctx.collapse_raw_fastq(str(plain), 0, "illumina"); ta = table(ctx)
    ctx.reset()
    b = ctx.collapse_raw_fastq(str(tmp_path / "raw.fastq.gz"), 0, "illumina")
    assert tuple(a) == tuple(b) and table(ctx) == ta and a[0] == 20000
