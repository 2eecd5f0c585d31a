"""Trimming on the GPU (mirge_collapse_raw_fastq*) against the CPU restatement of trim_file.py + cutadapt
(oracle/trim_oracle.c).  PARITY UNPINNED with respect to cutadapt itself: see tests/test_trim_cpu.py."""
import collections

import numpy as np
import pytest

from mirge_amd import capi, synth
from oracle import trim as otrim
from tests.test_trim_cpu import A, raw_case
from tests.util import gpu_result_dict

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(None, 1, 0)
    yield c
    c.close()


def table(ctx):
    return {k: v[3][0] for k, v in gpu_result_dict(ctx).items()}


@pytest.mark.parametrize("seed,adapter,t5,t3,kw", [
    (1, A, 0, 0, {}), (2, A, 1, 2, {}), (3, "illumina", 0, 0, {"read_len": 75}), (4, "none", 0, 0, {}), (5, "ion", 0, 0, {}),
    (6, "+11", 2, 0, {}), (7, "-5", 0, 0, {}), (8, A, 0, 0, {"phred": 64}), (9, A, 0, 0, {"crlf": True}),
    (10, "TGGAATTCTCGG," + A, 0, 0, {}), (11, A, 0, 0, {"adapter_error": 0.9, "low_tail": 0.6}), (12, A[:12], 3, 3, {"read_len": 36}),
    (13, A, 0, 0, {"read_len": 124}),
    # lines no key holds: 150-nt runs with the adapter still on, lower case / IUPAC codes in the reads
    (14, A, 0, 0, {"read_len": 150}), (15, A, 1, 1, {"read_len": 250, "odd": 0.2}), (16, "none", 0, 0, {"read_len": 150, "odd": 0.3}),
    # adapter syntax of cutadapt's -a beyond a plain short sequence: anchored, named, lower case, many, long
    (17, A + "$", 0, 0, {}), (18, "small=" + A.lower() + ",ACGTTGCATGCATGCAAGCTTGACGTACGATCGATCGTACGATCGGTTAACC$", 0, 0, {"read_len": 75}),
    (19, ",".join([A[:20]] + ["ACGTTGCAAGCTTGACCA"[i:] + "GATTACA" for i in range(9)]), 0, 0, {}),
    # IUPAC wildcard characters in the adapter (cutadapt's adapter_wildcards): codes inside it, a random 4-mer (NNNN) in front
    (20, A[:6] + "R" + A[7:14] + "N" + A[15:22] + "B" + A[23:], 0, 0, {}), (21, "NNNN" + A[4:] + ",TGGAATTCTYGG$", 0, 1, {"read_len": 75, "adapter_error": 0.5}),
])
def test_gpu_equals_restatement(ctx, tmp_path, seed, adapter, t5, t3, kw):
    data = raw_case(seed, n=20000, **kw)
    if seed == 18:          # the second, 50-nt adapter is on a share of the reads instead
        data += raw_case(seed + 100, n=5000, adapter="ACGTTGCATGCATGCAAGCTTGACGTACGATCGATCGTACGATCGGTTAACC", read_len=75).replace(b"@raw", b"@two")
    want, total, base, reported = otrim.trim_fastq(data, {"illumina": A, "ion": "+11"}.get(adapter, adapter), t5, t3)
    expect = dict(collections.Counter(x.decode() for x in want))
    ctx.reset()
    assert ctx.collapse_raw_fastq_mem(data, 0, adapter, t5, t3) == (total, len(want), reported)
    assert table(ctx) == expect
    p = tmp_path / "raw.fastq"
    p.write_bytes(data)
    ctx.reset()
    assert ctx.collapse_raw_fastq(str(p), 0, adapter, t5, t3) == (total, len(want), reported)
    assert table(ctx) == expect


def test_raw_equals_trimmed_path(ctx):
    """Raw reads through the trimming step give the table the trimmed reads give through mirge_collapse_fastq_mem."""
    data = raw_case(21, n=30000)
    want, total, _, _ = otrim.trim_fastq(data, A)
    trimmed = b"".join(b"@t\n" + s + b"\n+\n" + b"I" * len(s) + b"\n" for s in want)
    ctx.reset(); ctx.collapse_fastq_mem(trimmed, 0); a = table(ctx)
    ctx.reset(); ctx.collapse_raw_fastq_mem(data, 0, "illumina"); b = table(ctx)
    assert a == b and len(a) > 100


def test_errors(ctx):
    for data, code in [(b"@r\nACGTACGTACGTACGTACGT\n+\nIIII\n", capi.MIRGE_E_FORMAT),
                       (b"@r\n" + b"A" * 1001 + b"\n+\n" + b"I" * 1001 + b"\n", capi.MIRGE_E_LIMIT),
                       (b"@r\nACGTACGTACGTACGTA GT\n+\n" + b"I" * 20 + b"\n", capi.MIRGE_E_FORMAT)]:
        ctx.reset()
        with pytest.raises(capi.MirgeError) as e:
            ctx.collapse_raw_fastq_mem(data, 0, "none")
        assert e.value.code == code
    for bad in ("ACGJ", "ACGT{3}", "ACGTAC...GGTTAA", "A" * 65, ",".join(["ACGTAC"] * 17)):
        with pytest.raises(capi.MirgeError):
            ctx.collapse_raw_fastq_mem(b"", 0, bad)
    ctx.reset()
    assert ctx.collapse_raw_fastq_mem(b"", 0, "illumina") == (0, 0, 33)
