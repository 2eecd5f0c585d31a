"""FASTQ record splitting + packing (mirge_amd/csrc/fastq.h) on the HOST build, against a plain Python
restatement of the reference's read loop (miRge.pl:299-307: four lines per record, key = line 2 chomp'ed).
The GPU run of the same cases is tests/test_gpu_fastq.py."""
import ctypes as C

import numpy as np
import pytest


def perl_records(data: bytes):
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()                      # the text after the last "\n" is a line only if it is not empty
    return [lines[i + 1] if i + 1 < len(lines) else b"" for i in range(0, len(lines), 4)]


def decode(key, kw):
    sw = 2 if kw == 2 else 4
    ln = int(key[sw - 1] >> np.uint64(56))
    out = []
    for i in range(ln):
        code = int((int(key[i >> 5]) >> (2 * (i & 31))) & 3)
        is_n = kw == 6 and (int(key[4 + (i >> 6)]) >> (i & 63)) & 1
        out.append("N" if is_n else "ACGT"[code])
    return "".join(out).encode()


def sim_parse(hostsim, data: bytes, win=1 << 31):
    size = len(data)
    pad = np.zeros((size + 15) // 16 * 16 + 16, dtype=np.uint8)
    pad[:size] = np.frombuffer(data, dtype=np.uint8)
    pad[size:] = ord("\n")               # poison: nothing at or beyond `size` may be interpreted
    cap = data.count(b"\n") // 4 + 2
    keys = np.zeros((cap, 6), dtype=np.uint64); kw = np.zeros(cap, dtype=np.int32)
    assert pad.ctypes.data % 16 == 0
    n = hostsim.sim_fastq(pad.ctypes.data, size, win, keys.ctypes.data, kw.ctypes.data, cap)
    if n < 0:
        return n
    # kw 0: a verbatim read (no key holds it) -- the pass reports where its bytes lie
    return [decode(keys[i], int(kw[i])) if kw[i] else data[int(keys[i][0]):int(keys[i][0]) + int(keys[i][1])] for i in range(n)]


def random_fastq(rng, n, maxlen=60, with_n=True, tricky_quals=True):
    out = []
    for i in range(n):
        ln = int(rng.integers(0, maxlen + 1))
        alphabet = b"ACGTN" if with_n and rng.random() < 0.1 else b"ACGT"
        seq = bytes(rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size=ln).tolist())
        hdr = b"@" + bytes(rng.integers(33, 127, size=int(rng.integers(0, 40))).astype(np.uint8).tolist()).replace(b"\n", b"_")
        qual = bytes(rng.integers(33, 127, size=ln).astype(np.uint8).tolist())
        if tricky_quals and ln and rng.random() < 0.3:
            qual = (b"@" if rng.random() < 0.5 else b"+") + qual[1:]          # a quality line may start with '@' or '+'
        out.append(hdr + b"\n" + seq + b"\n+\n" + qual + b"\n")
    return b"".join(out)


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("win", [64, 16384, 1 << 31])
def test_random_files(hostsim, seed, win):
    rng = np.random.default_rng(seed)
    data = random_fastq(rng, 700, maxlen=124 if seed == 3 else 60)
    assert sim_parse(hostsim, data, win) == perl_records(data)


def test_every_truncation_point(hostsim):
    rng = np.random.default_rng(7)
    data = random_fastq(rng, 6, maxlen=30, with_n=False, tricky_quals=False)
    for cut in range(len(data) + 1):
        part = data[:cut]
        got = sim_parse(hostsim, part, 64)
        want = perl_records(part)
        if got == -4:                     # the cut left header/quality bytes on a sequence line?  never: line 2 is always sequence text
            raise AssertionError(cut)
        assert got == want, cut


def test_edge_files(hostsim):
    assert sim_parse(hostsim, b"") == []
    assert sim_parse(hostsim, b"@h") == [b""]
    assert sim_parse(hostsim, b"@h\n") == [b""]
    assert sim_parse(hostsim, b"@h\nACGT") == [b"ACGT"]
    assert sim_parse(hostsim, b"\n\n\n\n") == [b""]
    assert sim_parse(hostsim, b"\n\n\n\n\n") == [b"", b""]
    assert sim_parse(hostsim, b"@h\nAC\n+\nII\n@g\n\n+\n\n") == [b"AC", b""]
    assert sim_parse(hostsim, b"@a\n" + b"A" * 124 + b"\n+\n\n") == [b"A" * 124]
    # reads no key holds are verbatim reads (miRge.pl:306 keys the hash by whatever the line holds)
    assert sim_parse(hostsim, b"@a\n" + b"A" * 125 + b"\n+\n\n") == [b"A" * 125]
    assert sim_parse(hostsim, b"@a\n" + b"A" * 300 + b"\n+\n\n") == [b"A" * 300]
    assert sim_parse(hostsim, b"@a\n" + b"ACGT" * 250) == [b"ACGT" * 250]
    assert sim_parse(hostsim, b"@a\n" + b"A" * 1001 + b"\n+\n\n") == -5
    assert sim_parse(hostsim, b"@a\nacgt\n+\nIIII\n") == [b"acgt"]
    assert sim_parse(hostsim, b"@a\nAC.RT\n+\nIIIII\n@b\nACGT\n+\nIIII\n") == [b"AC.RT", b"ACGT"]
    assert sim_parse(hostsim, b"@a\nACGT\r\n+\nIIII\r\n") == -4          # chomp keeps "\r": refused (control character)
    assert sim_parse(hostsim, b"@a\nAC GT\n+\nIIII\n") == -4
    assert sim_parse(hostsim, b"@a\nAC\tGT\n+\nIIII\n") == -4
    assert sim_parse(hostsim, b"@a\nAC\xc3GT\n+\nIIII\n") == -4
    # 57 nt / N -> wide key; 56 nt ACGT -> narrow key: both decode to the same text
    assert sim_parse(hostsim, b"@a\n" + b"ACGT" * 14 + b"\n+\n\n@b\n" + b"ACGT" * 14 + b"A\n+\n\n@c\nNACGN\n+\n\n") == \
        [b"ACGT" * 14, b"ACGT" * 14 + b"A", b"NACGN"]


def test_newline_mask_exact(hostsim):
    # byte patterns that trip the inexact "haszero" trick (0x0A followed by 0x0B, 0x8A next to 0x0A, ...), and every
    # cut point so the masking of bytes at or beyond `size` is exercised in every word position
    rng = np.random.default_rng(11)
    body = rng.choice(np.array([0x0A, 0x0B, 0x09, 0x8A, 0x01, 0x41, 0x00, 0xFF], dtype=np.uint8), size=1024)
    pad = np.full(1024 + 32, 0x0A, dtype=np.uint8)
    pad[:1024] = body
    raw = body.tobytes()
    for cut in range(0, 1025):
        assert hostsim.sim_count_newlines(pad.ctypes.data, cut) == raw[:cut].count(b"\n"), cut
