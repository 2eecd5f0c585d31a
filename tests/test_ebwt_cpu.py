"""bowtie-index reader (mirge_amd/csrc/ebwt.h) against files written by tests/ebwt_writer.py.
PARITY UNPINNED: both sides follow the same recollection of bowtie 1's on-disk layout; no real index exists here."""
import ctypes as C

import numpy as np
import pytest

from tests.ebwt_writer import write_ebwt


def extract(hostsim, base):
    names = C.create_string_buffer(1 << 20); seqs = C.create_string_buffer(1 << 22)
    offs = np.zeros(1 << 16, dtype=np.uint64)
    n = hostsim.sim_ebwt_extract(str(base).encode(), names, len(names), seqs, len(seqs), offs.ctypes.data, len(offs))
    if n < 0:
        return int(n), names.value.decode()
    nm = names.value.decode().split("\n") if n else []
    return [(nm[i], seqs.raw[int(offs[i]):int(offs[i + 1])].decode()) for i in range(n)], None


def random_refs(rng, n):
    names, seqs = [], []
    for i in range(n):
        L = int(rng.integers(1, 300))
        s = "".join("ACGT"[x] for x in rng.integers(0, 4, size=L))
        if i % 3 == 0 and L > 10:          # ambiguous stretches: leading, inner, trailing
            a, b = sorted(int(x) for x in rng.integers(0, L, size=2))
            s = "N" * int(rng.integers(0, 4)) + s[:a] + "NRY"[i % 3] * (b - a) + s[b:] + "N" * int(rng.integers(0, 3))
        names.append("ref%d some description %d" % (i, L)); seqs.append(s)
    return names, seqs


@pytest.mark.parametrize("seed,big_endian,line_rate,ftab", [(1, False, 6, 4), (2, True, 6, 5), (3, False, 7, 3)])
def test_round_trip(hostsim, tmp_path, seed, big_endian, line_rate, ftab):
    rng = np.random.default_rng(seed)
    names, seqs = random_refs(rng, 200)
    names.append("all_n"); seqs.append("NNNN")                    # skipped by bowtie-build, name included
    want = write_ebwt(str(tmp_path / "lib"), names, seqs, big_endian=big_endian, line_rate=line_rate, ftab_chars=ftab,
                      entire_reverse=(seed == 3))
    got, _ = extract(hostsim, tmp_path / "lib")
    assert got == want


def test_inconsistent_files_fail_loudly(hostsim, tmp_path):
    rng = np.random.default_rng(5)
    names, seqs = random_refs(rng, 20)
    base = str(tmp_path / "lib")
    write_ebwt(base, names, seqs)
    blob = open(base + ".4.ebwt", "rb").read()
    open(base + ".4.ebwt", "wb").write(blob[:-3])
    rc, msg = extract(hostsim, base)
    assert rc == -2 and ".4.ebwt" in msg and ".fa" in msg
    open(base + ".4.ebwt", "wb").write(blob)
    b1 = open(base + ".1.ebwt", "rb").read()
    open(base + ".1.ebwt", "wb").write(b1[:len(b1) // 2])
    rc, msg = extract(hostsim, base)
    assert rc == -2
    open(base + ".1.ebwt", "wb").write(b1[:-1] + b"extra\n\0")    # one name too many
    rc, msg = extract(hostsim, base)
    assert rc == -2 and "names" in msg
    rc, msg = extract(hostsim, str(tmp_path / "missing"))
    assert rc == -1
