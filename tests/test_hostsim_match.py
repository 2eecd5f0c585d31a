"""CPU check of the matcher's bit arithmetic (keys.h + pack_chunk/verify/match_one of kernels.h,
compiled for the HOST by tests/hostsim) against the oracle's exhaustive ASCII matcher.  This is
what lets a kernel logic error show up here instead of on the GPU box; the GPU parity tests
(test_gpu_*.py) call the real kernels through the C ABI."""
import ctypes as C

import numpy as np
import pytest

from oracle.oracle import MIRGE_ROUNDS, OracleLib


def _rand_seq(rng, n):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))


def _mutate(rng, s, k, alphabet="ACGT"):
    s = list(s)
    for p in rng.choice(len(s), size=min(k, len(s)), replace=False):
        s[p] = alphabet[(alphabet.index(s[p]) + int(rng.integers(1, 4))) % 4] if s[p] in alphabet else "A"
    return "".join(s)


def _rc(s):
    return s[::-1].translate(str.maketrans("ACGTN", "TGCAN"))


class Sim:
    def __init__(self, L, seqs, kb, flags=0):
        self.L = L
        text = "".join(seqs).encode()
        offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
        np.cumsum([len(s) for s in seqs], out=offs[1:])
        self.h = L.sim_lib_new(len(seqs), text, offs.ctypes.data, kb, flags)

    def match(self, read, rnd):
        key = np.zeros(6, dtype=np.uint64)
        kw = self.L.sim_pack(read.encode(), len(read), key.ctypes.data)
        assert kw in (2, 6), kw
        r = np.array(rnd, dtype=np.int32)
        ref = C.c_int32(); amb = C.c_int32()
        self.L.sim_match(self.h, key.ctypes.data, kw, r.ctypes.data, C.byref(ref), C.byref(amb))
        return kw, ref.value, amb.value


def _make_lib(rng, n, lo, hi, with_n=False, dup=True):
    seqs = [_rand_seq(rng, int(rng.integers(lo, hi + 1))) for _ in range(n)]
    if dup:                               # near-duplicates => ties and ambiguity
        for i in range(0, n // 4):
            seqs.append(_mutate(rng, seqs[i], int(rng.integers(0, 3))))
    if with_n:
        for i in range(0, len(seqs), 5):
            p = int(rng.integers(0, len(seqs[i])))
            seqs[i] = seqs[i][:p] + "N" + seqs[i][p + 1:]
    return seqs


def _reads_from(rng, seqs, n, lo, hi, with_n=False):
    out = []
    for _ in range(n):
        s = seqs[int(rng.integers(0, len(seqs)))]
        L = int(rng.integers(lo, hi + 1))
        if rng.random() < 0.15:
            r = _rand_seq(rng, L)
        else:
            L = min(L, len(s)); a = int(rng.integers(0, len(s) - L + 1))
            r = s[a:a + L].replace("N", "A")
            r = _mutate(rng, r, int(rng.integers(0, 4)))
            if rng.random() < 0.3:   # flanks for -5/-3 trimming
                r = _rand_seq(rng, 1) + r
            if rng.random() < 0.5:
                r = _rc(r)
        if with_n and rng.random() < 0.3:
            p = int(rng.integers(0, len(r))); r = r[:p] + "N" + r[p + 1:]
        out.append(r)
    return out


# flags: 0 = the one-strand CSR index only (small libraries), 1 = both-strand slot tables (+ gapped seed tables at
# kb >= 9: reads with 16..2*kb-1 seed nt), 3 = the same without occupancy bits in front of the slots
# 5 / 7 = the same tables in their compact form (4-byte entries, candidates verified on the text: the LDS path)
@pytest.mark.parametrize("flags", [0, 1, 3, 5, 7])
@pytest.mark.parametrize("seed,kb,with_n,lo,hi", [
    (1, 4, False, 16, 47), (2, 6, False, 16, 47), (3, 8, False, 16, 56), (4, 5, True, 16, 60),
    (5, 7, True, 50, 124), (6, 4, False, 1, 20), (7, 10, False, 16, 47), (8, 3, True, 3, 30),
    (9, 12, False, 16, 30), (10, 11, True, 14, 40), (11, 9, False, 16, 24), (12, 12, True, 16, 26)])
def test_match_one_equals_brute_force(hostsim, oracle, seed, kb, with_n, lo, hi, flags):
    rng = np.random.default_rng(seed)
    seqs = _make_lib(rng, 24, 20, 140, with_n=with_n)
    sim = Sim(hostsim, seqs, kb, flags)
    olib = OracleLib(seqs)
    reads = _reads_from(rng, seqs, 250, lo, hi, with_n=with_n)
    extra_rounds = [(0, 0, 0, 2, 1, 10, 3, 3), (0, 0, 0, 0, 0, 5, 1, 1), (0, 0, 0, 0, 3, 28, 2, 0), (0, 0, 0, 0, 0, 12, 0, 3),
                    (0, 0, 0, 0, 0, 20, 1, 1), (0, 0, 0, 0, 0, 28, 2, 1), (0, 0, 0, 1, 2, 28, 1, 2), (0, 0, 0, 0, 0, 17, 1, 3)]
    n_aligned = 0
    for rd in reads:
        for rnd in list(MIRGE_ROUNDS) + extra_rounds:
            h = olib.match(rd, rnd)
            kw, ref, amb = sim.match(rd, rnd)
            assert (kw == 2) == (len(rd) <= 56 and "N" not in rd)
            assert ref == h.ref, (rd, rnd, ref, h.ref, h.off, h.fw, h.mm, h.stratum)
            assert amb == h.ambiguous, (rd, rnd)
            n_aligned += h.ref >= 0
    assert n_aligned > 100


@pytest.mark.parametrize("seed,kb,lo,hi", [(21, 5, 1, 60), (22, 8, 100, 400), (23, 10, 125, 300), (24, 4, 20, 130), (25, 12, 57, 260)])
def test_match_long_equals_brute_force(hostsim, oracle, seed, kb, lo, hi):
    """match_long (kernels.h: the matcher of the verbatim uniques -- reads longer than 124 nt, lower case, IUPAC codes,
    '.') against the oracle's exhaustive matcher, and against match_one wherever a key holds the read too."""
    rng = np.random.default_rng(seed)
    seqs = _make_lib(rng, 16, 60, 700, with_n=True)
    sim = Sim(hostsim, seqs, kb, 0)
    olib = OracleLib(seqs)
    reads = _reads_from(rng, seqs, 160, lo, hi, with_n=True)
    for i in range(0, len(reads), 3):                 # characters only the verbatim class takes
        r = reads[i]
        u = rng.random()
        if u < 0.4:
            reads[i] = r.lower()
        elif u < 0.8:
            p = int(rng.integers(0, len(r))); reads[i] = r[:p] + str(rng.choice(list("RYKMSWBDHV.x"))) + r[p + 1:]
        else:
            reads[i] = "".join(c.lower() if rng.random() < 0.5 else c for c in r)
    extra_rounds = [(0, 0, 0, 2, 1, 10, 3, 3), (0, 0, 0, 0, 3, 28, 2, 0), (0, 0, 0, 0, 0, 12, 0, 3), (0, 0, 0, 1, 2, 28, 1, 2), (0, 0, 0, 0, 0, 40, 2, 3)]
    n_aligned = 0
    for rd in reads:
        for rnd in list(MIRGE_ROUNDS) + extra_rounds:
            h = olib.match(rd, rnd)
            r = np.array(rnd, dtype=np.int32); ref = C.c_int32(); amb = C.c_int32()
            hostsim.sim_match_long(sim.h, rd.encode(), len(rd), r.ctypes.data, C.byref(ref), C.byref(amb))
            assert ref.value == h.ref, (rd, rnd, ref.value, h.ref, h.off, h.fw, h.mm, h.stratum)
            assert amb.value == h.ambiguous, (rd, rnd)
            if len(rd) <= 124 and all(c in "ACGTN" for c in rd):
                assert sim.match(rd, rnd)[1:] == (ref.value, amb.value)
            n_aligned += h.ref >= 0
    assert n_aligned > 60


def test_oracle_indexed_equals_brute(oracle):
    rng = np.random.default_rng(11)
    seqs = _make_lib(rng, 40, 20, 200, with_n=True)
    olib = OracleLib(seqs, kb=5)
    for rd in _reads_from(rng, seqs, 400, 14, 60, with_n=True):
        for rnd in MIRGE_ROUNDS:
            a = olib.match(rd, rnd); b = olib.match(rd, rnd, indexed=True)
            assert (a.stratum, a.ref, a.off, a.fw, a.mm, a.ambiguous) == (b.stratum, b.ref, b.off, b.fw, b.mm, b.ambiguous), (rd, rnd)


def test_pack_roundtrip_and_errors(hostsim):
    rng = np.random.default_rng(5)
    key = np.zeros(6, dtype=np.uint64)
    for L in [0, 1, 16, 31, 32, 33, 55, 56, 57, 64, 65, 96, 123, 124]:
        s = _rand_seq(rng, L)
        kw = hostsim.sim_pack(s.encode(), L, key.ctypes.data)
        assert kw == (2 if L <= 56 else 6)
        sw = 2 if kw == 2 else 4
        assert int(key[sw - 1]) >> 56 == L
        dec = "".join("ACGT"[(int(key[i >> 5]) >> (2 * (i & 31))) & 3] for i in range(L))
        assert dec == s
    assert hostsim.sim_pack(b"ACGTN", 5, key.ctypes.data) == 6 and int(key[4]) == 16
    assert hostsim.sim_pack(b"ACGTa", 5, key.ctypes.data) == -4
    assert hostsim.sim_pack(b"AC.T", 4, key.ctypes.data) == -4
    assert hostsim.sim_pack(b"A" * 125, 125, key.ctypes.data) == -5
