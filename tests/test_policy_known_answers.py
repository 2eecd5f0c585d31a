"""Known-answer tests of the matching policy, built by construction (SURVEY.md 8c lists them; bowtie 1.1.1
itself is absent, so these pin OUR reading of its manual, not bowtie -- parity with bowtie stays unpinned).
Each case is checked on the oracle's exhaustive matcher, on its indexed matcher, and on the host build of the
kernel arithmetic (tests/hostsim); tests/test_gpu_parity.py::test_policy_cases_on_gpu runs them on the device."""
import ctypes as C

import numpy as np
import pytest

from oracle.oracle import MIRGE_ROUNDS, OracleLib

R1, R2, R3, R4, R5 = MIRGE_ROUNDS


def rc(s):
    return s[::-1].translate(str.maketrans("ACGTN", "TGCAN"))


def mut(s, *pos):
    s = list(s)
    for p in pos:
        s[p] = "ACGT"[("ACGT".index(s[p]) + 2) % 4]
    return "".join(s)


def build_cases():
    rng = np.random.default_rng(2014)
    rs = lambda n: "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))
    refA, refB, refC = rs(60), rs(30), rs(50) + "N" + rs(50)
    lib = [refA, refB, refC]
    w = refA[5:45]                                  # a 40-nt window of refA
    cases = [
        ("planted exact hit", refA[10:32], R1, 0),
        ("exact hit on the reverse strand", rc(refA[10:32]), R1, 0),
        ("round 1 (-n 0, len < 28): one mismatch is no hit", mut(refA[10:32], 7), R1, -1),
        ("1 mm inside the 28-nt seed, -n 1", mut(w, 10), R3, 0),
        ("2 mm inside the seed, -n 1: invalid", mut(w, 3, 20), R3, -1),
        ("1 mm in seed + 1 mm beyond nt 28: valid (<= 2 overall)", mut(w, 10, 35), R3, 0),
        ("2 mm both beyond the seed, -n 1: valid", mut(w, 30, 38), R3, 0),
        ("3 mm (1 seed + 2 outside): invalid (-e 70 => 2)", mut(w, 5, 30, 38), R3, -1),
        ("round 4 (-n 0): a seed mismatch is rejected", mut(w, 27), R4, -1),
        ("round 4: 2 mm beyond nt 28 are fine", mut(w, 28, 39), R4, 0),
        ("round 4: 3 mm beyond nt 28 are not", mut(w, 28, 33, 39), R4, -1),
        ("round 5 trims 1 nt 5' and 2 nt 3', then <= 2 mm", "T" + mut(refA[20:42], 2, 18) + "GG", R5, 0),
        ("round 5: untrimmed flanks would have been 3 extra mismatches", rc("T" + refA[20:42] + "GG")[0:25], R5, 0),
        ("round 5: 3 mm after trimming: invalid", "A" + mut(refA[20:42], 2, 9, 18) + "CC", R5, -1),
        ("round 5, 16-nt read: 13 nt left, shorter than -l 15", "C" + refA[30:43] + "TT", R5, 0),
        ("read longer than the reference: no hit", refB + "ACGT", R3, -1),
        ("a window may not span two references", refA[-12:] + refB[:12], R3, -1),
        ("read N costs a mismatch", refA[10:20] + "N" + refA[21:40], R3, 0),
        ("read N in round 1 (-n 0): no hit", refA[10:20] + "N" + refA[21:32], R1, -1),
        ("reference N is not alignable", refC[40:70].replace("N", "A"), R3, -1),
        ("reference N: the window next to it is", refC[51:80], R3, 2),
        ("reads of <= -n nt are skipped", "AC", R5, -1),
        ("empty read", "", R3, -1),
    ]
    return lib, cases


def test_cases_cover_both_outcomes():
    lib, cases = build_cases()
    assert {c[3] >= 0 for c in cases} == {True, False}


@pytest.mark.parametrize("indexed", [False, True])
def test_oracle_policy(oracle, indexed):
    lib, cases = build_cases()
    L = OracleLib(lib, kb=4)
    for what, read, rnd, ref in cases:
        h = L.match(read, rnd, indexed=indexed)
        assert h.ref == ref, what


def test_best_stratum_and_tie_break(oracle):
    """-a --best --strata: only the lowest seed-mismatch stratum counts; inside it hypothesis H1 keeps the
    maximum (mismatches, reference, offset, strand)."""
    core = "ACGTTGCATGCCATGAATCGTACG"                       # 24 nt
    lib = [core, mut(core, 20), "GG" + core + "TT", mut(core, 3)]
    L = OracleLib(lib, kb=4)
    h = L.match(core, R5)                                      # trimmed to core[1:22]: exact in refs 0, 2, 3(mm at 3 -> pos 2)
    n, hits = L.match_list(core, R5)
    strata_ok = all(int(x[3]) >= 0 for x in hits)
    assert strata_ok and h.stratum == 0 and h.ambiguous == 1
    assert h.ref == max(int(x[0]) for x in hits if int(x[3]) == max(int(y[3]) for y in hits))
    # a read with one seed mismatch everywhere except an exact copy: only the exact copy is reported
    h = L.match(mut(core, 20), R3)
    assert (h.ref, h.stratum, h.ambiguous) == (1, 0, 0)


def test_hostsim_policy(hostsim, oracle):
    lib, cases = build_cases()
    text = "".join(lib).encode()
    offs = np.zeros(len(lib) + 1, dtype=np.uint64); np.cumsum([len(s) for s in lib], out=offs[1:])
    for kb in (3, 5, 8):
        h = hostsim.sim_lib_new(len(lib), text, offs.ctypes.data, kb, kb & 1)
        for what, read, rnd, ref in cases:
            key = np.zeros(6, dtype=np.uint64)
            kw = hostsim.sim_pack(read.encode(), len(read), key.ctypes.data)
            r = np.array(rnd, dtype=np.int32); got = C.c_int32(); amb = C.c_int32()
            hostsim.sim_match(h, key.ctypes.data, kw, r.ctypes.data, C.byref(got), C.byref(amb))
            assert got.value == ref, (what, kb)
        hostsim.sim_lib_free(h)
