"""ctypes wrapper over the CPU oracle (oracle/mirge_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by the product package `mirge_amd`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libmirge_oracle.so")

# miRge.pl:326, 333-339 -- the five rounds (lib, len_lt, len_gt, -5, -3, -l, -n, floor(-e/30))
MIRGE_ROUNDS = [
    (0, 26, 0, 0, 0, 28, 0, 2),
    (1, 0, 25, 0, 0, 28, 1, 2),
    (2, 0, 0, 0, 0, 28, 1, 2),
    (3, 0, 0, 0, 0, 28, 0, 2),
    (0, 0, 0, 1, 2, 15, 2, 2),
]


class Round(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("lib", "len_lt", "len_gt", "trim5", "trim3", "seedlen", "seedmm", "maxmm")]


class Hit(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("stratum", "ref", "off", "fw", "mm", "ambiguous")] + [("nhits", C.c_int64)]


def build(force: bool = False) -> str:
    src = [os.path.join(HERE, f) for f in ("mirge_oracle.c", "mirge_oracle.h", "orc_bowtie.c", "Makefile")]
    if force or not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in src):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            build()
        L = C.CDLL(SO)
        L.orc_lib_new.restype = C.c_void_p
        L.orc_lib_new.argtypes = [C.c_int32, C.c_char_p, C.c_void_p, C.c_int32]
        L.orc_lib_free.argtypes = [C.c_void_p]
        L.orc_match_brute.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.POINTER(Round), C.POINTER(Hit)]
        L.orc_match_indexed.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.POINTER(Round), C.POINTER(Hit)]
        L.orc_match_list.restype = C.c_int64
        L.orc_match_list.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.POINTER(Round), C.c_void_p, C.c_int64]
        L.orc_collapse.restype = C.c_int64
        L.orc_collapse.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_cascade.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                  C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_summarize.argtypes = [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
        _lib = L
    return _lib


def make_round(t) -> Round:
    return Round(*t)


def rounds_array(rounds=MIRGE_ROUNDS):
    arr = (Round * len(rounds))()
    for i, t in enumerate(rounds):
        arr[i] = Round(*t)
    return arr


class OracleLib:
    """One reference library (list of sequences) held by the C oracle."""

    def __init__(self, seqs, kb: int = 0):
        if isinstance(seqs, tuple):                        # (text uint8, offs int64)
            text, offs = seqs
            text = np.ascontiguousarray(text, dtype=np.uint8)
            offs = np.ascontiguousarray(offs, dtype=np.uint64)
        else:
            text = np.frombuffer("".join(seqs).encode(), dtype=np.uint8)
            offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
            np.cumsum([len(s) for s in seqs], out=offs[1:])
        self.n = len(offs) - 1
        self._h = lib().orc_lib_new(self.n, text.ctypes.data_as(C.c_char_p), offs.ctypes.data, kb)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_lib_free(self._h)
            self._h = None

    def match(self, read: str, rnd, indexed: bool = False) -> Hit:
        h = Hit()
        r = rnd if isinstance(rnd, Round) else Round(*rnd)
        f = lib().orc_match_indexed if indexed else lib().orc_match_brute
        f(self._h, read.encode(), len(read), C.byref(r), C.byref(h))
        return h

    def match_list(self, read: str, rnd, cap: int = 4096):
        r = rnd if isinstance(rnd, Round) else Round(*rnd)
        buf = np.zeros((cap, 4), dtype=np.int32)
        n = lib().orc_match_list(self._h, read.encode(), len(read), C.byref(r), buf.ctypes.data, cap)
        return n, buf[:min(n, cap)]


def _concat(seqs, lens=None):
    """list[str] or (uint8[n,W], lens) -> (uint8 concatenated, uint64 offs[n+1])."""
    if lens is not None:
        lens = np.asarray(lens, dtype=np.int64)
        W = seqs.shape[1]
        mask = np.arange(W)[None, :] < lens[:, None]
        text = np.ascontiguousarray(seqs[mask])
        offs = np.zeros(len(lens) + 1, dtype=np.uint64)
        np.cumsum(lens, out=offs[1:])
        return text, offs
    text = np.frombuffer("".join(seqs).encode(), dtype=np.uint8)
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    np.cumsum([len(s) for s in seqs], out=offs[1:])
    return text, offs


def collapse(seqs, lens=None):
    """miRge.pl:300-307.  Returns (uid int64[n], rep int64[U]) with ids in first-occurrence order."""
    text, offs = _concat(seqs, lens)
    n = len(offs) - 1
    uid = np.zeros(max(n, 1), dtype=np.int64)
    rep = np.zeros(max(n, 1), dtype=np.int64)
    U = lib().orc_collapse(n, text.ctypes.data, offs.ctypes.data, uid.ctypes.data, rep.ctypes.data)
    return uid[:n], rep[:U]


def cascade(libs, useqs, ulens=None, rounds=MIRGE_ROUNDS, use_index=True, nthreads=1):
    """miRge.pl:346-371 over unique sequences.  libs = 4 OracleLib.
    Returns (ann_round int8[U], ann_ref int32[U], ann_amb int8[U], stats int64[nrounds, 2])."""
    text, offs = _concat(useqs, ulens)
    U = len(offs) - 1
    arr = (C.c_void_p * len(libs))(*[l._h for l in libs])
    ra = rounds_array(rounds)
    ann_round = np.zeros(max(U, 1), dtype=np.int8)
    ann_ref = np.zeros(max(U, 1), dtype=np.int32)
    ann_amb = np.zeros(max(U, 1), dtype=np.int8)
    stats = np.zeros((len(rounds), 2), dtype=np.int64)
    lib().orc_cascade(arr, U, text.ctypes.data, offs.ctypes.data, len(rounds), ra, int(use_index), int(nthreads),
                      ann_round.ctypes.data, ann_ref.ctypes.data, ann_amb.ctypes.data, stats.ctypes.data)
    return ann_round[:U], ann_ref[:U], ann_amb[:U], stats


def summarize(counts, ulen, ann_round, ann_ref, M, maxlen=128, rounds=None):
    """miRge.pl:465-493.  counts uint32[U, S].  Returns dict of quant/iscan [M,S], cat [5,S], tuniq [S], lenhist.
    rounds: None = the reference's five rounds exactly as the Perl spells them; a list of round tuples = a cascade the
    reference cannot express (category by the library of the annotating round, see orc_summarize)."""
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    U, S = counts.shape
    ulen = np.ascontiguousarray(ulen, dtype=np.int32)
    ann_round = np.ascontiguousarray(ann_round, dtype=np.int8)
    ann_ref = np.ascontiguousarray(ann_ref, dtype=np.int32)
    quant = np.zeros((M, S), dtype=np.uint64); iscan = np.zeros((M, S), dtype=np.uint64)
    cat = np.zeros((5, S), dtype=np.uint64); tuniq = np.zeros(S, dtype=np.uint64)
    lenhist = np.zeros((maxlen + 1, S), dtype=np.uint64)
    lib().orc_summarize(U, S, counts.ctypes.data, ulen.ctypes.data, ann_round.ctypes.data, ann_ref.ctypes.data, M,
                        quant.ctypes.data, iscan.ctypes.data, cat.ctypes.data, tuniq.ctypes.data, maxlen,
                        lenhist.ctypes.data, 0 if rounds is None else len(rounds),
                        None if rounds is None else np.array([r[0] for r in rounds], dtype=np.int32).ctypes.data)
    return dict(quant=quant, iscan=iscan, cat=cat, tuniq=tuniq, lenhist=lenhist)


def run_path(libs, sample_reads, rounds=MIRGE_ROUNDS, use_index=True, nthreads=1, maxlen=128, want_uniques=True):
    """Whole hot path on the CPU: sample_reads = list (one per sample) of list[str] or (seqs, lens).
    Returns dict with uniques (list[str]; None with want_uniques=False: a 10M-entry Python list costs more than the C
    code it feeds), counts [U,S], ann_*, stats, the count vectors, and c_seconds = wall time spent inside the three C
    calls (collapse, cascade, count vectors) -- what bench.py's cpu_baseline leg reports."""
    import time
    texts, offs_l, sample_of = [], [], []
    for s, sr in enumerate(sample_reads):
        t, o = _concat(*sr) if isinstance(sr, tuple) else _concat(sr)
        texts.append(t); offs_l.append(o); sample_of.append(np.full(len(o) - 1, s, dtype=np.int64))
    text = np.concatenate(texts) if texts else np.zeros(0, np.uint8)
    lens = np.concatenate([np.diff(o.astype(np.int64)) for o in offs_l]) if offs_l else np.zeros(0, np.int64)
    offs = np.zeros(len(lens) + 1, dtype=np.uint64); np.cumsum(lens, out=offs[1:])
    sample_of = np.concatenate(sample_of) if sample_of else np.zeros(0, np.int64)
    n = len(lens)
    if len(text) == 0:
        text = np.zeros(1, np.uint8)
    uid = np.zeros(max(n, 1), dtype=np.int64); rep = np.zeros(max(n, 1), dtype=np.int64)
    t0 = time.perf_counter()
    U = lib().orc_collapse(n, text.ctypes.data, offs.ctypes.data, uid.ctypes.data, rep.ctypes.data)
    c_seconds = time.perf_counter() - t0
    uid = uid[:n]; rep = rep[:U]
    S = len(sample_reads)
    counts = np.zeros((U, S), dtype=np.uint32)
    np.add.at(counts, (uid, sample_of), 1)
    ulen = lens[rep].astype(np.int32)
    uo = offs[rep].astype(np.int64)
    # the uniques' text, gathered without a Python loop
    uoffs = np.zeros(U + 1, dtype=np.uint64); np.cumsum(ulen, out=uoffs[1:])
    tot = int(uoffs[U])
    src = np.repeat(uo - uoffs[:-1].astype(np.int64), ulen) + np.arange(tot, dtype=np.int64)
    utext = np.ascontiguousarray(text[src]) if tot else np.zeros(1, np.uint8)
    arr = (C.c_void_p * len(libs))(*[l._h for l in libs])
    ra = rounds_array(rounds)
    ann_round = np.zeros(max(U, 1), dtype=np.int8); ann_ref = np.zeros(max(U, 1), dtype=np.int32); ann_amb = np.zeros(max(U, 1), dtype=np.int8)
    stats = np.zeros((len(rounds), 2), dtype=np.int64)
    t0 = time.perf_counter()
    lib().orc_cascade(arr, U, utext.ctypes.data, uoffs.ctypes.data, len(rounds), ra, int(use_index), int(nthreads),
                      ann_round.ctypes.data, ann_ref.ctypes.data, ann_amb.ctypes.data, stats.ctypes.data)
    c_seconds += time.perf_counter() - t0
    ann_round, ann_ref, ann_amb = ann_round[:U], ann_ref[:U], ann_amb[:U]
    t0 = time.perf_counter()
    out = summarize(counts, ulen, ann_round, ann_ref, libs[0].n, maxlen=maxlen,
                    rounds=None if list(rounds) == list(MIRGE_ROUNDS) else rounds)
    c_seconds += time.perf_counter() - t0
    uniques = [utext[int(uoffs[i]):int(uoffs[i + 1])].tobytes().decode() for i in range(U)] if want_uniques else None
    out.update(uniques=uniques, counts=counts, ulen=ulen, ann_round=ann_round, ann_ref=ann_ref, ann_amb=ann_amb,
               stats=stats, uid=uid, c_seconds=c_seconds)
    return out
