"""ctypes wrapper over oracle/trim_oracle.c (the trimming step before the hot path).
TEST INFRASTRUCTURE ONLY: never imported by the product package `mirge_amd`."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libtrim_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(HERE, "trim_oracle.c")
        if not os.path.exists(SO) or os.path.getmtime(src) > os.path.getmtime(SO):
            subprocess.check_call(["make", "-C", HERE, "-s"])
        L = C.CDLL(SO)
        L.orc_quality_trim.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_adapter_back.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p]
        L.orc_trim_record.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_trim_fastq.restype = C.c_int64
        L.orc_trim_fastq.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_int, C.c_int, C.c_int64,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def quality_trim(qual: bytes, base=33, cutoff_front=0, cutoff_back=10):
    a, b = C.c_int(0), C.c_int(0)
    lib().orc_quality_trim(qual, len(qual), cutoff_front, cutoff_back, base, C.byref(a), C.byref(b))
    return a.value, b.value


def adapter_back(adapter: bytes, read: bytes, rate=0.12, min_overlap=3, anchored=False):
    """(astart, astop, rstart, rstop, matches, errors) or None; anchored: the adapter was written 'ADAPTER$'"""
    out = np.zeros(6, dtype=np.int32)
    ok = lib().orc_adapter_back(adapter, len(adapter), read, len(read), rate, min_overlap, int(anchored), out.ctypes.data)
    return tuple(int(x) for x in out) if ok else None


def trim_fastq(data: bytes, adapters: str, trim5=0, trim3=0):
    """-> (list of trimmed sequences of the records kept, total records, worker phred base, reported phred)"""
    cap = data.count(b"\n") // 4 + 2
    keep = np.zeros(cap, dtype=np.int32); start = np.zeros(cap, dtype=np.int32); ln = np.zeros(cap, dtype=np.int32)
    off = np.zeros(cap, dtype=np.uint64); ph = np.zeros(2, dtype=np.int32)
    n = lib().orc_trim_fastq(data, len(data), adapters.encode(), trim5, trim3, cap, keep.ctypes.data, start.ctypes.data,
                             ln.ctypes.data, off.ctypes.data, ph.ctypes.data)
    if n < 0:
        raise ValueError("orc_trim_fastq: %d" % n)
    out = [data[int(off[i]) + int(start[i]): int(off[i]) + int(start[i]) + int(ln[i])] for i in range(n) if keep[i]]
    return out, int(n), int(ph[0]), int(ph[1])
