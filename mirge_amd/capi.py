"""ctypes binding of libmirge_hip.so (include/mirge_hip.h).

There is no fallback: if the library is missing or no HIP device is usable, loading/opening
raises.  Pointers handed to the *_dev entry points are raw device addresses (e.g.
`torch.Tensor.data_ptr()`); torch is plumbing for memory and torch.distributed only.
"""
from __future__ import annotations

import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIRGE_HIP_LIB") or os.path.join(HERE, "libmirge_hip.so")   # override: A/B builds only

LIB_MIRNA, LIB_HAIRPIN, LIB_OTHER, LIB_MRNA = 0, 1, 2, 3
MAX_ROUNDS = 8
MAX_READ_LEN = 1000      # MIRGE_MAX_READ_LEN
MAX_KEY_LEN = 124        # longest read a device-resident key holds
NOALLOC_MAX_KEYS = 1 << 20

EXPORTS = [
    "mirge_open", "mirge_close", "mirge_last_error", "mirge_reset", "mirge_load_library_fasta", "mirge_load_library_ebwt",
    "mirge_load_library_mem", "mirge_lib_size", "mirge_lib_name", "mirge_lib_short_name", "mirge_set_rounds",
    "mirge_collapse_raw_fastq", "mirge_collapse_raw_fastq_mem",
    "mirge_shard_export_sizes", "mirge_shard_export", "mirge_shard_import",
    "mirge_collapse_fastq", "mirge_collapse_fastq_mem", "mirge_collapse_fastq_dev", "mirge_collapse_seqs", "mirge_collapse_ascii_dev", "mirge_pack_dev",
    "mirge_collapse_packed_dev", "mirge_annotate", "mirge_round_stats", "mirge_counts", "mirge_counts_flat_len",
    "mirge_counts_flat_get", "mirge_counts_flat_set", "mirge_counts_flat_get_dev", "mirge_counts_flat_set_dev", "mirge_validate_keys_dev", "mirge_collapse_annotate_packed_dev", "mirge_n_unique", "mirge_max_unique_len", "mirge_unique", "mirge_export_uniques",
    "mirge_verb_export_size", "mirge_verb_export", "mirge_verb_clear", "mirge_verb_import",
    "mirge_n_counts", "mirge_export_counts", "mirge_last_timing", "mirge_kernel_times", "mirge_set_kernel_timing", "mirge_write_tables",
    "mirge_collapse_keys16_noalloc",
    "mirge_collapse_fastq_batch", "mirge_round_handed_on", "mirge_group_open", "mirge_group_open_ex", "mirge_group_close", "mirge_group_last_error", "mirge_group_size",
    "mirge_group_ctx", "mirge_group_reset", "mirge_group_load_library_mem", "mirge_group_collapse_seqs", "mirge_group_collapse_fastq_batch",
    "mirge_group_annotate", "mirge_group_counts", "mirge_group_round_stats", "mirge_group_n_unique", "mirge_group_write_tables", "mirge_group_timing",
]


MIRGE_E_ARG, MIRGE_E_IO, MIRGE_E_HIP, MIRGE_E_FORMAT, MIRGE_E_LIMIT = -1, -2, -3, -4, -5    # include/mirge_hip.h


class MirgeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libmirge_hip error %d: %s" % (code, msg))
        self.code = code


class Round(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("lib", "len_lt", "len_gt", "trim5", "trim3", "seedlen", "seedmm", "maxmm")]


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 (soname libamdhip64.so.7).
    # Importing torch first makes the loader bind libmirge_hip.so's NEEDED libamdhip64.so.7 to that
    # copy; the other order leaves two runtimes in the process and torch then finds no GPU.
    # Standalone callers (the Perl host, tools/mirge-bowtie) use the system ROCm runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32, u32, u64 = C.c_void_p, C.c_int32, C.c_uint32, C.c_uint64
    L.mirge_open.restype = vp; L.mirge_open.argtypes = [C.c_char_p, i32, i32]
    L.mirge_close.restype = None; L.mirge_close.argtypes = [vp]
    L.mirge_last_error.restype = C.c_char_p; L.mirge_last_error.argtypes = [vp]
    L.mirge_reset.argtypes = [vp, i32]
    L.mirge_load_library_fasta.argtypes = [vp, i32, C.c_char_p]
    L.mirge_load_library_ebwt.argtypes = [vp, i32, C.c_char_p]
    L.mirge_write_tables.argtypes = [vp, C.c_char_p, C.c_char_p, vp, vp, vp]
    L.mirge_load_library_mem.argtypes = [vp, i32, i32, C.c_char_p, vp, vp]
    L.mirge_lib_size.argtypes = [vp, i32]
    L.mirge_lib_name.restype = C.c_char_p; L.mirge_lib_name.argtypes = [vp, i32, i32]
    L.mirge_lib_short_name.restype = C.c_char_p; L.mirge_lib_short_name.argtypes = [vp, i32, i32]
    L.mirge_set_rounds.argtypes = [vp, i32, vp]
    L.mirge_collapse_fastq.argtypes = [vp, C.c_char_p, i32, C.POINTER(u64)]
    L.mirge_collapse_raw_fastq.argtypes = [vp, C.c_char_p, i32, C.c_char_p, i32, i32, C.POINTER(u64), C.POINTER(u64), C.POINTER(i32)]
    L.mirge_collapse_raw_fastq_mem.argtypes = [vp, vp, u64, i32, C.c_char_p, i32, i32, C.POINTER(u64), C.POINTER(u64), C.POINTER(i32)]
    L.mirge_shard_export_sizes.argtypes = [vp, i32, vp, vp]
    L.mirge_shard_export.argtypes = [vp, i32, vp, vp]
    L.mirge_shard_import.argtypes = [vp, vp, u64, vp, u64]
    L.mirge_collapse_fastq_mem.argtypes = [vp, vp, u64, i32, C.POINTER(u64)]
    L.mirge_collapse_fastq_dev.argtypes = [vp, vp, u64, i32, C.POINTER(u64)]
    L.mirge_collapse_seqs.argtypes = [vp, vp, vp, u64, i32]
    L.mirge_collapse_ascii_dev.argtypes = [vp, vp, u32, vp, u64, i32]
    L.mirge_pack_dev.argtypes = [vp, vp, u32, vp, u64, vp, C.POINTER(u64), vp, C.POINTER(u64)]
    L.mirge_collapse_packed_dev.argtypes = [vp, vp, u64, vp, u64, i32]
    L.mirge_collapse_keys16_noalloc.argtypes = [vp, vp, u64, vp, vp, C.POINTER(u64)]
    L.mirge_annotate.argtypes = [vp]
    L.mirge_round_stats.argtypes = [vp, i32, C.POINTER(u64), C.POINTER(u64), C.POINTER(C.c_double)]
    L.mirge_round_handed_on.argtypes = [vp, i32, C.POINTER(u64)]
    L.mirge_counts.argtypes = [vp] + [C.POINTER(C.POINTER(u64))] * 5
    L.mirge_counts_flat_len.restype = u64; L.mirge_counts_flat_len.argtypes = [vp]
    L.mirge_counts_flat_get.argtypes = [vp, vp]
    L.mirge_counts_flat_set.argtypes = [vp, vp]
    L.mirge_counts_flat_get_dev.argtypes = [vp, vp]
    L.mirge_counts_flat_set_dev.argtypes = [vp, vp]
    L.mirge_n_unique.restype = u64; L.mirge_n_unique.argtypes = [vp]
    L.mirge_max_unique_len.restype = u32; L.mirge_max_unique_len.argtypes = [vp]
    L.mirge_validate_keys_dev.argtypes = [vp, vp, u64, vp, u64]
    L.mirge_collapse_annotate_packed_dev.argtypes = [vp, vp, u64, vp, u64, i32]
    L.mirge_verb_export_size.restype = u64; L.mirge_verb_export_size.argtypes = [vp]
    L.mirge_verb_export.argtypes = [vp, vp]
    L.mirge_verb_clear.argtypes = [vp]
    L.mirge_verb_import.argtypes = [vp, vp, u64, i32, i32]
    L.mirge_unique.argtypes = [vp, u64, C.POINTER(C.c_char_p), C.POINTER(i32), C.POINTER(C.c_uint8), C.POINTER(C.POINTER(u32))]
    L.mirge_export_uniques.argtypes = [vp, vp, u32, vp, vp, vp, vp]
    L.mirge_n_counts.restype = u64; L.mirge_n_counts.argtypes = [vp]
    L.mirge_export_counts.argtypes = [vp, vp, vp, vp]
    L.mirge_last_timing.argtypes = [vp] + [C.POINTER(C.c_double)] * 3
    L.mirge_kernel_times.argtypes = [vp, vp, vp, vp, i32]
    L.mirge_set_kernel_timing.argtypes = [vp, i32]
    L.mirge_collapse_fastq_batch.argtypes = [vp, i32, vp, vp, C.c_char_p, i32, i32, vp, vp, vp]
    L.mirge_group_open.restype = vp; L.mirge_group_open.argtypes = [C.c_char_p, i32, i32]
    L.mirge_group_open_ex.restype = vp; L.mirge_group_open_ex.argtypes = [C.c_char_p, i32, i32, vp, u32]
    L.mirge_group_close.restype = None; L.mirge_group_close.argtypes = [vp]
    L.mirge_group_last_error.restype = C.c_char_p; L.mirge_group_last_error.argtypes = [vp]
    L.mirge_group_size.argtypes = [vp]
    L.mirge_group_ctx.restype = vp; L.mirge_group_ctx.argtypes = [vp, i32]
    L.mirge_group_reset.argtypes = [vp, i32]
    L.mirge_group_load_library_mem.argtypes = [vp, i32, i32, C.c_char_p, vp, vp]
    L.mirge_group_collapse_seqs.argtypes = [vp, vp, vp, u64, i32]
    L.mirge_group_collapse_fastq_batch.argtypes = [vp, i32, vp, vp, C.c_char_p, i32, i32, vp, vp, vp]
    L.mirge_group_annotate.argtypes = [vp]
    L.mirge_group_counts.argtypes = [vp] + [C.POINTER(C.POINTER(u64))] * 5
    L.mirge_group_round_stats.argtypes = [vp, i32, C.POINTER(u64), C.POINTER(u64), C.POINTER(C.c_double)]
    L.mirge_group_n_unique.restype = u64; L.mirge_group_n_unique.argtypes = [vp]
    L.mirge_group_write_tables.argtypes = [vp, C.c_char_p, C.c_char_p, vp, vp, vp]
    L.mirge_group_timing.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    _lib = L
    return L


def _batch(fn, obj, paths, samples, adapter, trim5, trim3):
    n = len(paths)
    arr = (C.c_char_p * max(n, 1))(*[p.encode() for p in paths])
    smp = np.ascontiguousarray(samples, dtype=np.int32)
    tot = np.zeros(max(n, 1), dtype=np.uint64); kept = np.zeros(max(n, 1), dtype=np.uint64); ph = np.zeros(max(n, 1), dtype=np.int32)
    obj._chk(fn(obj.h, n, C.cast(arr, C.c_void_p), smp.ctypes.data, adapter.encode() if adapter is not None else None, trim5, trim3,
                tot.ctypes.data, kept.ctypes.data, ph.ctypes.data))
    return tot[:n], kept[:n], ph[:n]


class Context:
    """One mirge_ctx.  Thin: every method is one C-ABI call plus numpy marshalling."""

    def __init__(self, species_dir=None, n_samples=1, device=0):
        self.L = load()
        self.S = n_samples
        self.h = self.L.mirge_open(species_dir.encode() if species_dir else None, n_samples, device)
        if not self.h:
            raise MirgeError(-3, self.L.mirge_last_error(None).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.mirge_close(self.h)
            self.h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc < 0:
            raise MirgeError(rc, self.L.mirge_last_error(self.h).decode())
        return rc

    # ---- libraries
    def load_library(self, lib, names, text=None, offs=None, seqs=None):
        """names: list[str]; either (text uint8, offs int64[n+1]) or seqs list[str]."""
        if seqs is not None:
            text = np.frombuffer("".join(seqs).encode(), dtype=np.uint8)
            offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
            np.cumsum([len(s) for s in seqs], out=offs[1:])
        text = np.ascontiguousarray(text, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        blob = ("\0".join(names) + "\0").encode()
        self._chk(self.L.mirge_load_library_mem(self.h, lib, len(names), blob, text.ctypes.data, offs.ctypes.data))

    def load_library_fasta(self, lib, path):
        self._chk(self.L.mirge_load_library_fasta(self.h, lib, path.encode()))

    def write_tables(self, merges_csv, output_dir, sample_names):
        """summarize bookkeeping + miRNAmerge + filter + the four CSV tables in native code (miRge.pl:453-554, 690-907).
        Returns (mirnaReadsFiltered[S], mirnaUniqFiltered[S])."""
        S = len(sample_names)
        arr = (C.c_char_p * S)(*[n.encode() for n in sample_names])
        tot = np.zeros(S, dtype=np.uint64); uq = np.zeros(S, dtype=np.uint64)
        self._chk(self.L.mirge_write_tables(self.h, merges_csv.encode() if merges_csv else None, output_dir.encode(), arr,
                                            tot.ctypes.data, uq.ctypes.data))
        return tot, uq

    def load_library_ebwt(self, lib, base):
        self._chk(self.L.mirge_load_library_ebwt(self.h, lib, base.encode()))

    def lib_size(self, lib):
        return self._chk(self.L.mirge_lib_size(self.h, lib))

    def lib_names(self, lib, short=False):
        f = self.L.mirge_lib_short_name if short else self.L.mirge_lib_name
        return [f(self.h, lib, i).decode() for i in range(self.lib_size(lib))]

    def set_rounds(self, rounds):
        arr = (Round * len(rounds))(*[Round(*r) for r in rounds])
        self._chk(self.L.mirge_set_rounds(self.h, len(rounds), arr))

    def reset(self, n_samples=None):
        if n_samples is not None:
            self.S = n_samples
        self._chk(self.L.mirge_reset(self.h, self.S))

    # ---- collapse
    def collapse_fastq(self, path, sample):
        n = C.c_uint64(0)
        self._chk(self.L.mirge_collapse_fastq(self.h, path.encode(), sample, C.byref(n)))
        return n.value

    def collapse_fastq_batch(self, paths, samples, adapter=None, trim5=0, trim3=0):
        """Many sample files (plain or .gz) in order, host threads reading / inflating ahead -> (total[n], kept[n], phred[n])."""
        return _batch(self.L.mirge_collapse_fastq_batch, self, paths, samples, adapter, trim5, trim3)

    # ---- multi-GPU exchange step (include/mirge_hip.h: mirge_shard_*)
    def shard_export_sizes(self, world):
        n16 = np.zeros(world, dtype=np.uint64); n48 = np.zeros(world, dtype=np.uint64)
        self._chk(self.L.mirge_shard_export_sizes(self.h, world, n16.ctypes.data, n48.ctypes.data))
        return n16, n48

    def shard_export(self, world, d_rec16, d_rec48):
        self._chk(self.L.mirge_shard_export(self.h, world, d_rec16, d_rec48))

    def shard_import(self, d_rec16, n16, d_rec48, n48):
        self._chk(self.L.mirge_shard_import(self.h, d_rec16, int(n16), d_rec48, int(n48)))

    def collapse_raw_fastq(self, path, sample, adapter="none", trim5=0, trim3=0):
        """-> (totalReads, trimmedReads, phred)"""
        tot, kept, ph = C.c_uint64(0), C.c_uint64(0), C.c_int32(0)
        self._chk(self.L.mirge_collapse_raw_fastq(self.h, path.encode(), sample, adapter.encode(), trim5, trim3,
                                                  C.byref(tot), C.byref(kept), C.byref(ph)))
        return tot.value, kept.value, ph.value

    def collapse_raw_fastq_mem(self, data: bytes, sample, adapter="none", trim5=0, trim3=0):
        tot, kept, ph = C.c_uint64(0), C.c_uint64(0), C.c_int32(0)
        buf = C.create_string_buffer(data, len(data)) if len(data) else None
        self._chk(self.L.mirge_collapse_raw_fastq_mem(self.h, C.cast(buf, C.c_void_p) if buf is not None else None, len(data), sample,
                                                      adapter.encode(), trim5, trim3, C.byref(tot), C.byref(kept), C.byref(ph)))
        return tot.value, kept.value, ph.value

    def collapse_fastq_mem(self, data: bytes, sample):
        n = C.c_uint64(0)
        buf = C.create_string_buffer(data, len(data)) if len(data) else None
        self._chk(self.L.mirge_collapse_fastq_mem(self.h, C.cast(buf, C.c_void_p) if buf is not None else None, len(data), sample, C.byref(n)))
        return n.value

    def collapse_fastq_dev(self, d_data, size, sample):
        n = C.c_uint64(0)
        self._chk(self.L.mirge_collapse_fastq_dev(self.h, d_data, size, sample, C.byref(n)))
        return n.value

    def collapse_seqs(self, reads, sample, lens=None):
        """reads: list[str], or (uint8[n, W] matrix with lens)."""
        if lens is not None:
            lens = np.asarray(lens, dtype=np.int64)
            mask = np.arange(reads.shape[1])[None, :] < lens[:, None]
            text = np.ascontiguousarray(reads[mask])
        else:
            text = np.frombuffer("".join(reads).encode(), dtype=np.uint8)
            lens = np.array([len(r) for r in reads], dtype=np.int64)
        offs = np.zeros(len(lens) + 1, dtype=np.uint64)
        np.cumsum(lens, out=offs[1:])
        if len(text) == 0:
            text = np.zeros(1, dtype=np.uint8)
        self._chk(self.L.mirge_collapse_seqs(self.h, text.ctypes.data, offs.ctypes.data, len(lens), sample))

    def collapse_annotate_packed_dev(self, d16, n16, d48, n48, sample):
        self._chk(self.L.mirge_collapse_annotate_packed_dev(self.h, d16, n16, d48, n48, sample))

    def validate_keys_dev(self, d16, n16, d48, n48):
        self._chk(self.L.mirge_validate_keys_dev(self.h, d16, n16, d48, n48))

    def collapse_ascii_dev(self, d_seqs, width, d_lens, n, sample):
        self._chk(self.L.mirge_collapse_ascii_dev(self.h, d_seqs, width, d_lens, n, sample))

    def pack_dev(self, d_seqs, width, d_lens, n, d_keys16, d_keys48):
        n16 = C.c_uint64(0); n48 = C.c_uint64(0)
        self._chk(self.L.mirge_pack_dev(self.h, d_seqs, width, d_lens, n, d_keys16, C.byref(n16), d_keys48, C.byref(n48)))
        return n16.value, n48.value

    def collapse_packed_dev(self, d_keys16, n16, d_keys48, n48, sample):
        self._chk(self.L.mirge_collapse_packed_dev(self.h, d_keys16, n16, d_keys48, n48, sample))

    def collapse_keys16_noalloc(self, d_keys16, n, d_ukeys, d_counts):
        """The no-allocation collapse of one 16-byte key stream into caller buffers -> number of distinct keys."""
        u = C.c_uint64(0)
        self._chk(self.L.mirge_collapse_keys16_noalloc(self.h, d_keys16, n, d_ukeys, d_counts, C.byref(u)))
        return u.value

    # ---- cascade + counts
    def annotate(self):
        self._chk(self.L.mirge_annotate(self.h))

    def round_stats(self, rnd):
        p = C.c_uint64(); a = C.c_uint64(); s = C.c_double()
        self._chk(self.L.mirge_round_stats(self.h, rnd, C.byref(p), C.byref(a), C.byref(s)))
        return p.value, a.value, s.value

    def round_handed_on(self, rnd):
        n = C.c_uint64()
        self._chk(self.L.mirge_round_handed_on(self.h, rnd, C.byref(n)))
        return n.value

    def counts(self):
        ptrs = [C.POINTER(C.c_uint64)() for _ in range(5)]
        self._chk(self.L.mirge_counts(self.h, *[C.byref(p) for p in ptrs]))
        M, S = self.lib_size(LIB_MIRNA), self.S
        shp = [(M, S), (M, S), (5, S), (MAX_READ_LEN + 1, S), (S,)]
        out = []
        for p, sh in zip(ptrs, shp):
            n = int(np.prod(sh))
            out.append(np.ctypeslib.as_array(p, shape=(max(n, 1),))[:n].reshape(sh).copy())
        return dict(quant=out[0], iscan=out[1], cat=out[2], lenhist=out[3], tuniq=out[4])

    def counts_flat(self):
        n = self.L.mirge_counts_flat_len(self.h)
        buf = np.zeros(n, dtype=np.uint64)
        self._chk(self.L.mirge_counts_flat_get(self.h, buf.ctypes.data))
        return buf

    def set_counts_flat(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint64)
        assert len(buf) == self.L.mirge_counts_flat_len(self.h)
        self._chk(self.L.mirge_counts_flat_set(self.h, buf.ctypes.data))

    def counts_flat_len(self):
        return int(self.L.mirge_counts_flat_len(self.h))

    def counts_flat_get_dev(self, d_out):
        self._chk(self.L.mirge_counts_flat_get_dev(self.h, d_out))

    def counts_flat_set_dev(self, d_in):
        self._chk(self.L.mirge_counts_flat_set_dev(self.h, d_in))

    # ---- per-unique results
    def n_unique(self):
        return self.L.mirge_n_unique(self.h)

    def export_uniques(self, with_seqs=True):
        """(seqs[U][stride] zero padded, lens, ann_round, ann_ref, ann_amb); with_seqs=False leaves the sequences out
        (seqs and lens come back as None): the annotation arrays of a 50M-unique table without 7 GB of text."""
        U = self.n_unique()
        stride = max(128, (int(self.L.mirge_max_unique_len(self.h)) + 15) // 16 * 16)
        seqs = np.zeros((max(U, 1), stride), dtype=np.uint8) if with_seqs else None
        lens = np.zeros(max(U, 1), dtype=np.uint16) if with_seqs else None
        rnd = np.zeros(max(U, 1), dtype=np.int8); ref = np.zeros(max(U, 1), dtype=np.int32); amb = np.zeros(max(U, 1), dtype=np.uint8)
        self._chk(self.L.mirge_export_uniques(self.h, seqs.ctypes.data if with_seqs else None, stride,
                                              lens.ctypes.data if with_seqs else None, rnd.ctypes.data, ref.ctypes.data, amb.ctypes.data))
        if not with_seqs:
            return None, None, rnd[:U], ref[:U], amb[:U]
        return seqs[:U], lens[:U], rnd[:U], ref[:U], amb[:U]

    # ---- verbatim uniques across ranks (host blobs)
    def verb_export(self):
        n = int(self.L.mirge_verb_export_size(self.h))
        buf = np.zeros(max(n, 1), dtype=np.uint8)
        self._chk(self.L.mirge_verb_export(self.h, buf.ctypes.data))
        return buf[:n]

    def verb_clear(self):
        self._chk(self.L.mirge_verb_clear(self.h))

    def verb_import(self, blob, rank, world):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        self._chk(self.L.mirge_verb_import(self.h, blob.ctypes.data if len(blob) else None, len(blob), rank, world))

    def export_counts(self):
        n = self.L.mirge_n_counts(self.h)
        uid = np.zeros(max(n, 1), dtype=np.uint64); smp = np.zeros(max(n, 1), dtype=np.uint32); cnt = np.zeros(max(n, 1), dtype=np.uint32)
        self._chk(self.L.mirge_export_counts(self.h, uid.ctypes.data, smp.ctypes.data, cnt.ctypes.data))
        return uid[:n], smp[:n], cnt[:n]

    def unique(self, i):
        seq = C.c_char_p(); refs = (C.c_int32 * MAX_ROUNDS)(); amb = C.c_uint8(); cnt = C.POINTER(C.c_uint32)()
        self._chk(self.L.mirge_unique(self.h, i, C.byref(seq), refs, C.byref(amb), C.byref(cnt)))
        return seq.value.decode(), list(refs), amb.value, [cnt[s] for s in range(self.S)]

    # ---- timing
    def set_kernel_timing(self, on):
        self._chk(self.L.mirge_set_kernel_timing(self.h, 1 if on else 0))

    def last_timing(self):
        a = C.c_double(); b = C.c_double(); c = C.c_double()
        self._chk(self.L.mirge_last_timing(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(collapse_ms=a.value, annotate_ms=b.value, counts_ms=c.value)

    def kernel_times_raw(self):
        """The same without building Python objects per kernel (bench.py calls this inside its timed steps): (n, names,
        ms, bytes) as ctypes arrays owned by the context -- copy what is to be kept."""
        if not hasattr(self, "_kt"):
            cap = 64
            self._kt = ((C.c_char_p * cap)(), (C.c_double * cap)(), (C.c_uint64 * cap)(), cap)
        names, ms, by, cap = self._kt
        n = self._chk(self.L.mirge_kernel_times(self.h, names, ms, by, cap))
        return min(n, cap), names, ms, by

    def kernel_times(self):
        cap = 64
        names = (C.c_char_p * cap)(); ms = (C.c_double * cap)(); by = (C.c_uint64 * cap)()
        n = self._chk(self.L.mirge_kernel_times(self.h, names, ms, by, cap))
        return {names[i].decode(): (ms[i], by[i]) for i in range(min(n, cap))}


GROUP_HOST_STAGED = 1


class _Borrowed(Context):
    """A rank's context inside a Group: same methods, owned (and closed) by the group."""

    def __init__(self, L, h, S):
        self.L, self.h, self.S = L, h, S

    def close(self):
        self.h = None

    __del__ = close


class Group:
    """One mirge_group: a context per GPU inside the library, merge of the unique tables + count all-reduce over RCCL
    (include/mirge_hip.h: mirge_group_*).  flags=GROUP_HOST_STAGED: exchanges through host memory, ranks may share a device."""

    def __init__(self, species_dir=None, n_samples=1, n_ranks=1, devices=None, flags=0):
        self.L = load()
        self.S = n_samples
        dv = np.ascontiguousarray(devices, dtype=np.int32) if devices is not None else None
        self.h = self.L.mirge_group_open_ex(species_dir.encode() if species_dir else None, n_samples, n_ranks,
                                            dv.ctypes.data if dv is not None else None, flags)
        if not self.h:
            raise MirgeError(-3, self.L.mirge_group_last_error(None).decode())
        self.world = n_ranks

    def close(self):
        if getattr(self, "h", None):
            self.L.mirge_group_close(self.h)
            self.h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc < 0:
            raise MirgeError(rc, self.L.mirge_group_last_error(self.h).decode())
        return rc

    def ctx(self, rank):
        return _Borrowed(self.L, self.L.mirge_group_ctx(self.h, rank), self.S)

    def reset(self, n_samples=None):
        if n_samples is not None:
            self.S = n_samples
        self._chk(self.L.mirge_group_reset(self.h, self.S))

    def load_library(self, lib, names, text, offs):
        text = np.ascontiguousarray(text, dtype=np.uint8); offs = np.ascontiguousarray(offs, dtype=np.uint64)
        blob = ("\0".join(names) + "\0").encode()
        self._chk(self.L.mirge_group_load_library_mem(self.h, lib, len(names), blob, text.ctypes.data, offs.ctypes.data))

    def collapse_seqs(self, reads, sample, lens):
        lens = np.asarray(lens, dtype=np.int64)
        mask = np.arange(reads.shape[1])[None, :] < lens[:, None]
        text = np.ascontiguousarray(reads[mask])
        offs = np.zeros(len(lens) + 1, dtype=np.uint64)
        np.cumsum(lens, out=offs[1:])
        self._chk(self.L.mirge_group_collapse_seqs(self.h, text.ctypes.data, offs.ctypes.data, len(lens), sample))

    def collapse_fastq_batch(self, paths, samples, adapter=None, trim5=0, trim3=0):
        return _batch(self.L.mirge_group_collapse_fastq_batch, self, paths, samples, adapter, trim5, trim3)

    def annotate(self):
        self._chk(self.L.mirge_group_annotate(self.h))

    def counts(self):
        ptrs = [C.POINTER(C.c_uint64)() for _ in range(5)]
        self._chk(self.L.mirge_group_counts(self.h, *[C.byref(p) for p in ptrs]))
        M, S = self.ctx(0).lib_size(LIB_MIRNA), self.S
        shp = [(M, S), (M, S), (5, S), (MAX_READ_LEN + 1, S), (S,)]
        out = []
        for p, sh in zip(ptrs, shp):
            n = int(np.prod(sh))
            out.append(np.ctypeslib.as_array(p, shape=(max(n, 1),))[:n].reshape(sh).copy())
        return dict(quant=out[0], iscan=out[1], cat=out[2], lenhist=out[3], tuniq=out[4])

    def round_stats(self, rnd):
        p = C.c_uint64(); a = C.c_uint64(); s = C.c_double()
        self._chk(self.L.mirge_group_round_stats(self.h, rnd, C.byref(p), C.byref(a), C.byref(s)))
        return p.value, a.value, s.value

    def n_unique(self):
        return int(self.L.mirge_group_n_unique(self.h))

    def timing(self):
        a = C.c_double(); b = C.c_double()
        self._chk(self.L.mirge_group_timing(self.h, C.byref(a), C.byref(b)))
        return dict(merge_ms=a.value, allreduce_ms=b.value)

    def write_tables(self, merges_csv, output_dir, sample_names):
        S = len(sample_names)
        arr = (C.c_char_p * S)(*[n.encode() for n in sample_names])
        tot = np.zeros(S, dtype=np.uint64); uq = np.zeros(S, dtype=np.uint64)
        self._chk(self.L.mirge_group_write_tables(self.h, merges_csv.encode() if merges_csv else None, output_dir.encode(), arr,
                                                  tot.ctypes.data, uq.ctypes.data))
        return tot, uq
