"""BASELINE.json configs[4] through FILES: S samples x n reads, every sample a trimmed FASTQ file on disk, read by
mirge_collapse_fastq (reader threads -> pinned ring -> DMA -> record split and pack on the device), one shared table,
then the cascade and the per-sample count tables.  Usage: python -u tools/fullsize/batch48_files.py 48 2000000 [dir]"""
import os, sys, time, tempfile, shutil, faulthandler
sys.path.insert(0, os.getcwd())
faulthandler.dump_traceback_later(1100, exit=True)
import numpy as np
from mirge_amd import capi, synth
S = int(sys.argv[1]); n = int(sys.argv[2])
d = sys.argv[3] if len(sys.argv) > 3 else tempfile.mkdtemp(prefix="mirge_b48_")
libs, _ = synth.make_libraries(synth.Profile.human(), synth.BASE_SEED)
ctx = capi.Context(None, S, 0)
for i, nm in enumerate(synth.LIB_NAMES):
    ctx.load_library(i, libs[nm].names, libs[nm].text, libs[nm].offs)
t_write = t_col = 0.0; nbytes = 0
path = os.path.join(d, "sample.fastq")              # one file at a time on disk: written, read by the library, overwritten
for s in range(S):
    seqs, lens = synth.make_reads(libs, n, synth.BASE_SEED + 5000 + s)
    t0 = time.time(); nbytes += synth.write_fastq_fast(path, seqs, lens); t_write += time.time() - t0
    t0 = time.time(); got = ctx.collapse_fastq(path, s); t_col += time.time() - t0
    assert got == n
    if s % 4 == 3:
        print("sample %d: uniques so far %d, file -> table %.2f s so far (%.2f GB/s of FASTQ text), writing the files %.1f s" %
              (s, ctx.n_unique(), t_col, nbytes / t_col / 1e9, t_write), flush=True)
t0 = time.time(); ctx.annotate(); ta = time.time() - t0
cv = ctx.counts()
print("files -> table: %.2f s for %d reads = %.1f M reads/s; cascade + counts %.4f s" % (t_col, S * n, S * n / t_col / 1e6, ta))
print("kernels", {k: round(v[0], 3) for k, v in ctx.kernel_times().items()})
print("U", ctx.n_unique(), "per-sample totals ok:", bool((cv["cat"].sum(axis=0) == n).all()), bool((cv["lenhist"].sum(axis=0) == n).all()),
      "stats", [ctx.round_stats(r + 1)[:2] for r in range(5)])
shutil.rmtree(d, ignore_errors=True)
