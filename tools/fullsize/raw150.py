"""Raw 150-nt reads with the adapter still on (lines no key holds: trimmed on their bytes): n reads through
mirge_collapse_raw_fastq_mem, device time of the trimming kernel.  With adapter "none" nearly every read stays longer than
124 nt: the verbatim class (host-side string table) takes the whole run.
Usage: python -u tools/fullsize/raw150.py [n] [read_len] [adapter]"""
import os, sys, time, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np
from mirge_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
rl = int(sys.argv[2]) if len(sys.argv) > 2 else 150
adapter = sys.argv[3] if len(sys.argv) > 3 else "illumina"
libs, _ = synth.make_libraries(synth.Profile.small(), 5)
seqs, lens = synth.make_reads(libs, n, 6)
d = tempfile.mkdtemp(prefix="mirge_raw150_")
p = os.path.join(d, "raw.fastq")
nb = synth.write_raw_fastq_fast(p, seqs, lens, 7, read_len=rl)
data = open(p, "rb").read()
with capi.Context(None, 1, 0) as ctx:
    for rep in range(3):
        ctx.reset()
        t0 = time.perf_counter()
        total, kept, phred = ctx.collapse_raw_fastq_mem(data, 0, adapter, 0, 0)
        dt = time.perf_counter() - t0
        kt = ctx.kernel_times()
        print("read_len %d: %d reads, %d kept, %.1f MB, call %.1f ms, kernels %s" %
              (rl, total, kept, nb / 1e6, dt * 1e3, {k: round(v[0], 2) for k, v in kt.items() if k.startswith("fastq")}), flush=True)
    for i, nm in enumerate(synth.LIB_NAMES):
        ctx.load_library(i, libs[nm].names, libs[nm].text, libs[nm].offs)
    t0 = time.perf_counter(); ctx.annotate(); dt = time.perf_counter() - t0
    print("uniques", ctx.n_unique(), "cascade + counts %.1f ms" % (dt * 1e3), [ctx.round_stats(r + 1)[:2] for r in range(5)])
os.remove(p); os.rmdir(d)
