import os, sys, time, faulthandler
sys.path.insert(0, os.getcwd())
faulthandler.dump_traceback_later(170, exit=True)
import numpy as np, torch
from mirge_amd import capi, synth
S = int(sys.argv[1]); n = int(sys.argv[2])
libs, _ = synth.make_libraries(synth.Profile.human(), synth.BASE_SEED)
ctx = capi.Context(None, S, 0)
for i, nm in enumerate(synth.LIB_NAMES):
    ctx.load_library(i, libs[nm].names, libs[nm].text, libs[nm].offs)
tcol = 0.0
for s in range(S):
    seqs, lens = synth.make_reads(libs, n, synth.BASE_SEED + 5000 + s)
    t0 = time.time(); ctx.collapse_seqs(seqs, s, lens); tcol += time.time() - t0
    faulthandler.dump_traceback_later(170, exit=True)
    if s % 4 == 3: print("sample", s, "uniques so far", ctx.n_unique(), "collapse (incl. H2D + pack) s so far %.2f" % tcol, flush=True)
t0 = time.time(); ctx.annotate(); ta = time.time() - t0
cv = ctx.counts()
print("annotate s %.4f" % ta, ctx.kernel_times(), flush=True)
print("U", ctx.n_unique(), "per-sample totals ok:", bool((cv["cat"].sum(axis=0) == n).all()), bool((cv["lenhist"].sum(axis=0) == n).all()),
      "tuniq", cv["tuniq"][:4], "stats", [ctx.round_stats(r + 1)[:2] for r in range(5)])
uid, smp, cnt = ctx.export_counts()
print("entries", len(cnt), "sum", int(cnt.sum()), S * n, "tuniq sum", int(cv["tuniq"].sum()))
