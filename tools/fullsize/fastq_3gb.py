import os, sys, time, faulthandler
sys.path.insert(0, os.getcwd())
faulthandler.dump_traceback_later(170, exit=True)
import numpy as np
from mirge_amd import capi, synth
libs, _ = synth.make_libraries(synth.Profile.small(), synth.BASE_SEED + 9)
seqs, lens = synth.make_reads(libs, 4_000_000, synth.BASE_SEED + 11)
blk = "/tmp/blk.fastq"; big = "/tmp/big.fastq"
nb = synth.write_fastq_fast(blk, seqs, lens)
data = open(blk, "rb").read()
REP = 12
with open(big, "wb") as f:
    for _ in range(REP): f.write(data)
print("file bytes", os.path.getsize(big), flush=True)
ctx = capi.Context(None, 1, 0)
t0 = time.time(); n1 = ctx.collapse_fastq(blk, 0); t1 = time.time()
uid, smp, c1 = ctx.export_counts(); u1 = ctx.n_unique()
s1, l1, *_ = ctx.export_uniques()
tab1 = {s1[i, :l1[i]].tobytes(): 0 for i in range(0)}     # (not needed: compare via sorted counts + keys below)
k1 = sorted(zip([s1[i, :l1[i]].tobytes() for i in range(u1)], np.bincount(uid.astype(np.int64), weights=c1, minlength=u1).astype(np.int64).tolist()))
ctx.reset()
faulthandler.dump_traceback_later(170, exit=True)
t0 = time.time(); n2 = ctx.collapse_fastq(big, 0); t2 = time.time() - t0
print("reads", n1, n2, "seconds for the big file %.2f (%.2f GB/s)" % (t2, os.path.getsize(big) / t2 / 1e9), flush=True)
uid, smp, c2 = ctx.export_counts(); u2 = ctx.n_unique()
s2, l2, *_ = ctx.export_uniques()
k2 = sorted(zip([s2[i, :l2[i]].tobytes() for i in range(u2)], np.bincount(uid.astype(np.int64), weights=c2, minlength=u2).astype(np.int64).tolist()))
ok = n2 == REP * n1 and u1 == u2 and all(a[0] == b[0] and b[1] == REP * a[1] for a, b in zip(k1, k2))
print("uniques", u1, u2, "OK" if ok else "MISMATCH")
os.unlink(blk); os.unlink(big)
