"""Debug helper (GPU box): run a seeded case on the GPU and the oracle, describe the uniques that differ."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mirge_amd import capi, synth
from oracle import oracle as orc
from tests import util

prof = getattr(synth.Profile, sys.argv[1] if len(sys.argv) > 1 else "small")()
libs, _ = synth.make_libraries(prof, 102)
samples = [synth.make_reads(libs, 40000, 300)]
with capi.Context(None, 1, 0) as ctx:
    util.load_libs_into(ctx, libs)
    ctx.collapse_seqs(samples[0][0], 0, samples[0][1])
    ctx.annotate()
    got = util.gpu_result_dict(ctx)
olibs = util.oracle_libs(libs, orc)
res = orc.run_path(olibs, samples, use_index=True, nthreads=8, maxlen=124)
exp = util.oracle_result_dict(res)
bad = [s for s in exp if got.get(s) != exp[s]]
print("differing:", len(bad), "of", len(exp))
hist = collections.Counter()
for s in bad[:100000]:
    r = exp[s][0]
    if r == 0:
        hist[("exp0 got", got[s][0])] += 1; continue
    h = olibs[orc.MIRGE_ROUNDS[r - 1][0]].match(s, orc.MIRGE_ROUNDS[r - 1])
    ref_start = int(libs[synth.LIB_NAMES[orc.MIRGE_ROUNDS[r - 1][0]]].offs[h.ref]) + h.ref
    pos = ref_start + h.off
    hist[("r", r, "fw", h.fw, "len", len(s), "mm", h.mm, "got", got[s][0])] += 1
    hist[("r", r, "fw", h.fw, "pos%32", pos % 32)] += 1
for k, v in sorted(hist.items(), key=lambda kv: str(kv[0])):
    print(k, v)
