import os, sys, time, faulthandler
sys.path.insert(0, os.getcwd())
faulthandler.dump_traceback_later(100, exit=True)
import numpy as np, torch
from mirge_amd import capi, synth
N = int(sys.argv[1]); pool = int(sys.argv[2])
libs, _ = synth.make_libraries(synth.Profile.human(), synth.BASE_SEED)
ctx = capi.Context(None, 1, 0)
for i, nm in enumerate(synth.LIB_NAMES):
    ctx.load_library(i, libs[nm].names, libs[nm].text, libs[nm].offs)
print("libs loaded", flush=True)
g = torch.Generator(device="cuda"); g.manual_seed(1)
idx = torch.randint(0, pool, (N,), device="cuda", generator=g, dtype=torch.int64)
k0 = (idx * 0x9E3779B97F4A7C15) & ((1 << 40) - 1)      # 20 nt of pseudo-random sequence per pool member
k16 = torch.stack([k0, torch.full_like(k0, 20 << 56)], dim=1).contiguous()
k48 = torch.zeros((1, 6), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
for it in range(2):
    t0 = time.time(); ctx.reset(); ctx.collapse_packed_dev(k16.data_ptr(), N, k48.data_ptr(), 0, 0)
    print("collapse", it, time.time() - t0, ctx.n_unique(), flush=True)
    t0 = time.time(); ctx.annotate()
    print("annotate", it, time.time() - t0, ctx.kernel_times(), flush=True)
print([ctx.round_stats(r + 1)[:2] for r in range(5)])
