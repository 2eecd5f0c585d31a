import os, sys, time, faulthandler
sys.path.insert(0, os.getcwd())
faulthandler.dump_traceback_later(150, exit=True)
import numpy as np, torch
from mirge_amd import capi, shard
W, N, POOL = 8, 8_000_000, 60_000_000
g = torch.Generator(device="cuda"); g.manual_seed(3)
ctxs = [capi.Context(None, 1, 0) for _ in range(W)]
k48 = torch.zeros((1, 6), dtype=torch.int64, device="cuda")
sends, allidx = [], []
for r, c in enumerate(ctxs):
    idx = torch.randint(0, POOL, (N,), device="cuda", generator=g, dtype=torch.int64); allidx.append(idx)
    k0 = (idx * 0x9E3779B97F4A7C15) & ((1 << 40) - 1)
    k16 = torch.stack([k0, torch.full_like(k0, 20 << 56)], dim=1).contiguous()
    torch.cuda.synchronize()
    c.collapse_packed_dev(k16.data_ptr(), N, k48.data_ptr(), 0, 0)
    n16, n48 = c.shard_export_sizes(W)
    b16 = torch.empty(max(int(n16.sum()) * shard.REC16, 8), dtype=torch.uint8, device="cuda")
    b48 = torch.empty(8, dtype=torch.uint8, device="cuda")
    c.shard_export(W, b16.data_ptr(), b48.data_ptr())
    sends.append((b16, np.concatenate([[0], np.cumsum(n16)]).astype(np.int64) * shard.REC16))
    print("rank", r, "uniques", c.n_unique(), "to owners", n16.tolist(), flush=True)
want = int(torch.unique(torch.cat(allidx)).numel())
tot = 0; reads = 0
for d, c in enumerate(ctxs):
    r16 = torch.cat([b[o[d]:o[d + 1]] for b, o in sends]); torch.cuda.synchronize()
    t0 = time.time(); c.shard_import(r16.data_ptr(), r16.numel() // shard.REC16, 0, 0); dt = time.time() - t0
    uid, smp, cnt = c.export_counts(); tot += c.n_unique(); reads += int(cnt.sum())
    print("owner", d, "records", r16.numel() // shard.REC16, "uniques", c.n_unique(), "import s %.3f" % dt, flush=True)
print("distinct overall", want, "sum of owned uniques", tot, "reads", reads, W * N, "OK" if (want == tot and reads == W * N) else "MISMATCH")
