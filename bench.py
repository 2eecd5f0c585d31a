#!/usr/bin/env python3
"""bench.py -- reads/s through collapse + 5-round library cascade + count vectors on MI355X.

One "step" = one pass of the whole hot path over one batch of synthetic adapter-trimmed reads
whose 2-bit keys are already resident in HBM: mirge_reset -> mirge_collapse_packed_dev
-> mirge_annotate (cascade + count vectors) [-> one RCCL sum all-reduce of the count vectors when
N > 1].  Workload at N = 1: BASELINE.json configs[1] -- human-profile synthetic seqLibs, 20M reads.
For N > 1 every rank gets its own 20M-read shard (weak scaling; independent shards, no data-path
exchange, only the count-vector all-reduce).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=20_000_000, help="reads per GPU")
    ap.add_argument("--profile", default="human", choices=["human", "mouse", "medium", "small", "tiny"])
    ap.add_argument("--mixture", default="default", choices=["default", "isomir_rich"])
    ap.add_argument("--cpu-sample", type=int, default=100_000, help="reads of the same workload timed on the CPU oracle (0 = skip)")
    ap.add_argument("--chunk", type=int, default=1_000_000)
    ap.add_argument("--merge-uniques", action="store_true",
                    help="N > 1: all-to-all of the unique tables by key owner between collapse and cascade (exact trimmed_uniq / "
                         "per-round unique counts; off by default: the shards of this synthetic workload are independent)")
    ap.add_argument("--e2e-reads", type=int, default=4_000_000, help="reads of the end-to-end leg: FASTQ file -> count vectors (0 = skip)")
    args = ap.parse_args()

    import torch
    from mirge_amd import capi, shard, synth

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a one-GPU box only: MIRGE_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and swaps RCCL (which refuses two
    # ranks on one device) for gloo, so the N > 1 code path can be run where there is a single card
    rehearsal = os.environ.get("MIRGE_BENCH_ONE_DEVICE") == "1"
    if rehearsal:
        local = 0
        if args.merge_uniques:
            sys.exit("MIRGE_BENCH_ONE_DEVICE rehearses the all-reduce path only: gloo has no all-to-all over device buffers")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    prof = getattr(synth.Profile, args.profile)()
    mix = synth.Mixture.default() if args.mixture == "default" else synth.Mixture.isomir_rich()
    libs, _ = synth.make_libraries(prof, synth.BASE_SEED)
    ctx = capi.Context(None, 1, local)
    t0 = time.perf_counter()
    for i, nm in enumerate(synth.LIB_NAMES):
        ctx.load_library(i, libs[nm].names, libs[nm].text, libs[nm].offs)
    t_index = time.perf_counter() - t0

    # ---- synthetic reads of this rank's shard -> ASCII matrix in HBM -> 2-bit keys in HBM
    N = args.reads
    W = 48
    k16 = torch.empty((N, 2), dtype=torch.int64, device="cuda")
    k48_parts = []
    n16 = 0
    sample_seqs = sample_lens = None
    e2e_parts = []
    t16 = torch.empty((args.chunk, 2), dtype=torch.int64, device="cuda")
    t48 = torch.empty((args.chunk, 6), dtype=torch.int64, device="cuda")
    h2d_bytes = 0
    for c, b in enumerate(range(0, N, args.chunk)):
        m = min(args.chunk, N - b)
        seqs, lens = synth.make_reads(libs, m, synth.BASE_SEED + 1000 * (rank + 1) + c, mix, W=W)
        if c == 0:
            sample_seqs, sample_lens = seqs[:args.cpu_sample].copy(), lens[:args.cpu_sample].copy()
        if rank == 0 and world == 1 and b < args.e2e_reads:
            e2e_parts.append((seqs[:args.e2e_reads - b], lens[:args.e2e_reads - b]))
        d_seqs = torch.from_numpy(seqs).cuda(); d_lens = torch.from_numpy(lens).cuda()
        h2d_bytes += seqs.nbytes + lens.nbytes
        torch.cuda.synchronize()
        a, w = ctx.pack_dev(d_seqs.data_ptr(), W, d_lens.data_ptr(), m, t16.data_ptr(), t48.data_ptr())
        k16[n16:n16 + a].copy_(t16[:a]); n16 += a
        if w:
            k48_parts.append(t48[:w].clone())
        torch.cuda.synchronize()
        if rank == 0 and c % 20 == 19:
            print("[bench] %d / %d reads generated and packed" % (b + m, N), file=sys.stderr, flush=True)
    k48 = torch.cat(k48_parts) if k48_parts else torch.empty((1, 6), dtype=torch.int64, device="cuda")
    n48 = N - n16
    del t16, t48, k48_parts
    torch.cuda.synchronize()

    # measured device copy bandwidth of this GPU in this run (report against it as well as the 8 TB/s spec)
    src = torch.empty(1 << 28, dtype=torch.int32, device="cuda").fill_(1); dst = torch.empty_like(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dst.copy_(src); torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        dst.copy_(src)
    e1.record(); torch.cuda.synchronize()
    copy_gbs = 5 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del src, dst

    flat_dev = None

    def step():
        nonlocal flat_dev
        ctx.reset()
        ctx.collapse_packed_dev(k16.data_ptr(), n16, k48.data_ptr(), n48, 0)
        if world > 1 and args.merge_uniques:
            shard.merge_uniques(ctx, dist, torch.device("cuda", local))
        ctx.annotate()
        if world > 1:
            ctx.set_counts_flat(shard.allreduce_flat(ctx.counts_flat(), dist, torch.device("cuda", local)))

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    dev_ms = []
    ktimes = []
    for _ in range(args.steps):
        step()
        tm = ctx.last_timing(); dev_ms.append(tm["collapse_ms"] + tm["annotate_ms"] + tm["counts_ms"])
        ktimes.append(ctx.kernel_times())
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- algorithmic bytes: B = 16N [+4N uid, not written here] + 20U + 4nnz + sum_r 20 U_r + sum_r |lib_r|/4 + 16 M S
    U = ctx.n_unique(); nnz = U
    stats = [ctx.round_stats(r + 1) for r in range(5)]
    libT = [len(libs[nm].text) + libs[nm].n for nm in synth.LIB_NAMES]
    round_lib = [0, 1, 2, 3, 0]
    M = libs["mirna"].n
    b_alg = 16 * N + 20 * U + 4 * nnz + sum(20 * s[0] for s in stats) + sum(libT[l] // 4 for l in round_lib) + 16 * M
    # per-kernel: average duration over the timed steps, algorithmic bytes as the library reports them
    names = list(ktimes[-1].keys())
    kavg = {k: (float(np.mean([kt[k][0] for kt in ktimes if k in kt])), int(ktimes[-1][k][1])) for k in names if k != "pack"}
    dom = max(kavg, key=lambda k: kavg[k][0])
    dom_ms, dom_bytes = kavg[dom]
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("reads") == N and tj.get("profile") == args.profile:
                traffic = tj.get("kernels", {}).get(dom)
        except Exception:
            traffic = None
    dev_ms_avg = float(np.mean(dev_ms))

    out = {
        "metric": "reads/sec annotated (collapse+library cascade)",
        "value": world * N * args.steps / elapsed,
        "unit": "reads/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64 (2-bit packed nucleotides, integer XOR/popcount)", "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[1]: %s-profile synthetic seqLibs, %d adapter-trimmed reads per GPU, "
                               "5-round cascade (exact miRNA, hairpin 1mm, other 1mm, mRNA, isomiR 2mm)" % (args.profile, N),
                   "reads_per_gpu": N, "uniques": int(U), "samples": 1, "mixture": args.mixture,
                   "lib_nt": {nm: int(len(libs[nm].text)) for nm in synth.LIB_NAMES},
                   "round_uniques": [int(s[0]) for s in stats], "round_aligned": [int(s[1]) for s in stats],
                   "parallelism": "shard%d%s" % (world, "+merge_uniques" if (world > 1 and args.merge_uniques) else "")},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "copy_measured": copy_gbs, "frac_of_copy": achieved / copy_gbs,
                     "kernel_ms": dom_ms, "kernel_bytes_alg": dom_bytes},
        "roofline_path": {"bytes_alg_per_read": b_alg / N, "device_ms": dev_ms_avg,
                          "achieved": b_alg / (dev_ms_avg * 1e-3) / 1e9 if dev_ms_avg > 0 else 0.0,
                          "frac": (b_alg / (dev_ms_avg * 1e-3) / 1e9) / HBM_PEAK_GBS if dev_ms_avg > 0 else 0.0},
        "kernels_ms": {k: round(v[0], 4) for k, v in kavg.items()},
        "setup": {"index_build_s": round(t_index, 3), "h2d_bytes": int(h2d_bytes)},
    }

    # ---- CPU baseline: the oracle (a port; the reference's bowtie/cutadapt are not installed) on a bounded sample
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        from oracle import oracle as orc
        cores = min(16, os.cpu_count() or 1)
        olibs = [orc.OracleLib((libs[nm].text, libs[nm].offs)) for nm in synth.LIB_NAMES]   # index build: not timed
        t0 = time.perf_counter()
        res = orc.run_path(olibs, [(sample_seqs, sample_lens)], use_index=True, nthreads=cores, maxlen=capi.MAX_READ_LEN)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": len(sample_lens) / dt, "unit": "reads/s", "cores": cores, "kind": "port",
                               "sample": "first %d reads of the same workload, same libraries; collapse + 5-round cascade + "
                                         "count vectors by oracle/mirge_oracle.c (k-mer-indexed matcher, OpenMP over uniques); "
                                         "index build excluded; %.1f s" % (len(sample_lens), dt)}
        # and the GPU path must agree with it on that sample
        ctx.reset()
        ctx.collapse_seqs(sample_seqs, 0, sample_lens)
        ctx.annotate()
        cv = ctx.counts()
        ok = all(np.array_equal(cv[k], res[k]) for k in ("quant", "iscan", "cat", "tuniq", "lenhist"))
        out["parity_on_cpu_sample"] = bool(ok)
    # ---- end-to-end leg (reported separately, never `value`): trimmed FASTQ file (page cache) -> HBM ->
    #      record splitting + packing on the device -> collapse -> cascade -> count vectors on the host
    if rank == 0 and world == 1 and e2e_parts:
        import tempfile
        path = os.path.join(tempfile.gettempdir(), "mirge_bench_%d.fastq" % os.getpid())
        es = np.concatenate([p[0] for p in e2e_parts]); el = np.concatenate([p[1] for p in e2e_parts])
        nbytes = synth.write_fastq_fast(path, es, el)
        best = None
        for _ in range(3):
            ctx.reset()
            t0 = time.perf_counter()
            nr = ctx.collapse_fastq(path, 0)
            t1 = time.perf_counter()
            ctx.annotate(); ctx.counts()
            t2 = time.perf_counter()
            kt = ctx.kernel_times()
            cur = {"reads": int(nr), "fastq_bytes": int(nbytes), "seconds": t2 - t0, "reads_per_s": nr / (t2 - t0),
                   "file_to_keys_s": t1 - t0, "file_GBps": nbytes / (t1 - t0) / 1e9,
                   "fastq_kernels_ms": {k: round(v[0], 4) for k, v in kt.items() if k.startswith("fastq")},
                   "note": "file in the page cache; reader threads -> pinned ring -> DMA -> k_fq_count/index/pack -> collapse -> "
                           "cascade -> counts; best of 3"}
            if best is None or cur["seconds"] < best["seconds"]:
                best = cur
        os.unlink(path)
        assert nr == len(el)
        out["e2e"] = best
        # the same from RAW reads: 50-nt reads with the 3' adapter still on, trimming on the device
        nbytes = synth.write_raw_fastq_fast(path, es, el, synth.BASE_SEED + 77)
        best = None
        for _ in range(3):
            ctx.reset()
            t0 = time.perf_counter()
            tot, kept, _ = ctx.collapse_raw_fastq(path, 0, "illumina")
            t1 = time.perf_counter()
            ctx.annotate(); ctx.counts()
            t2 = time.perf_counter()
            kt = ctx.kernel_times()
            cur = {"reads": int(tot), "kept": int(kept), "fastq_bytes": int(nbytes), "seconds": t2 - t0, "reads_per_s": tot / (t2 - t0),
                   "file_to_keys_s": t1 - t0, "fastq_kernels_ms": {k: round(v[0], 4) for k, v in kt.items() if k.startswith("fastq")},
                   "note": "raw 50-nt reads, adapter on: quality trim + adapter alignment + length filter on the device, then as e2e; best of 3"}
            if best is None or cur["seconds"] < best["seconds"]:
                best = cur
        os.unlink(path)
        out["e2e_raw"] = best
    # size-independent invariants of the full-size run (every read lands in exactly one category / length bin)
    if world == 1:
        print("[bench] timed steps done; full-size invariants", file=sys.stderr, flush=True)
        ctx.reset()
        ctx.collapse_packed_dev(k16.data_ptr(), n16, k48.data_ptr(), n48, 0)
        ctx.annotate()
        cvf = ctx.counts()
        uid, smp, cnt = ctx.export_counts()
        out["invariants_full_size"] = bool(int(cvf["cat"].sum()) == N and int(cvf["lenhist"].sum()) == N and
                                           int(cnt.sum()) == N and int(cvf["tuniq"][0]) == len(cnt) == ctx.n_unique() and
                                           int(cvf["quant"].sum()) == int(cvf["cat"][0].sum()) and
                                           sum(s[1] for s in stats) + int((ctx.export_uniques(with_seqs=False)[2] == 0).sum()) == ctx.n_unique())
    if rank == 0:
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
