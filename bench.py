#!/usr/bin/env python3
"""bench.py -- reads/s through collapse + 5-round library cascade + count vectors on MI355X.

One "step" = one pass of the whole hot path over one batch of synthetic adapter-trimmed reads whose 2-bit keys are
already resident in HBM: mirge_reset -> mirge_collapse_packed_dev -> [N > 1: merge of the unique tables
by key owner, one all-to-all] -> mirge_annotate (cascade + count vectors) -> [N > 1: ONE sum all-reduce of the flat
count buffer].

Workload.  N = 1: BASELINE.json configs[1] -- human-profile synthetic seqLibs, 20M reads.  N > 1: configs[2] -- ONE read
set of 100M reads, rank r takes shard_bounds(100M, r, N) (strong scaling: total work fixed; `--weak` gives every rank
its own 20M-read shard instead).  value = reads of the whole job / max-over-ranks time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
REF_PUBLISHED_READS_PER_S = 92000   # the reference's only published run: 1,436,249 trimmed reads through the cascade in 15.55 s


def gen_keys(torch, synth, capi, ctx, libs, lo, hi, chunk, mix, W, rank, keep_first=0, e2e_reads=0, label="reads"):
    """Reads [lo, hi) of the run's read set (chunk c of it = make_reads(seed BASE + 1000 + c): the same on every rank)
    -> ASCII matrix in HBM -> 2-bit keys in HBM.  Returns (k16, n16, k48, n48, first reads kept on the host, e2e parts, bytes)."""
    n = hi - lo
    k16 = torch.empty((max(n, 1), 2), dtype=torch.int64, device="cuda")
    t16 = torch.empty((chunk, 2), dtype=torch.int64, device="cuda")
    t48 = torch.empty((chunk, 6), dtype=torch.int64, device="cuda")
    k48_parts, n16, h2d, e2e = [], 0, 0, []
    firsts, kept = [], 0
    done = 0
    for c in range(lo // chunk, (hi + chunk - 1) // chunk):
        seqs, lens = synth.make_reads(libs, chunk, synth.BASE_SEED + 1000 + c, mix, W=W)
        a, b = max(lo, c * chunk) - c * chunk, min(hi, (c + 1) * chunk) - c * chunk
        seqs, lens = seqs[a:b], lens[a:b]
        m = len(lens)
        if kept < keep_first:
            firsts.append((seqs[:keep_first - kept].copy(), lens[:keep_first - kept].copy())); kept += len(firsts[-1][1])
        if done < e2e_reads:
            e2e.append((seqs[:e2e_reads - done], lens[:e2e_reads - done]))
        d_seqs = torch.from_numpy(np.ascontiguousarray(seqs)).cuda(); d_lens = torch.from_numpy(np.ascontiguousarray(lens)).cuda()
        h2d += seqs.nbytes + lens.nbytes
        torch.cuda.synchronize()
        x, w = ctx.pack_dev(d_seqs.data_ptr(), W, d_lens.data_ptr(), m, t16.data_ptr(), t48.data_ptr())
        k16[n16:n16 + x].copy_(t16[:x]); n16 += x
        if w:
            k48_parts.append(t48[:w].clone())
        torch.cuda.synchronize()
        done += m
        if rank == 0 and (done // chunk) % 20 == 0:
            print("[bench] %d / %d %s generated and packed" % (done, n, label), file=sys.stderr, flush=True)
    k48 = torch.cat(k48_parts) if k48_parts else torch.empty((1, 6), dtype=torch.int64, device="cuda")
    first = (np.concatenate([f[0] for f in firsts]), np.concatenate([f[1] for f in firsts])) if firsts else None
    return k16, n16, k48, n - n16, first, e2e, h2d


def tool_version(exe):
    p = shutil.which(exe)
    if not p:
        return None
    try:
        out = subprocess.run([p, "--version"], capture_output=True, text=True, timeout=20).stdout
        return (p, out.strip().split("\n")[0])
    except Exception:
        return (p, "?")


def reference_cpu_leg(synth, libs, seqs, lens, cores):
    """BASELINE.md step 1 / SURVEY.md 8d: if bowtie (1.x) is on PATH, time the reference's own five command lines
    (miRge.pl:333-339) on the sample, index built once with bowtie-build.  Returns a cpu_baseline dict or None."""
    bt = tool_version("bowtie"); btb = shutil.which("bowtie-build")
    if not bt or not btb:
        return None
    import tempfile
    from collections import Counter
    d = tempfile.mkdtemp(prefix="mirge_ref_")
    try:
        for nm in synth.LIB_NAMES:
            fa = os.path.join(d, nm + ".fa")
            with open(fa, "w") as f:
                L = libs[nm]
                for i, name in enumerate(L.names):
                    f.write(">%s\n%s\n" % (name, L.text[L.offs[i]:L.offs[i + 1]].tobytes().decode()))
            subprocess.run([btb, "-q", fa, os.path.join(d, nm)], check=True, capture_output=True, timeout=3600)
        reads = synth.reads_to_list(seqs, lens)
        t0 = time.perf_counter()
        todo = list(Counter(reads))                          # quantReads: the hash keys
        rounds = [("mirna", lambda s: len(s) < 26, ["-n", "0"]), ("hairpin", lambda s: len(s) > 25, ["-n", "1"]),
                  ("other", lambda s: True, ["-n", "1"]), ("mrna", lambda s: True, ["-n", "0"]),
                  ("mirna", lambda s: True, ["-l", "15", "-5", "1", "-3", "2", "-n", "2"])]
        for lib, ok, opts in rounds:
            fa = os.path.join(d, "SeqToAnnot.fasta")
            with open(fa, "w") as f:
                for s in todo:
                    if ok(s):
                        f.write(">%s\n%s\n" % (s, s))
            out = subprocess.run([bt[0], "-a", "--best", "--strata", "--threads", str(cores), os.path.join(d, lib)] + opts + ["-f", fa],
                                 capture_output=True, text=True, timeout=3600, check=True).stdout
            hit = {line.split("\t", 1)[0] for line in out.split("\n") if line}
            todo = [s for s in todo if s not in hit]
        dt = time.perf_counter() - t0
        return {"value": len(lens) / dt, "unit": "reads/s", "cores": cores, "kind": "reference",
                "sample": "first %d reads of the same workload: Perl-style collapse + the five bowtie command lines of miRge.pl:333-339 "
                          "(%s, --threads %d), indexes built once with bowtie-build (not timed); %.1f s" % (len(lens), bt[1], cores, dt)}
    except Exception as e:                                    # a tool that is there but does not run: say so, keep the port
        print("[bench] reference CPU leg failed: %r" % (e,), file=sys.stderr)
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=20_000_000, help="reads per GPU (N = 1, and N > 1 with --weak)")
    ap.add_argument("--total-reads", type=int, default=100_000_000, help="N > 1: reads of the one read set the ranks share")
    ap.add_argument("--weak", action="store_true", help="N > 1: independent --reads-sized shards per rank instead of one shared read set")
    ap.add_argument("--no-merge-uniques", action="store_true",
                    help="N > 1: skip the all-to-all of the unique tables by key owner (count vectors stay exact; trimmed_uniq, "
                         "annotStats and the per-unique tables then count a sequence once per shard that holds it)")
    ap.add_argument("--profile", default="human", choices=["human", "mouse", "medium", "small", "tiny"])
    ap.add_argument("--mixture", default="default", choices=["default", "isomir_rich", "realistic"])
    ap.add_argument("--cpu-sample", type=int, default=20_000_000, help="leading reads of the same workload timed on the CPU, by default all of them (0 = skip)")
    ap.add_argument("--chunk", type=int, default=1_000_000)
    ap.add_argument("--e2e-reads", type=int, default=4_000_000, help="reads of the end-to-end legs: FASTQ file -> count vectors (0 = skip)")
    ap.add_argument("--realistic-reads", type=int, default=20_000_000, help="reads of the extra line at the reference's published redundancy (0 = skip)")
    ap.add_argument("--repeats-reads", type=int, default=10_000_000, help="reads of the extra line against repeat-rich libraries (0 = skip)")
    ap.add_argument("--perl-reads", type=int, default=1_000_000, help="raw reads of the e2e_perl leg, BASELINE.json configs[0] (0 = skip)")
    args = ap.parse_args()

    import torch
    from mirge_amd import capi, shard, synth

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a one-GPU box only: MIRGE_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and swaps RCCL (which refuses two
    # ranks on one device) for gloo (the all-to-all of merge_uniques is then staged through host tensors)
    rehearsal = os.environ.get("MIRGE_BENCH_ONE_DEVICE") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    merge = world > 1 and not args.no_merge_uniques
    solo = rank == 0 and world == 1

    prof = getattr(synth.Profile, args.profile)()
    mix = getattr(synth.Mixture, args.mixture)()
    libs, merges = synth.make_libraries(prof, synth.BASE_SEED)
    ctx = capi.Context(None, 1, local)
    t0 = time.perf_counter()
    for i, nm in enumerate(synth.LIB_NAMES):
        ctx.load_library(i, libs[nm].names, libs[nm].text, libs[nm].offs)
    t_index = time.perf_counter() - t0

    # ---- this rank's reads
    W = 48
    if world == 1:
        total, lo, hi = args.reads, 0, args.reads
    elif args.weak:
        total, lo, hi = args.reads * world, rank * args.reads, (rank + 1) * args.reads
    else:
        total = args.total_reads
        lo, hi = shard.shard_bounds(total, rank, world)
    N = hi - lo
    k16, n16, k48, n48, first, e2e_parts, h2d_bytes = gen_keys(torch, synth, capi, ctx, libs, lo, hi, args.chunk, mix, W, rank,
                                                               keep_first=args.cpu_sample if solo else 0,
                                                               e2e_reads=max(args.e2e_reads, args.perl_reads) if solo else 0)
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

    def step(c=ctx, a=k16, na=n16, b=k48, nb=n48):
        c.reset()
        if world == 1:
            # one call: the cascade is queued behind the collapse, the host does not wait in between to learn the unique count
            c.collapse_annotate_packed_dev(a.data_ptr(), na, b.data_ptr(), nb, 0)
            return
        c.collapse_packed_dev(a.data_ptr(), na, b.data_ptr(), nb, 0)
        if merge:
            shard.merge_uniques(c, dist, dev)
        c.annotate()
        shard.allreduce_counts(c, dist, dev)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_plain(n_steps, **kw):
        """The timed region of `value`: K steps between two fences, per-kernel event records switched off (each named
        kernel time is two markers on the stream; ~40 of them per step sit between the kernels)."""
        ctx.set_kernel_timing(False)
        fence()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step(**kw)
        fence()
        el = time.perf_counter() - t0
        ctx.set_kernel_timing(True)
        return el

    def timed(n_steps, **kw):
        """The same steps once more WITH the per-kernel HIP events (on the library's own streams): the kernel breakdown
        and the roofline figures come from here, never `value`."""
        fence()
        t0 = time.perf_counter()
        dev_ms, kts = [], []
        raw = []
        for _ in range(n_steps):
            step(**kw)
            tm = ctx.last_timing(); dev_ms.append(tm["collapse_ms"] + tm["annotate_ms"] + tm["counts_ms"])
            n, names, ms, by = ctx.kernel_times_raw()            # (the events of this step are re-recorded by the next one)
            raw.append((n, list(ms[:n]), list(by[:n]), [names[i] for i in range(n)] if not raw else None))
        fence()
        el = time.perf_counter() - t0
        nm = [x.decode() for x in raw[0][3]] if raw else []
        for n, ms, by, names in raw:
            if names is not None:
                nm = [x.decode() for x in names]
            kts.append({nm[i]: (ms[i], by[i]) for i in range(min(n, len(nm)))})
        return el, dev_ms, kts

    for _ in range(args.warmup):
        step()
    elapsed = timed_plain(args.steps)
    elapsed_events, dev_ms, ktimes = timed(args.steps)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if not rehearsal else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- algorithmic bytes: B = 16N [+4N uid, not written here] + 20U + 4nnz + sum_r 20 U_r + sum_r |lib_r|/4 + 16 M S
    # (after merge + all-reduce U and the round statistics are those of the whole job)
    stats = [ctx.round_stats(r + 1) for r in range(5)]
    if merge:
        ut = torch.tensor([ctx.n_unique()], dtype=torch.int64, device="cuda" if not rehearsal else "cpu")
        dist.all_reduce(ut)
        U = int(ut.item())
    else:
        U = ctx.n_unique()
    nnz = U
    libT = [len(libs[nm].text) + libs[nm].n for nm in synth.LIB_NAMES]
    round_lib = [0, 1, 2, 3, 0]
    M = libs["mirna"].n
    n_job = total if (world == 1 or merge) else N
    b_alg = 16 * n_job + 20 * U + 4 * nnz + sum(20 * s[0] for s in stats) + sum(libT[l] // 4 for l in round_lib) + 16 * M
    # per kernel: average / median duration over the timed steps, algorithmic bytes as the library reports them
    names = [k for k in ktimes[-1].keys() if k != "pack"]
    kavg = {k: (float(np.mean([kt[k][0] for kt in ktimes if k in kt])), int(ktimes[-1][k][1]),
                float(np.median([kt[k][0] for kt in ktimes if k in kt]))) for k in names}
    dom = max(kavg, key=lambda k: kavg[k][0])
    dom_ms, dom_bytes, _ = kavg[dom]
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    # HBM-side traffic of the dominant kernel: PMC counters cannot be read from inside this process, so the figure is the
    # builder's rocprofv3 pass over this same command (committed file), and it is reported only when that file says it was
    # taken from the same workload AND the same build of the library (sha256 of the .so); else null
    traffic, traffic_source, requests = None, None, None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            import hashlib
            tj = json.load(open(tpath))
            so = os.environ.get("MIRGE_HIP_LIB") or os.path.join(ROOT, "mirge_amd", "libmirge_hip.so")
            sha = hashlib.sha256(open(so, "rb").read()).hexdigest()[:16]
            if tj.get("reads") == N and tj.get("profile") == args.profile and tj.get("lib_sha16") == sha:
                traffic = tj.get("kernels", {}).get(dom)
                raw = tj.get("raw_fetch", {}).get(dom)
                if raw and dom_ms > 0:
                    # the same counters read as REQUESTS past the L2 (FETCH_SIZE counts 64 B for one that goes to HBM): the bound of
                    # the index kernels is how many of those the memory system serves per second, not bytes
                    requests = {"past_l2": int(raw // 64), "rate_G_per_s": raw / 64 / (dom_ms * 1e-3) / 1e9,
                                "machine_random_fetch_G_per_s": {"512 MB table": 55.3, "8 GB table": 41.2, "Infinity-Cache-resident (8 MB)": 118.4},
                                "source": "raw FETCH_SIZE of the same PMC pass / 64 B / the kernel's HIP-event time; machine rates: profiles/r02/micro_random_lines.txt"}
                traffic_source = ("profiles/pmc_traffic.json: the builder's rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command with "
                                  "this build of the library (lib_sha16 %s), not counters of this run; 2*FETCH + WRITE KiB, i.e. bytes past "
                                  "the L2 -- Infinity Cache hits included" % sha)
            else:
                traffic_source = "profiles/pmc_traffic.json is of another build or workload: not reported"
        except Exception:
            traffic, traffic_source, requests = None, None, None
    dev_ms_avg, dev_ms_med = float(np.mean(dev_ms)), float(np.median(dev_ms))

    if world == 1:
        workload = ("BASELINE.json configs[1]: %s-profile synthetic seqLibs, %d adapter-trimmed reads on one GPU, 5-round cascade "
                    "(exact miRNA, hairpin 1mm, other 1mm, mRNA, isomiR 2mm)" % (args.profile, N))
    elif args.weak:
        workload = "weak scaling: %s-profile synthetic seqLibs, an independent %d-read shard per GPU, 5-round cascade" % (args.profile, args.reads)
    else:
        workload = ("BASELINE.json configs[2]: %s-profile synthetic seqLibs, ONE set of %d adapter-trimmed reads sharded over %d GPUs "
                    "(contiguous chunks), 5-round cascade, RCCL count all-reduce" % (args.profile, total, world))
    out = {
        "metric": "reads/sec annotated (collapse+library cascade)",
        "value": (N * world if (world > 1 and args.weak) else total) * args.steps / elapsed,
        "unit": "reads/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak" if (world == 1 or args.weak) else "strong", "vs_baseline": None,
        "dtype": "u64 (2-bit packed nucleotides, integer XOR/popcount)", "data": "synthetic",
        "config": {"workload": workload, "reads_total": int(N * world if (world > 1 and args.weak) else total), "reads_this_rank": int(N),
                   "uniques": int(U), "samples": 1, "mixture": args.mixture,
                   "lib_nt": {nm: int(len(libs[nm].text)) for nm in synth.LIB_NAMES},
                   "round_uniques": [int(s[0]) for s in stats], "round_aligned": [int(s[1]) for s in stats],
                   "parallelism": "shard%d%s%s" % (world, "+merge_uniques" if merge else "", "+gloo_one_device_rehearsal" if rehearsal and world > 1 else "")},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "requests": requests, "copy_measured": copy_gbs, "frac_of_copy": achieved / copy_gbs,
                     "kernel_ms": dom_ms, "kernel_bytes_alg": dom_bytes,
                     "note": "achieved = SURVEY.md 8d's algorithmic bytes of the kernel / its HIP-event time: an index-based matcher cannot "
                             "reach that floor.  What bounds the kernel on the machine is the rate of random 128-B line fetches past the L2 "
                             "(`traffic`, PMC) and vector issue, not streaming HBM bandwidth -- DESIGN.md section 4"},
        "roofline_path": {"bytes_alg_per_read": b_alg / n_job, "device_ms": dev_ms_avg, "device_ms_median": dev_ms_med,
                          "achieved": b_alg / world / (dev_ms_avg * 1e-3) / 1e9 if dev_ms_avg > 0 else 0.0,
                          "frac": (b_alg / world / (dev_ms_avg * 1e-3) / 1e9) / HBM_PEAK_GBS if dev_ms_avg > 0 else 0.0},
        "ms_per_step_with_kernel_events": 1e3 * elapsed_events / args.steps,
        "kernels_ms": {k: round(v[0], 4) for k, v in kavg.items()},
        "kernels_ms_median": {k: round(v[2], 4) for k, v in kavg.items()},
        "setup": {"index_build_s": round(t_index, 3), "h2d_bytes": int(h2d_bytes)},
    }

    # ---- how much of the assignment rides on the tie-order hypothesis H1: uniques / reads whose best stratum held > 1 reference
    if solo:
        _, _, rnd_, _, amb_ = ctx.export_uniques(with_seqs=False)
        uid_, _, cnt_ = ctx.export_counts()
        per = np.zeros(len(rnd_), dtype=np.int64); np.add.at(per, uid_.astype(np.int64), cnt_.astype(np.int64))
        ann = rnd_ > 0
        out["h1_dependent"] = {"uniques_annotated": int(ann.sum()), "uniques_ambiguous": int((ann & (amb_ > 0)).sum()),
                               "reads_annotated": int(per[ann].sum()), "reads_ambiguous": int(per[ann & (amb_ > 0)].sum()),
                               "note": "ambiguous = the best stratum held more than one reference, so which one miRge keeps follows hypothesis H1 "
                                       "(parity with bowtie 1.1.1 unpinned)"}

    # ---- CPU baseline on a bounded sample: the reference's own tools when the box has them, else the oracle (a port)
    if solo and args.cpu_sample > 0 and first is not None:
        from oracle import oracle as orc
        cores = min(16, os.cpu_count() or 1)                    # threads used; the box's count goes beside it (cores_on_box)
        sample_seqs, sample_lens = first
        ref = reference_cpu_leg(synth, libs, sample_seqs[:1_000_000], sample_lens[:1_000_000], cores)
        olibs = [orc.OracleLib((libs[nm].text, libs[nm].offs)) for nm in synth.LIB_NAMES]   # index build: not timed
        res = orc.run_path(olibs, [(sample_seqs, sample_lens)], use_index=True, nthreads=cores, maxlen=capi.MAX_READ_LEN, want_uniques=False)
        dt = res["c_seconds"]                                    # inside the oracle's C calls: the Python glue around them is not the baseline
        port = {"value": len(sample_lens) / dt, "unit": "reads/s", "cores": cores, "kind": "port",
                "sample": "%s %d reads of the same workload, same libraries; collapse (one thread) + 5-round cascade (OpenMP over uniques, "
                          "%d threads) + count vectors by oracle/mirge_oracle.c (k-mer-indexed matcher); index build excluded; %.1f s"
                          % ("all" if len(sample_lens) == N else "first", len(sample_lens), cores, dt)}
        out["cpu_baseline"] = ref or port
        out["cpu_baseline"]["reference_published"] = {"value": REF_PUBLISHED_READS_PER_S, "unit": "reads/s",
                                                      "note": "miRge's own example run (cascade only, hardware and threads not stated): report.html:170-175"}
        out["cpu_baseline"]["cores_on_box"] = os.cpu_count()
        out["cpu_baseline"]["tools_found"] = {t: tool_version(t) for t in ("bowtie", "bowtie-build", "cutadapt")}
        if ref:
            out["cpu_baseline_port"] = port
        # and the GPU path must agree with the checker on that sample
        ctx.reset()
        ctx.collapse_seqs(sample_seqs, 0, sample_lens)
        ctx.annotate()
        cv = ctx.counts()
        ok = all(np.array_equal(cv[k], res[k]) for k in ("quant", "iscan", "cat", "tuniq", "lenhist"))
        ok = ok and all(tuple(int(x) for x in ctx.round_stats(r + 1)[:2]) == tuple(int(x) for x in res["stats"][r]) for r in range(5))
        out["parity_on_cpu_sample"] = bool(ok)                   # count vectors, category totals, length histogram, trimmedUniq, round statistics

    # ---- the same path at the redundancy of the reference's published run (U/N ~ 0.13): an extra line, never `value`
    if solo and args.realistic_reads > 0 and args.mixture != "realistic":
        rk16, rn16, rk48, rn48, _, _, _ = gen_keys(torch, synth, capi, ctx, libs, 0, args.realistic_reads, args.chunk,
                                                   synth.Mixture.realistic(), W, rank, label="reads (realistic mixture)")
        kw = dict(a=rk16, na=rn16, b=rk48, nb=rn48)
        step(**kw); step(**kw)
        nst = max(3, args.steps // 2)
        el = timed_plain(nst, **kw)
        _, dms, kts = timed(nst, **kw)
        rs = [ctx.round_stats(r + 1) for r in range(5)]
        out["realistic"] = {"reads": int(args.realistic_reads), "uniques": int(ctx.n_unique()), "unique_fraction": ctx.n_unique() / args.realistic_reads,
                            "reads_per_s": args.realistic_reads * nst / el, "ms_per_step": 1e3 * el / nst,
                            "kernels_ms": {k: round(float(np.mean([kt[k][0] for kt in kts if k in kt])), 4) for k in kts[-1] if k != "pack"},
                            "round_uniques": [int(s[0]) for s in rs], "round_aligned": [int(s[1]) for s in rs],
                            "note": "class proportions and redundancy of docs/miRge/miRge.exampleOutput/report.html:159-175 (synth.Mixture.realistic)"}
        del rk16, rk48

    # ---- the same path against REPEAT-RICH libraries (synth.Profile.human_repeats: mRNA isoform families, tRNA-like families,
    # low-complexity stretches): an extra line, never `value`.  The default profile draws `other` and `mrna` as i.i.d. bases --
    # no repeated k-mers, the best case for direct-addressed slot tables.
    if solo and args.repeats_reads > 0 and args.profile == "human":
        rlibs, _ = synth.make_libraries(synth.Profile.human_repeats(), synth.BASE_SEED)
        rctx = capi.Context(None, 1, local)
        for i, nm in enumerate(synth.LIB_NAMES):
            rctx.load_library(i, rlibs[nm].names, rlibs[nm].text, rlibs[nm].offs)
        rk16, rn16, rk48, rn48, _, _, _ = gen_keys(torch, synth, capi, rctx, rlibs, 0, args.repeats_reads, args.chunk, mix, W, rank,
                                                   label="reads (repeat-rich libraries)")
        kw = dict(c=rctx, a=rk16, na=rn16, b=rk48, nb=rn48)
        step(**kw); step(**kw)
        nst = max(3, args.steps // 2)
        rctx.set_kernel_timing(False)
        fence(); t0 = time.perf_counter()
        for _ in range(nst):
            step(**kw)
        fence(); el = time.perf_counter() - t0
        rctx.set_kernel_timing(True)
        kts = []
        for _ in range(nst):
            step(**kw)
            kts.append(rctx.kernel_times())
        rs = [rctx.round_stats(r + 1) for r in range(5)]
        _, _, rnd_, _, amb_ = rctx.export_uniques(with_seqs=False)
        uid_, _, cnt_ = rctx.export_counts()
        per = np.zeros(len(rnd_), dtype=np.int64); np.add.at(per, uid_.astype(np.int64), cnt_.astype(np.int64))
        ann = rnd_ > 0
        out["repeats"] = {"reads": int(args.repeats_reads), "uniques": int(rctx.n_unique()), "ms_per_step": 1e3 * el / nst,
                          "reads_per_s": args.repeats_reads * nst / el,
                          "lib_nt": {nm: int(len(rlibs[nm].text)) for nm in synth.LIB_NAMES},
                          "kernels_ms": {k: round(float(np.mean([kt[k][0] for kt in kts if k in kt])), 4) for k in kts[-1] if k != "pack"},
                          "round_uniques": [int(x[0]) for x in rs], "round_aligned": [int(x[1]) for x in rs],
                          "handed_to_generic_kernel": [int(rctx.round_handed_on(r + 1)) for r in range(5)],
                          "h1_dependent": {"uniques_annotated": int(ann.sum()), "uniques_ambiguous": int((ann & (amb_ > 0)).sum()),
                                           "reads_annotated": int(per[ann].sum()), "reads_ambiguous": int(per[ann & (amb_ > 0)].sum())},
                          "note": "synth.Profile.human_repeats: mrna = isoform families (2-6 transcripts per gene sharing >= 70 % of its exons), "
                                  "other = 60 % tRNA-like families of 10-50 members differing by 1-3 nt + low-complexity stretches; same read "
                                  "mixture as the headline; handed_to_generic_kernel = uniques per round whose bucket lists the slot-table "
                                  "kernel refused (> 4096 entries) or could not queue"}
        rctx.close()
        del rk16, rk48, rlibs

    # ---- end-to-end legs (reported separately, never `value`)
    if solo and e2e_parts:
        import tempfile
        es = np.concatenate([p[0] for p in e2e_parts]); el = np.concatenate([p[1] for p in e2e_parts])
        tmpd = tempfile.mkdtemp(prefix="mirge_bench_")
        path = os.path.join(tmpd, "s.fastq")
        if args.e2e_reads > 0:
            ne = min(args.e2e_reads, len(el))
            # trimmed FASTQ file (page cache) -> HBM -> record splitting + packing on the device -> collapse -> cascade -> count vectors
            nbytes = synth.write_fastq_fast(path, es[:ne], el[:ne])
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
            assert nr == ne
            out["e2e"] = best
            # the same file gzip'd (the reference takes fastq or fastq.gz, miRge.pl:1092): one file = one serial inflate stream;
            # eight such files as a batch = eight host threads inflating ahead of the device (mirge_collapse_fastq_batch)
            import zlib
            raw = open(path, "rb").read()
            co = zlib.compressobj(1, zlib.DEFLATED, 31)
            gz = co.compress(raw) + co.flush()
            del raw
            gpaths = [os.path.join(tmpd, "g%d.fastq.gz" % i) for i in range(8)]
            for gp in gpaths:
                with open(gp, "wb") as f:
                    f.write(gz)
            ctx.reset(1)
            t0 = time.perf_counter(); nr1 = ctx.collapse_fastq(gpaths[0], 0); t_one = time.perf_counter() - t0
            ctx.reset(8)
            t0 = time.perf_counter(); totb, _, _ = ctx.collapse_fastq_batch(gpaths, list(range(8))); t_b = time.perf_counter() - t0
            ctx.annotate(); cvb = ctx.counts()
            out["e2e_gz"] = {"reads_per_file": int(nr1), "gz_bytes_per_file": len(gz), "fastq_bytes_per_file": int(nbytes),
                             "single_stream": {"seconds": t_one, "reads_per_s": nr1 / t_one, "gz_GBps": len(gz) / t_one / 1e9},
                             "batch_of_8": {"seconds": t_b, "reads_per_s": float(totb.sum()) / t_b, "gz_GBps": 8 * len(gz) / t_b / 1e9,
                                            "speedup_over_single_stream": (float(totb.sum()) / t_b) / (nr1 / t_one),
                                            "samples_identical": bool(all(int(cvb["tuniq"][i]) == int(cvb["tuniq"][0]) for i in range(8)))},
                             "host_cpus": os.cpu_count(),
                             "note": "zlib level 1 files in the page cache -> HBM -> collapse; single = mirge_collapse_fastq (one inflater thread "
                                     "feeding the pinned ring), batch = mirge_collapse_fastq_batch (one inflater thread per file, up to 16, "
                                     "files collapsed in sample order as they complete)"}
            del gz
            ctx.reset(1)
            # the same from RAW reads: 50-nt reads with the 3' adapter still on, trimming on the device
            nbytes = synth.write_raw_fastq_fast(path, es[:ne], el[:ne], synth.BASE_SEED + 77)
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
            out["e2e_raw"] = best
        # BASELINE.json configs[0] through the drop-in's own hosts: raw reads with adapter -> tables on disk, wall clock, each in
        # a process of its own (library load + index build included, as a user pays them)
        if args.perl_reads > 0:
            npl = min(args.perl_reads, len(el))
            synth.write_raw_fastq_fast(path, es[:npl], el[:npl], synth.BASE_SEED + 78)
            sp = os.path.join(tmpd, "species")
            synth.write_species_dir(sp, libs, merges)
            leg = {"reads": int(npl), "note": "BASELINE.json configs[0]: raw 50-nt reads with adapter, FASTQ file -> trimming + collapse + cascade on "
                                              "the device -> mapped/unmapped/miR.Counts/miR.RPM tables on disk; wall clock of a fresh process each"}
            env = dict(os.environ, HIP_VISIBLE_DEVICES=os.environ.get("HIP_VISIBLE_DEVICES", "0"))
            ann = os.path.join(ROOT, "tools", "mirge-annotate")
            if os.path.exists(ann):
                od = os.path.join(tmpd, "out_native")
                t0 = time.perf_counter()
                r = subprocess.run([ann, "--species", sp, "--SampleFiles", path, "--SampleNames", "s", "--adapter", "illumina", "--outputDir", od],
                                   capture_output=True, text=True, env=env, timeout=600)
                leg["native_seconds"] = time.perf_counter() - t0
                leg["native_ok"] = r.returncode == 0 and os.path.exists(os.path.join(od, "miR.Counts.csv"))
                if r.returncode:
                    leg["native_error"] = (r.stderr or r.stdout)[-300:]
            pl = os.path.join(ROOT, "perl", "t", "hotpath.pl")
            if os.path.exists(pl) and shutil.which("perl") and os.path.exists(os.path.join(ROOT, "perl", "auto", "MirgeHip", "MirgeHip.so")):
                od = os.path.join(tmpd, "out_perl")
                t0 = time.perf_counter()
                r = subprocess.run(["perl", "-I" + os.path.join(ROOT, "perl"), pl, "--species", sp, "--out", od, "--adapter", "illumina",
                                    "--names", "s", path], capture_output=True, text=True, env=env, timeout=900)
                leg["perl_seconds"] = time.perf_counter() - t0
                leg["perl_ok"] = r.returncode == 0 and os.path.exists(os.path.join(od, "miR.Counts.csv"))
                if r.returncode:
                    leg["perl_error"] = (r.stderr or r.stdout)[-300:]
                for line in (r.stderr or "").split("\n"):            # the script's own stage times
                    if line.startswith("[hotpath]"):
                        leg.setdefault("perl_stages", []).append(line[len("[hotpath] "):])
                if leg.get("native_ok") and leg.get("perl_ok"):
                    leg["tables_identical"] = all(
                        open(os.path.join(tmpd, "out_native", f)).read() == open(os.path.join(tmpd, "out_perl", f)).read()
                        for f in ("miR.Counts.csv", "miR.RPM.csv"))
            out["e2e_perl"] = leg
        shutil.rmtree(tmpd, ignore_errors=True)

    # size-independent invariants of the full-size run (every read lands in exactly one category / length bin)
    if world == 1:
        print("[bench] timed steps done; full-size invariants", file=sys.stderr, flush=True)
        ctx.reset()
        ctx.collapse_packed_dev(k16.data_ptr(), n16, k48.data_ptr(), n48, 0)
        ctx.annotate()
        cvf = ctx.counts()
        uid, smp, cnt = ctx.export_counts()
        st2 = [ctx.round_stats(r + 1) for r in range(5)]
        out["invariants_full_size"] = bool(int(cvf["cat"].sum()) == N and int(cvf["lenhist"].sum()) == N and
                                           int(cnt.sum()) == N and int(cvf["tuniq"][0]) == len(cnt) == ctx.n_unique() and
                                           int(cvf["quant"].sum()) == int(cvf["cat"][0].sum()) and
                                           sum(s[1] for s in st2) + int((ctx.export_uniques(with_seqs=False)[2] == 0).sum()) == ctx.n_unique())
    elif rank == 0 and merge:
        # after merge + all-reduce every read of the whole job sits in exactly one category on every rank's copy of the sums
        cvf = ctx.counts()
        out["invariants_full_size"] = bool(int(cvf["cat"].sum()) == total and int(cvf["lenhist"].sum()) == total and int(cvf["tuniq"][0]) == U)
    if rank == 0:
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
