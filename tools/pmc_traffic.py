#!/usr/bin/env python3
"""Reduce two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py` into profiles/pmc_traffic.json.

usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <steps> <reads> <profile> <out.json> [warmup]

Per kernel symbol: its launches per step = dispatches / (steps + warmup); the launches of the last <steps> steps are
summed and divided by <steps> (the timed steps come last), and symbols that make up one named kernel time of
mirge_kernel_times (a round = its probe kernel + the generic kernel over what that hands on; the partitioned collapse =
both split passes) are added up.  rocprofv3 reports both
counters in KiB; traffic = 2*FETCH + WRITE bytes (MI355X_MICROARCH.md, "HBM [CDNA4]": on gfx950 FETCH_SIZE
tallies each 128-B request as 64 B -- tools/calib_fetch.sh measured the same factor for scattered 32-byte loads;
WRITE_SIZE is exact for streaming stores).  Kernel keys follow the
names `mirge_kernel_times` reports, so bench.py can look its dominant kernel up.
"""
import csv
import json
import re
import sys
from collections import defaultdict


def key_of(name):
    m = re.search(r"k_match_round<(\d+), (\d+)>", name)
    if m:
        return "match_round%d%s" % (int(m.group(2)) + 1, "" if m.group(1) == "2" else "_48")
    m = re.search(r"k_(probe|lds)_round<(\d+), (\d+)(?:, (\d+))?", name)
    if m:
        return "match_round%d%s" % (int(m.group(2)) + 1, "_48" if m.group(4) == "6" else "")
    if "k_part_pass" in name:
        return "collapse_part16"
    if "k_part_fold" in name:
        return "collapse_fold16"
    m = re.search(r"k_(collapse_local|collapse_emit|counts)<(\d+)", name)
    if m:
        wide = m.group(2) != "2"
        if m.group(1) == "counts":
            return "counts_48" if wide else "counts"
        return m.group(1) + ("48" if wide else "16")
    m = re.search(r"(k_\w+)", name)
    return m.group(1) if m else name[:40]


def load(path, counter, steps, warmup):
    per = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            per[r["Kernel_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    out = defaultdict(float)
    for name, v in per.items():
        per_step = max(1, round(len(v) / (steps + warmup)))
        take = sorted(v)[-steps * per_step:]
        out[key_of(name)] += sum(x[1] for x in take) / max(1, len(take) / per_step) * 1024.0
    return dict(out)


def main(a):
    fetch_csv, write_csv, steps, reads, profile, out = a[0], a[1], int(a[2]), int(a[3]), a[4], a[5]
    warmup = int(a[6]) if len(a) > 6 else 2
    f = load(fetch_csv, "FETCH_SIZE", steps, warmup)
    w = load(write_csv, "WRITE_SIZE", steps, warmup)
    ks = sorted(set(f) | set(w))
    import hashlib, os
    so = os.environ.get("MIRGE_HIP_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mirge_amd", "libmirge_hip.so")
    sha = hashlib.sha256(open(so, "rb").read()).hexdigest()[:16] if os.path.exists(so) else None
    json.dump({"unit": "bytes per launch", "reads": reads, "profile": profile, "lib_sha16": sha,
               "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (KiB * 1024), mean of the last "
                       "%d steps of each kernel; kernels = 2*FETCH + WRITE per MI355X_MICROARCH.md (gfx950 FETCH_SIZE "
                       "counts 64 B per 128-B request); calibrated for scattered loads by tools/calib_fetch.sh "
                       "(profiles/r03/calib_fetch_size.txt: a random 32-byte load from a table beyond the L2 is reported as "
                       "64.0 B, x2 = the 128-B line it moves); Infinity-Cache hits are counted (fabric-side counters)" % steps,
               "kernels": {k: int(2 * f.get(k, 0) + w.get(k, 0)) for k in ks},
               "raw_fetch": {k: int(f.get(k, 0)) for k in ks}, "raw_write": {k: int(w.get(k, 0)) for k in ks}},
              open(out, "w"), indent=1, sort_keys=True)
    for k in ks:
        print("%-22s %8.1f MB" % (k, (2 * f.get(k, 0) + w.get(k, 0)) / 1e6))


if __name__ == "__main__":
    main(sys.argv[1:])
