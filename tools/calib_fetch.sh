#!/bin/bash
# Runs on the GPU box: FETCH_SIZE of tools/micro/random_lines (40M independent random 32-byte loads per launch, tables from
# 8 MB to 32 GB) -> bytes the counter reports per scattered 32-byte load, i.e. the factor to read FETCH_SIZE with for
# the slot-table kernels.  usage: tools/calib_fetch.sh   -> gpurun_out/calib_fetch.txt
export TMPDIR=/tmp
rm -rf gpurun_out/calib_fetch
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/calib_fetch -o p -- tools/micro/random_lines > gpurun_out/calib_fetch_run.txt 2> gpurun_out/calib_fetch.err || { echo "calibration pass failed"; tail -3 gpurun_out/calib_fetch.err; exit 1; }
python3 - <<'PY' > gpurun_out/calib_fetch.txt
import csv, glob
rows = [r for f in glob.glob("gpurun_out/calib_fetch/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE" and "k_fetch" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
sizes = [8, 64, 512, 2048, 8192, 32768]
print("tools/micro/random_lines under rocprofv3 --pmc FETCH_SIZE (MI355X): 40M independent random 32-byte loads per launch")
print("table MB | in flight | FETCH_SIZE KiB (mean of 4 launches) | reported bytes per 32-byte load | x2 (the guide's gfx950 correction)")
i = 0
for mb in sizes:
    for per in (1, 2, 4):
        v = [float(r["Counter_Value"]) for r in rows[i:i + 4]]; i += 4
        if not v: continue
        kib = sum(v) / len(v)
        print("%8d | %d | %12.0f | %6.1f | %6.1f" % (mb, per, kib, kib * 1024 / 40e6, 2 * kib * 1024 / 40e6))
PY
cat gpurun_out/calib_fetch.txt
