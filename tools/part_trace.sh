#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of a short bench run per library build -> average duration of the collapse
# kernels (k_part_pass<first pass>, k_part_pass<256>, k_part_fold) by name.  usage: tools/part_trace.sh <lib.so> ...   (PT_ARGS: extra bench.py arguments)
export TMPDIR=/tmp
for lib in "$@"; do
  d=gpurun_out/pt_$(basename $lib .so); rm -rf $d
  MIRGE_HIP_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 bench.py --steps 4 --warmup 1 --cpu-sample 0 --e2e-reads 0 --realistic-reads 0 --perl-reads 0 --repeats-reads 0 $PT_ARGS > /dev/null 2> $d.err || { echo "$lib: failed"; tail -3 $d.err; exit 1; }
  python3 - $lib $(find $d -name '*kernel_stats.csv') <<'PY'
import csv, sys, re
out = []
for r in csv.DictReader(open(sys.argv[2])):
    m = re.search(r"k_part_(pass<[\d, ]+>|fold)", r["Name"])
    if m: out.append("%s %.1f us (min %.1f, n=%s)" % (m.group(1), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, r["Calls"]))
print("%-28s %s" % (sys.argv[1], " | ".join(sorted(out))))
PY
done
