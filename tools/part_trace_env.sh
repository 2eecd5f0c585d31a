#!/bin/bash
# as part_trace.sh, one library, several environments: tools/part_trace_env.sh <lib.so> "<ENV=VAL ...>" ...
export TMPDIR=/tmp
lib=$1; shift
i=0
for cfg in "$@"; do
  i=$((i+1)); d=gpurun_out/pte_$i; rm -rf $d
  export MIRGE_HIP_LIB=$lib
  for kv in $cfg; do export "$kv"; done
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 bench.py --steps 4 --warmup 1 --cpu-sample 0 --e2e-reads 0 --realistic-reads 0 --perl-reads 0 --repeats-reads 0 $PT_ARGS > $d.json 2> $d.err || { echo "$cfg: failed"; tail -3 $d.err; exit 1; }
  python3 - "$cfg" $(find $d -name '*kernel_stats.csv') $d.json <<'PY'
import csv, sys, re, json
out = []
for r in csv.DictReader(open(sys.argv[2])):
    m = re.search(r"k_part_(pass<[\d, ]+>|fold)", r["Name"])
    if m: out.append("%s %.1f us (min %.1f, n=%s)" % (m.group(1), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, r["Calls"]))
d = json.load(open(sys.argv[3]))
print("%-28s %s | inv=%s" % (sys.argv[1], " | ".join(sorted(out)), d.get("invariants_full_size")))
PY
  for kv in $cfg; do unset "${kv%%=*}"; done
done
