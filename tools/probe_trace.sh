#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of a short bench run per environment -> average duration of every launch of
# the probe kernel by template arguments <round, schemes, key words, mode>.  usage: tools/probe_trace.sh "<ENV=VAL ...>" ...
export TMPDIR=/tmp
i=0
for cfg in "$@"; do
  i=$((i+1)); d=gpurun_out/prt_$i; rm -rf $d
  for kv in $cfg; do export "$kv"; done
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 bench.py --steps 4 --warmup 1 --cpu-sample 0 --e2e-reads 0 --realistic-reads 0 --perl-reads 0 --repeats-reads 0 $PT_ARGS > $d.json 2> $d.err || { echo "$cfg: failed"; tail -3 $d.err; exit 1; }
  python3 - "$cfg" $(find $d -name '*kernel_stats.csv') <<'PY'
import csv, sys, re
out = []
for r in csv.DictReader(open(sys.argv[2])):
    m = re.search(r"k_(probe_round<[\d, ]+>|match_round<[\d, ]+>)", r["Name"])
    if m and float(r["AverageNs"]) > 20000: out.append("%s %.0f" % (m.group(1).replace("probe_round", "P").replace("match_round", "M").replace(" ", ""), float(r["AverageNs"]) / 1e3))
print("%-24s %s" % (sys.argv[1], " | ".join(sorted(out))))
PY
  for kv in $cfg; do unset "${kv%%=*}"; done
done
