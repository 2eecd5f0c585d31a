#!/bin/bash
# quick A/B on the GPU box: bench_quick.sh "<ENV=VAL ...>" ...   -> one line per configuration (BQ_ARGS: extra bench.py arguments)
for cfg in "$@"; do
  env $cfg timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --cpu-sample 0 --e2e-reads 0 --realistic-reads 0 --perl-reads 0 --repeats-reads 0 $BQ_ARGS > gpurun_out/bq.json 2>/dev/null
  python - "$cfg" <<'PY'
import json, sys
d = json.load(open("gpurun_out/bq.json"))
k = d["kernels_ms"]
print("%-34s %6.2f ms  %s" % (sys.argv[1], d["ms_per_step"], " ".join("%s=%.2f" % (n.replace("match_round", "R").replace("collapse_", "c_"), v) for n, v in k.items() if v >= 0.05)))
PY
done
