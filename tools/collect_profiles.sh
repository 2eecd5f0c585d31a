#!/bin/bash
# Runs on the GPU box (gpurun): kernel trace + the two HBM-traffic PMC passes + the default bench line.
# usage: tools/collect_profiles.sh <tag>     -> gpurun_out/prof_<tag>/...  (copy what is to be kept into profiles/)
set -e
tag=${1:-run}; root=$PWD; out=$root/gpurun_out/prof_$tag; mkdir -p $out
export TMPDIR=/tmp
cd $root
B="python3 bench.py --steps 5 --warmup 2 --cpu-sample 0 --e2e-reads 0 --realistic-reads 0 --perl-reads 0 --repeats-reads 0"   # every launch in these passes is a full-size (20M-read) one
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- $B > $out/bench_under_rocprof.json
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- $B > /dev/null
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- $B > /dev/null
python3 tools/pmc_traffic.py $(find $out/fetch -name '*counter_collection.csv') $(find $out/write -name '*counter_collection.csv') 5 20000000 human $out/pmc_traffic.json 8      # (13 launches of every kernel per pass: 2 warm-up + 5 plain + 5 with events + 1 invariants step; the last 5 are averaged)
cp $out/pmc_traffic.json profiles/pmc_traffic.json
cp $(find $out/trace -name '*kernel_stats.csv') $out/kernel_stats.csv
timeout -k 10 400 python3 bench.py > $out/bench.json
cat $out/bench.json
