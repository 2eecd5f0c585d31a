#!/bin/bash
# Runs on the GPU box: FETCH_SIZE / WRITE_SIZE / SQ passes of the default bench, rounds 1 and 5 (the miRNA-library rounds).
# usage: tools/pmc_r15.sh <tag> ["ENV=VAL ..."]   -> one summary block on stdout
tag=${1:-x}; export TMPDIR=/tmp
for kv in $2; do export "$kv"; done
B="python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --e2e-reads 0 --realistic-reads 0 --perl-reads 0 --repeats-reads 0"
run() { name=$1; shift; rm -rf gpurun_out/r15_${tag}_$name; timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/r15_${tag}_$name -o p -- $B > /dev/null 2> gpurun_out/r15_${tag}_$name.err || { echo "pass $name failed"; tail -3 gpurun_out/r15_${tag}_$name.err; return 1; }
        python3 tools/pmc_summary.py $(find gpurun_out/r15_${tag}_$name -name '*counter_collection.csv') | grep -E "k_probe_round<[04]|k_lds_round<[04]" | sed 's/mg:://'; }
echo "== $tag ($2)"
run fetch FETCH_SIZE && run write WRITE_SIZE && run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
