#!/bin/bash
# Runs on the GPU box: counter passes over the default bench (each in a run of its own, as the box requires).
# usage: tools/pmc_passes.sh <tag>  -> gpurun_out/pmc_<tag>_<pass>/ + a summary on stdout
tag=${1:-x}; export TMPDIR=/tmp
B="python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --e2e-reads 0 --realistic-reads 0 --perl-reads 0 --repeats-reads 0"
run() { name=$1; shift; timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_${tag}_$name -o p -- $B > /dev/null 2> gpurun_out/pmc_${tag}_$name.err || { echo "pass $name failed"; tail -3 gpurun_out/pmc_${tag}_$name.err; return 1; }
        python3 tools/pmc_summary.py $(find gpurun_out/pmc_${tag}_$name -name '*counter_collection.csv') | grep -E "==|k_match_round<2|k_collapse_(local|emit)<2"; }
run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD &&
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum &&
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_WRITE_REQ_sum
