#!/bin/bash
# Runs on the GPU box: SQ counter pass of the default bench.  usage: tools/pmc_sq.sh <tag> [lib.so]
tag=${1:-x}; lib=${2:-}; export TMPDIR=/tmp
[ -n "$lib" ] && export MIRGE_HIP_LIB=$lib
B="python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --e2e-reads 0 --realistic-reads 0 --perl-reads 0 --repeats-reads 0"
rm -rf gpurun_out/sq_$tag
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/sq_$tag -o p -- $B > /dev/null 2> gpurun_out/sq_$tag.err || { echo "sq pass failed"; tail -3 gpurun_out/sq_$tag.err; exit 1; }
python3 tools/pmc_summary.py $(find gpurun_out/sq_$tag -name '*counter_collection.csv') | grep -E "k_match_round<2|k_probe|k_lds|k_collapse_(local|emit)<2|k_counts<2" | sed 's/mg:://'
