#!/bin/bash
# Runs on the GPU box: FETCH/WRITE + TCC + TCP counter passes of the default bench for one build of the library.
# usage: tools/pmc_quick.sh <tag> [lib.so]   -> one summary block on stdout (match rounds and collapse kernels)
tag=${1:-x}; lib=${2:-}; export TMPDIR=/tmp
[ -n "$lib" ] && export MIRGE_HIP_LIB=$lib
B="python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --e2e-reads 0 --realistic-reads 0 --perl-reads 0 --repeats-reads 0"
run() { name=$1; shift; rm -rf gpurun_out/pq_${tag}_$name; timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pq_${tag}_$name -o p -- $B > /dev/null 2> gpurun_out/pq_${tag}_$name.err || { echo "pass $name failed"; tail -3 gpurun_out/pq_${tag}_$name.err; return 1; }
        python3 tools/pmc_summary.py $(find gpurun_out/pq_${tag}_$name -name '*counter_collection.csv') | grep -E "k_match_round<2|k_probe|k_lds|k_collapse_(local|emit)<2" | sed 's/mg:://'; }
echo "== $tag"
run fetch FETCH_SIZE && run write WRITE_SIZE &&
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum &&
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
