#!/bin/bash
# vector-cache / L2 request counters of one bench workload: tools/prof_cache.sh <tag> [workload]
set -e -o pipefail
TAG=${1:-x}; WL=${2:-logmel}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/cache_$TAG
mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_c -- python $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_c.json 2> $O/pmc_c.err
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_f.json 2> $O/pmc_f.err
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCP_TCC_WRITE_REQ_sum --output-format csv -d $O/pmc_w -- python $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_w.json 2> $O/pmc_w.err
