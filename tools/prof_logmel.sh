#!/bin/bash
# rocprofv3 kernel trace + SQ counter passes of the log-mel workload (config 2): tools/prof_logmel.sh <tag> [workload]
set -e -o pipefail
TAG=${1:-x}; WL=${2:-logmel}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/logmel_$TAG
mkdir -p $O
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_kt.json 2> $O/kt.err
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $O/pmc_a -- python $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_a.json 2> $O/pmc_a.err
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/pmc_b -- python $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_b.json 2> $O/pmc_b.err
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs head -8
