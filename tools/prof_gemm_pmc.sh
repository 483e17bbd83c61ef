#!/bin/bash
# SQ counter passes for the persistent GEMM inside pipeline-small (B = 16 keeps the passes short): tools/prof_gemm_pmc.sh <tag>
set -e -o pipefail
TAG=${1:-x}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/gemm_pmc_$TAG
mkdir -p $O
B="python $R/bench.py --workload pipeline-small --steps 1 --warmup 1 --no-cpu-baseline"
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_ADDR_CONFLICT --output-format csv -d $O/pmc_a -- $B > $O/pmc_a.json 2> $O/pmc_a.err
timeout -s KILL 150 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/pmc_b -- $B > $O/pmc_b.json 2> $O/pmc_b.err
timeout -s KILL 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_ACTIVE_INST_MISC SQ_INSTS_SALU --output-format csv -d $O/pmc_c -- $B > $O/pmc_c.json 2> $O/pmc_c.err
python $R/tools/pmc_summary.py $O gemm256p > $O/summary.txt
cat $O/summary.txt
