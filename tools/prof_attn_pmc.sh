#!/bin/bash
# SQ counter passes for the attention kernel inside pipeline-small (B = 16 keeps the passes short): tools/prof_attn_pmc.sh <tag> [lib name]
set -e -o pipefail
TAG=${1:-x}
R=$GRAFT_REPO_ROOT
if [ -n "$2" ]; then export ASRK_LIB_PATH=$R/asr_amd/lib$2.so; fi
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/attn_pmc_$TAG
mkdir -p $O
B="python $R/bench.py --workload pipeline-small --batch 16 --steps 1 --warmup 1 --no-cpu-baseline"
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAVES --output-format csv -d $O/pmc_a -- $B > $O/pmc_a.json 2> $O/pmc_a.err
timeout -s KILL 150 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/pmc_b -- $B > $O/pmc_b.json 2> $O/pmc_b.err
timeout -s KILL 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $O/pmc_c -- $B > $O/pmc_c.json 2> $O/pmc_c.err
python $R/tools/pmc_summary.py $O attention > $O/summary.txt
cat $O/summary.txt
