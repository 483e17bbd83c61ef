#!/bin/bash
# Round 2, first GPU call: parity at the benchmark's shapes, the whole GPU suite, the bench line, and SQ counters of
# the log-mel kernel (counter evidence before the rewrite).  Stops at the first failing step.
set -e -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 600 python -u -m pytest tests/test_bench_shapes_gpu.py -x -v --timeout 120 --timeout-method thread -s > gpurun_out/t_shapes.log 2>&1 || { tail -40 gpurun_out/t_shapes.log; exit 1; }
tail -25 gpurun_out/t_shapes.log
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t_all.log 2>&1 || { tail -40 gpurun_out/t_all.log; exit 1; }
tail -5 gpurun_out/t_all.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err
timeout -k 10 200 python bench.py --workload logmel --steps 20 --warmup 3 > gpurun_out/bench_logmel_a.json 2> gpurun_out/bench_logmel_a.err
timeout -k 10 200 python bench.py --workload logmel-128 --steps 20 --warmup 3 > gpurun_out/bench_logmel128_a.json 2> gpurun_out/bench_logmel128_a.err
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_logmel_a -- python $R/bench.py --workload logmel --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_logmel_a.json 2> $R/gpurun_out/pmc_logmel_a.err
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_logmel_b -- python $R/bench.py --workload logmel --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_logmel_b.json 2> $R/gpurun_out/pmc_logmel_b.err
echo ALL_OK
