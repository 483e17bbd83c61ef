set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_g $R/gpurun_out/pmc_f $R/gpurun_out/pmc_w
cd $R
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > gpurun_out/prof_g/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_g/kt -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_g/bench_under_rocprof.json
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_f/bench.json
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_w/bench.json
find gpurun_out/prof_g/kt -name "*kernel_stats.csv" | head -3
