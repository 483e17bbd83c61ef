#!/bin/bash
# One GPU call that measures every bench workload: the bench line (not profiled), a rocprofv3 kernel trace with
# statistics, and -- for the headline workload -- the two HBM counter passes (separate runs, counters only).
#   bash tools/profile_all.sh [workloads...]      results under gpurun_out/profall/<workload>/
set -e -o pipefail
R=$GRAFT_REPO_ROOT
WLS=${@:-pipeline-small pipeline-large train-large logmel logmel-128 logmel-ragged splice}
cd /tmp && export TMPDIR=/tmp
for wl in $WLS; do
  O=$R/gpurun_out/profall/$wl
  rm -rf $O && mkdir -p $O
  echo "== $wl"
  timeout -k 10 400 python $R/bench.py --workload $wl --warmup 3 > $O/bench.json 2> $O/bench.err
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt.err
  if [ $wl = pipeline-small ]; then
    timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python $R/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_f.json 2> $O/pmc_f.err
    timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python $R/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_w.json 2> $O/pmc_w.err
  fi
  python - "$O/bench.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["config"]["workload"], "value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3), "roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"], 3))
PY
done
echo PROFILE_ALL_OK
