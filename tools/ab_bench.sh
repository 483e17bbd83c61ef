#!/bin/bash
# A/B of two builds of the library inside one GPU call: tools/ab_bench.sh prev new prev new  (names: asr_amd/lib<name>.so)
# Stops at the first failing step; nothing else touches the GPU after a failure.
set -e -o pipefail
for l in "$@"; do
  export ASRK_LIB_PATH=$PWD/asr_amd/lib$l.so
  timeout -k 10 200 python bench.py --workload ${WL:-pipeline-small} --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab_$l.json
  python - "$l" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_%s.json" % sys.argv[1]))
print(sys.argv[1], round(d["value"]), round(d["ms_per_step"], 3), {k: round(v["ms_per_step"], 3) for k, v in d["stages"].items()})
PY
done
