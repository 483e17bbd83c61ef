#!/bin/bash
# A/B of one library under different environment settings inside one GPU call:
#   LIB=asrk_diag tools/ab_env.sh "ASRK_GEMM_PSKEW=0" "ASRK_GEMM_PSKEW=3" ...
set -e -o pipefail
export ASRK_LIB_PATH=$PWD/asr_amd/lib${LIB:-asrk_diag}.so
for rep in 1 2; do
for e in "$@"; do
  env $e timeout -k 10 200 python bench.py --workload ${WL:-pipeline-small} --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/abenv.json
  python - "$e" <<'PY'
import json, sys
d = json.load(open("gpurun_out/abenv.json"))
print(sys.argv[1], round(d["value"]), round(d["ms_per_step"], 3), {k: round(v["ms_per_step"], 3) for k, v in d["stages"].items() if "gemm" in k or "attention" in k})
PY
done
done
