#!/bin/bash
# A/B of library builds on the log-mel workload inside one GPU call: tools/ab_logmel.sh name1 name2 ... (asr_amd/lib<name>.so)
set -e -o pipefail
for wl in ${WLS:-logmel logmel-128}; do
for l in "$@"; do
  export ASRK_LIB_PATH=$PWD/asr_amd/lib$l.so
  timeout -k 10 100 python bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/abl_$l.json
  python - "$l" "$wl" <<'PY'
import json, sys
d = json.load(open("gpurun_out/abl_%s.json" % sys.argv[1]))
print(sys.argv[2], sys.argv[1], "ms/step", round(d["ms_per_step"], 4), "GB/s", round(d["roofline"]["achieved"]))
PY
done
done
