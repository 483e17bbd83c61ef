for b in 16 32 64 96 128; do
  timeout -k 10 200 python bench.py --workload pipeline-small --batch $b --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bs_$b.json || exit 1
  python - $b <<'PY'
import json, sys
d = json.load(open("gpurun_out/bs_%s.json" % sys.argv[1]))
print(sys.argv[1], round(d["value"]), round(d["ms_per_step"], 3), {k: (round(v["ms_per_step"], 3), round(v["frac"], 3)) for k, v in d["stages"].items()})
PY
done
