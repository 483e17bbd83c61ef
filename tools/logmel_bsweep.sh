#!/bin/bash
# tools/logmel_bsweep.sh [batches]: tile-kernel duration per batch size from a rocprofv3 kernel trace
set -e -o pipefail
R=$GRAFT_REPO_ROOT
BL=${1:-1,8,32,64,128,256,512}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/bsweep
rm -rf $O && mkdir -p $O
PYTHONPATH=$R timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python $R/tools/logmel_bsweep.py $BL > $O/out.txt 2> $O/err.txt
python - "$O" "$BL" <<'PY'
import csv, glob, sys
o, bl = sys.argv[1], [int(x) for x in sys.argv[2].split(",")]
f = glob.glob(o + "/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "logmel_tile_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
for i, b in enumerate(bl):
    seg = d[4 * i:4 * i + 4]
    print(f"B={b:5d}  tile kernel us: " + " ".join(f"{x:8.1f}" for x in seg) + f"   min/clip {min(seg) / b:7.2f}")
PY
