"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files: python tools/pmc_summary.py <dir> [name filter]"""
import csv
import glob
import os
import sys

d, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
tot = {}
csv.field_size_limit(1 << 30)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        if flt in row["Kernel_Name"]:
            tot.setdefault((row["Kernel_Name"].split("(")[0][-40:], row["Counter_Name"]), []).append(float(row["Counter_Value"]))
for k, v in sorted(tot.items()):
    print(f"{k[0]:42s} {k[1]:24s} n={len(v):3d} avg={sum(v) / len(v):16.1f}")
