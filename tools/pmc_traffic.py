"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into HBM bytes per launch, per kernel symbol
(the names bench.py's `roofline.kernel` / `stages` use).

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [workload] [clips_per_gpu]

Corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE tallies the
128-byte requests of wide coalesced reads at 64 bytes, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane
stores.  The two counters come from separate passes (FETCH_SIZE takes 3 of the 4 TCC counters, WRITE_SIZE 2).
"""
import csv
import glob
import json
import os
import sys

# substring of the rocprof kernel name -> key.  The log-mel front-end is one logical launch: its bytes are summed over
# all its kernels and divided by the launches of the tile kernel alone (COUNT_ONLY).
FAMILIES = {"gemm256p_kernel": "gemm256p_kernel", "gemm256_kernel": "gemm256_kernel", "gemm_bf16_kernel": "gemm_bf16_kernel",
            "attention_kernel": "attention_kernel", "layernorm_kernel": "layernorm_kernel", "logmel": "logmel_tile_kernel+logmel_finish_kernel"}
NOT_A_LAUNCH = ("logmel_finish_kernel",)


def family(name):
    for key, fam in FAMILIES.items():
        if key in name:
            return fam
    return None


def collect(d, counter):
    tot, cnt = {}, {}
    csv.field_size_limit(1 << 30)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                fam = family(row["Kernel_Name"])
                if fam is None:
                    continue
                tot[fam] = tot.get(fam, 0.0) + float(row["Counter_Value"])
                if not any(k in row["Kernel_Name"] for k in NOT_A_LAUNCH):
                    cnt[fam] = cnt.get(fam, 0) + 1
    return tot, cnt


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    workload = sys.argv[4] if len(sys.argv) > 4 else "pipeline-small"
    batch = int(sys.argv[5]) if len(sys.argv) > 5 else 64
    ft, fc = collect(fetch_dir, "FETCH_SIZE")
    wt, wc = collect(write_dir, "WRITE_SIZE")
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py; FETCH_SIZE x 2 (gfx950), KiB -> bytes",
           "workload": workload, "clips_per_gpu": batch, "kernels": {}}
    for fam in sorted(set(ft) | set(wt)):
        rd = ft.get(fam, 0.0) * 1024 * 2 / max(fc.get(fam, 1), 1)
        wr = wt.get(fam, 0.0) * 1024 / max(wc.get(fam, 1), 1)
        res["kernels"][fam] = {"launches_profiled": fc.get(fam, 0), "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                               "bytes_per_launch": rd + wr}
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
