"""Copy the summaries of tools/profile_all.sh from gpurun_out/profall/ into profiles/ (tracked):
    python tools/collect_profiles.py <tag>          e.g. r2_d  ->  profiles/r2_d_<workload>_{bench,bench_under_rocprof}.json,
                                                                   profiles/r2_d_<workload>_kernel_stats.csv
and, when the HBM counter passes of pipeline-small are there, regenerate profiles/hbm_traffic.json."""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "profall")
for wl in sorted(os.listdir(src)):
    d = os.path.join(src, wl)
    for name in ("bench.json", "bench_under_rocprof.json"):
        if os.path.exists(os.path.join(d, name)) and os.path.getsize(os.path.join(d, name)) > 0:
            shutil.copy(os.path.join(d, name), os.path.join(ROOT, "profiles", f"{tag}_{wl}_{name}"))
    ks = glob.glob(os.path.join(d, "kt", "**", "*kernel_stats.csv"), recursive=True)
    if ks:
        shutil.copy(ks[0], os.path.join(ROOT, "profiles", f"{tag}_{wl}_kernel_stats.csv"))
    if os.path.isdir(os.path.join(d, "pmc_f")) and os.path.isdir(os.path.join(d, "pmc_w")):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), os.path.join(d, "pmc_f"),
                               os.path.join(d, "pmc_w"), os.path.join(ROOT, "profiles", "hbm_traffic.json"), wl, "64"],
                              stdout=subprocess.DEVNULL)
    print("collected", wl)
