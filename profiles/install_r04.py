#!/usr/bin/env python3
"""Copy the summaries profiles/collect_r04.sh left under gpurun_out/r04/ into profiles/ under their r04_* names."""
import glob
import os
import shutil

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S = os.path.join(R, "gpurun_out", "r04")
P = os.path.join(R, "profiles")


def cp(src, dst):
    src = os.path.join(S, src)
    if os.path.exists(src):
        shutil.copyfile(src, os.path.join(P, dst))
        print("installed", dst)


def last_line(src, dst):
    src = os.path.join(S, src)
    if os.path.exists(src):
        lines = [l for l in open(src).read().splitlines() if l.startswith("{")]
        if lines:
            open(os.path.join(P, dst), "w").write(lines[-1] + "\n")
            print("installed", dst)


last_line("bench.json", "r04_bench.json")
last_line("bench_traced.json", "r04_bench_under_rocprof.json")
cp("kernel_stats.csv", "r04_bench_kernel_stats.csv")
cp("pmc_traffic.csv", "r04_pmc_traffic.csv")
cp("timed_region_kernel_stats.csv", "r04_timed_region_kernel_stats.csv")
raw = glob.glob(os.path.join(S, "trace", "*", "*_kernel_stats.csv"))
if raw:
    shutil.copyfile(raw[0], os.path.join(P, "r04_rocprofv3_kernel_stats_raw.csv"))
for cfg in ("suim", "cityscapes", "hela", "cityscapes_a2"):
    last_line(f"bench_{cfg}.json", f"r04_configs_bench_{cfg}.json")
    cp(f"cfg_{cfg}/kernel_stats.csv", f"r04_configs_kernel_stats_{cfg}.csv")
    cp(f"cfg_{cfg}/pmc_traffic.csv", f"r04_pmc_traffic_{cfg}.csv")
for f in sorted(glob.glob(os.path.join(S, "step_timeline_*.txt"))):
    shutil.copyfile(f, os.path.join(P, "r04_" + os.path.basename(f)))
for f in sorted(glob.glob(os.path.join(S, "sq_counters_*.csv"))):
    shutil.copyfile(f, os.path.join(P, "r04_" + os.path.basename(f)))
cp("configs_step_times_raw.txt", "r04_configs_step_times.txt")
