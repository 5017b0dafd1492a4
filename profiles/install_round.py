#!/usr/bin/env python3
"""Copy the summaries profiles/collect_round.sh left under gpurun_out/<rNN>/ into profiles/ under their <rNN>_* names:
    python profiles/install_round.py r05"""
import glob
import os
import shutil
import sys

RND = sys.argv[1] if len(sys.argv) > 1 else "r05"

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S = os.path.join(R, "gpurun_out", RND)
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


last_line("bench.json", f"{RND}_bench.json")
last_line("bench_traced.json", f"{RND}_bench_under_rocprof.json")
cp("kernel_stats.csv", f"{RND}_bench_kernel_stats.csv")
cp("pmc_traffic.csv", f"{RND}_pmc_traffic.csv")
cp("timed_region_kernel_stats.csv", f"{RND}_timed_region_kernel_stats.csv")
raw = glob.glob(os.path.join(S, "trace", "*", "*_kernel_stats.csv"))
if raw:
    shutil.copyfile(raw[0], os.path.join(P, f"{RND}_rocprofv3_kernel_stats_raw.csv"))
for cfg in ("suim", "cityscapes", "hela", "cityscapes_a2"):
    last_line(f"bench_{cfg}.json", f"{RND}_configs_bench_{cfg}.json")
    cp(f"cfg_{cfg}/kernel_stats.csv", f"{RND}_configs_kernel_stats_{cfg}.csv")
    cp(f"cfg_{cfg}/pmc_traffic.csv", f"{RND}_pmc_traffic_{cfg}.csv")
for f in sorted(glob.glob(os.path.join(S, "step_timeline_*.txt"))):
    shutil.copyfile(f, os.path.join(P, f"{RND}_" + os.path.basename(f)))
for f in sorted(glob.glob(os.path.join(S, "sq_counters_*.csv"))):
    shutil.copyfile(f, os.path.join(P, f"{RND}_" + os.path.basename(f)))
cp("configs_step_times_raw.txt", f"{RND}_configs_step_times.txt")
