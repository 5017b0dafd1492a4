#!/usr/bin/env python3
"""Copy the summaries profiles/collect_round.sh left under gpurun_out/<rNN>/cfg<tag>/ into profiles/ as <rNN>_<stem><tag>.*:
    python profiles/install_round.py r06
then profiles/traffic_ratio.py <rNN> for the default configuration's traffic / algorithmic table."""
import glob
import json
import os
import shutil
import sys

RND = sys.argv[1] if len(sys.argv) > 1 else "r06"

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S = os.path.join(R, "gpurun_out", RND)
P = os.path.join(R, "profiles")


INSTALLED = []


def cp(src, dst):
    src = os.path.join(S, src)
    if os.path.exists(src):
        shutil.copyfile(src, os.path.join(P, dst))
        INSTALLED.append(dst)
        print("installed", dst)


def last_line(src, dst):
    src = os.path.join(S, src)
    if os.path.exists(src):
        lines = [l for l in open(src).read().splitlines() if l.startswith("{")]
        if lines:
            open(os.path.join(P, dst), "w").write(lines[-1] + "\n")
            print("installed", dst)


for d in sorted(glob.glob(os.path.join(S, "cfg_*"))):
    name = os.path.basename(d)[4:]
    tag = "" if name == "isic" else "_" + name
    sub = os.path.basename(d)
    INSTALLED.clear()
    last_line(f"{sub}/bench.json", f"{RND}_bench{tag}.json")
    cp(f"{sub}/bench_detail.json", f"{RND}_bench_detail{tag}.json")
    last_line(f"{sub}/bench_traced.json", f"{RND}_bench_under_rocprof{tag}.json")
    cp(f"{sub}/kernel_stats.csv", f"{RND}_kernel_stats{tag}.csv")
    cp(f"{sub}/timed_region_kernel_stats.csv", f"{RND}_timed_region_kernel_stats{tag}.csv")
    cp(f"{sub}/excess_by_kernel.csv", f"{RND}_excess_by_kernel{tag}.csv")
    cp(f"{sub}/timed_region_family_union.csv", f"{RND}_timed_region_family_union{tag}.csv")
    cp(f"{sub}/pmc_traffic.csv", f"{RND}_pmc_traffic{tag}.csv")
    # the provenance of a configuration's files counts only when its trace was actually cut
    if os.path.exists(os.path.join(d, "timed_region_kernel_stats.csv")):
        rec = json.load(open(os.path.join(d, "provenance.json")))
        rec["files"] = sorted(INSTALLED)
        json.dump(rec, open(os.path.join(P, f"{RND}_provenance{tag}.json"), "w"), indent=1)
        print("installed", f"{RND}_provenance{tag}.json")
    raw = glob.glob(os.path.join(d, "trace", "*", "*_kernel_stats.csv"))
    if raw and not tag:
        shutil.copyfile(raw[0], os.path.join(P, f"{RND}_rocprofv3_kernel_stats_raw.csv"))
for f in sorted(glob.glob(os.path.join(S, "step_timeline_*.txt"))):
    shutil.copyfile(f, os.path.join(P, f"{RND}_" + os.path.basename(f)))
for f in sorted(glob.glob(os.path.join(S, "sq_counters_*.csv"))):
    shutil.copyfile(f, os.path.join(P, f"{RND}_" + os.path.basename(f)))
cp("configs_step_times_raw.txt", f"{RND}_configs_step_times.txt")
