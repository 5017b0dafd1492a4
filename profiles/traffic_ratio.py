#!/usr/bin/env python3
"""Counter traffic against algorithmic bytes, per variant of a configuration's dominant family:
<rNN>_pmc_traffic<tag>.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, 2 x fetch + write) x <rNN>_timed_region_kernel_stats<tag>.csv (the
library's algorithmic bytes per launch, rocprofv3's durations inside the timed region) -> <rNN>_traffic_vs_algorithmic<tag>.csv
    python profiles/traffic_ratio.py r06            every configuration that has both files
The family is the one the configuration's bench line names (`roofline.kernel`; conv_pipe stands for conv_pipe + conv_wide: one family in the
bench's hook); without a line: conv_pipe + conv_wide."""
import csv
import glob
import json
import os
import sys
RND = sys.argv[1] if len(sys.argv) > 1 else "r06"
HERE = os.path.dirname(os.path.abspath(__file__))


def one(tag):
    # From round 5 on the timed-region file has one row per (stage, variant) and the bench runs other batch sizes outside the clock (the
    # T(B) fit: 8 / 16 images): a variant is compared only where the two populations are the same -- it runs in ONE stage of the
    # generation, and of its PMC rows (one per grid size) the one with the most launches is that stage's.
    fam = ("conv_pipe", "conv_wide")
    try:
        k = json.load(open(os.path.join(HERE, f"{RND}_bench{tag}.json")))["roofline"]["kernel"].split("<")[0].split("+")[0]
        fam = ("conv_pipe", "conv_wide") if k.startswith("conv_pipe") else (k.replace("_kernel", ""),)
    except Exception:
        pass
    tr, n_stage = {}, {}
    for r in csv.DictReader(open(os.path.join(HERE, f"{RND}_timed_region_kernel_stats{tag}.csv"))):
        n_stage[r["kernel"]] = n_stage.get(r["kernel"], 0) + 1
        tr[r["kernel"]] = r
    pmc = {}
    for r in csv.DictReader(open(os.path.join(HERE, f"{RND}_pmc_traffic{tag}.csv"))):
        if r["kernel"] not in pmc or int(r["launches"]) > int(pmc[r["kernel"]]["launches"]):
            pmc[r["kernel"]] = r
    rows = []
    for k, r in pmc.items():
        t = tr.get(k)
        if not k.startswith(fam) or not t or not t["algorithmic_MB_per_launch"] or n_stage.get(k, 0) != 1:
            continue
        alg, hbm = float(t["algorithmic_MB_per_launch"]), float(r["hbm_MB_per_launch_corrected(2*fetch+write)"])
        rows.append([k, int(r["launches"]), alg, hbm, round(hbm / alg, 3), float(t["avg_us"]), float(t["frac_of_8TBps"]), t.get("stage", "")])
    if not rows:
        return None
    rows.sort(key=lambda x: -x[1] * x[3])
    ta, th = sum(r[1] * r[2] for r in rows), sum(r[1] * r[3] for r in rows)
    with open(os.path.join(HERE, f"{RND}_traffic_vs_algorithmic{tag}.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches_in_pmc_pass", "algorithmic_MB_per_launch", "pmc_MB_per_launch(2*fetch+write)", "ratio", "rocprof_avg_us_timed_region", "frac_of_8TBps", "stage"])
        w.writerows(rows)
        w.writerow(["FAMILY (launch-weighted) " + " + ".join(fam), sum(r[1] for r in rows), round(ta / sum(r[1] for r in rows), 1), round(th / sum(r[1] for r in rows), 1), round(th / ta, 3), "", "", ""])
    return round(th / ta, 3)


for p in sorted(glob.glob(os.path.join(HERE, f"{RND}_pmc_traffic*.csv"))):
    tag = os.path.basename(p)[len(f"{RND}_pmc_traffic"):-4]
    if os.path.exists(os.path.join(HERE, f"{RND}_timed_region_kernel_stats{tag}.csv")):
        print(f"family ratio{tag or ' (default)'}:", one(tag))
