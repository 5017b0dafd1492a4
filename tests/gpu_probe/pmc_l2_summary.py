"""Per-kernel, per-launch averages of several rocprofv3 --pmc passes: python pmc_l2_summary.py <dir with p1, p2, ...>"""
import csv, glob, re, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(set))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name).split("(")[0][:70] + " g" + r.get("Grid_Size", "?")
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[name][r["Counter_Name"]].add(r["Dispatch_Id"])
ctrs = sorted({c for n in agg for c in agg[n]})
names = sorted(agg, key=lambda n: -agg[n].get("FETCH_SIZE", 0))
print("kernel;launches;" + ";".join(c + "/launch" for c in ctrs) + ";l2_hit")
for n in names[:60]:
    per = {c: agg[n][c] / max(1, len(cnt[n][c])) for c in ctrs if c in agg[n]}
    h, m = per.get("TCC_HIT_sum", 0), per.get("TCC_MISS_sum", 0)
    print(n + ";" + str(max(len(v) for v in cnt[n].values())) + ";" + ";".join(f"{per.get(c, 0):.5g}" for c in ctrs) +
          ";" + (f"{h / (h + m):.3f}" if h + m else ""))
