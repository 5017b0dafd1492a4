"""Per-kernel sums of a rocprofv3 --pmc counter_collection CSV: python pmc_summary.py <dir>"""
import csv, glob, re, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"^void ", "", name).split("(")[0][:60]
    agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[name].add(r["Dispatch_Id"])
names = sorted(agg, key=lambda n: -agg[n].get("SQ_WAVE_CYCLES", 0))
ctrs = sorted({c for n in agg for c in agg[n]})
print("kernel;launches;" + ";".join(ctrs))
for n in names[:40]:
    print(n + ";" + str(len(cnt[n])) + ";" + ";".join(f"{agg[n].get(c, 0):.4g}" for c in ctrs))
