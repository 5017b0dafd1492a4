"""Per-kernel totals of the LAST `n` launches-per-step of a rocprofv3 --kernel-trace run: python kernel_table.py <dir> <steps>"""
import csv, glob, re, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
steps = int(sys.argv[2])
rows = []
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"^void ", "", name).split("(")[0]
    if name.startswith("_ZN"):
        name = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", name)[:40]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
agg = defaultdict(lambda: [0, 0.0])
for s, e, n in rows:
    if n.startswith("at::") or "elementwise" in n or "distribution" in n: continue
    agg[n][0] += 1; agg[n][1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"all {steps} steps: {tot / steps:.1f} us of kernel time per step")
for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"   {n:52s} n/step={c / steps:5.1f} us/step={us / steps:8.1f}  avg={us / c:7.1f}")
