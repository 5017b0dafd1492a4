"""Top kernels of a rocprofv3 --kernel-trace --stats run: python kstats.py <output dir> [n]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 22]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).replace("void ", "").split("(")[0][:60]
    print(f"{n:62s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:9.1f} us  {100 * float(r['TotalDurationNs']) / tot:5.1f} %")
