"""Condense a rocprofv3 --kernel-trace CSV of step_trace.py into a per-step timeline: python trace_summary.py <dir> > out.txt"""
import csv, glob, re, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"^void ", "", name).split("(")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?")))
rows.sort()
begins = [i for i, r in enumerate(rows) if r[2].startswith("ctl_begin_step")]
def show(seg, title):
    t0 = seg[0][0]
    span = (seg[-1][1] - t0) / 1e3
    busy = sum(e - s for s, e, *_ in seg) / 1e3
    print(f"== {title}: span {span:.1f} us, kernels {len(seg)}, busy {busy:.1f} us")
    agg = defaultdict(lambda: [0, 0.0])
    for s, e, n, *_ in seg:
        agg[n][0] += 1; agg[n][1] += (e - s) / 1e3
    for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"   {n:42s} n={c:3d} us={us:8.1f}")
    for s, e, n, gx, wx in seg:
        try: g = int(gx) // max(int(wx), 1)
        except ValueError: g = gx
        print(f"  {(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} {n:42s} {g}")
if len(begins) >= 2:
    show(rows[begins[-2]:begins[-1]], "training step (second to last)")
last_train_end = begins[-1]
# inference calls follow the last training step: split at the u8 stem kernel
tail = rows[last_train_end:]
stems = [i for i, r in enumerate(tail) if "conv_pipe_kernel<4" in r[2]]
if len(stems) >= 3:
    show(tail[stems[-2]:stems[-1]], "inference call B=128 (second to last)")
