"""Condense a rocprofv3 --kernel-trace CSV of step_trace.py into a per-step timeline: python trace_summary.py <dir> > out.txt"""
import csv, glob, re, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"^void ", "", name).split("(")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?"),
                 r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
# start of a forward: the uint8 stem (training) or the first encoder conv with the input block on load (inference, LM_STEM = 6)
stems = [i for i, r in enumerate(rows) if re.match(r"conv_(pipe|wide)_kernel<\(ImkLoadMode\)[46]|conv_(pipe|wide)_kernel<[46]", r[2])]   # wide: alpha > 1
folds = [i for i, r in enumerate(rows) if r[2].startswith("pack_conv_batched_kernel")]   # end of an optimizer step
heads = [i for i, r in enumerate(rows) if "head_kernel" in r[2] or "head_softmax_kernel" in r[2]]                          # end of an inference call
def show(seg, title):
    t0 = seg[0][0]
    span = (seg[-1][1] - t0) / 1e3
    busy = sum(e - s for s, e, *_ in seg) / 1e3
    print(f"== {title}: span {span:.1f} us, kernels {len(seg)}, busy {busy:.1f} us")
    agg = defaultdict(lambda: [0, 0.0])
    for s, e, n, *_ in seg:
        agg[n][0] += 1; agg[n][1] += (e - s) / 1e3
    busy_main = sum(e - s for s, e, n, gx, wx, q in seg if q == seg[0][5]) / 1e3
    print(f"   first kernel's stream: busy {busy_main:.1f} us of the span")
    for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"   {n:42s} n={c:3d} us={us:8.1f}")
    main = seg[0][5]                     # the stream of the step's first kernel; "gap" = idle time of that stream before a kernel
    last_end = {}
    for s, e, n, gx, wx, q in seg:
        try: g = int(gx) // max(int(wx), 1)
        except ValueError: g = gx
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        print(f"  {(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} {'M' if q == main else 's'} gap {gap:6.1f} {n:42s} {g}")
train = [(st, min(f for f in folds if f > st)) for st in stems if any(f > st for f in folds)
         and not any(h > st and h < min(f for f in folds if f > st) for h in heads)]
if len(train) >= 2:
    st, en = train[-2]
    show(rows[st:en + 1], "training step (second to last)")
infer = [(st, min(h for h in heads if h > st)) for st in stems if any(h > st for h in heads)
         and not any(f > st and f < min(h for h in heads if h > st) for f in folds)]
if len(infer) >= 2:
    st, en = infer[-2]
    show(rows[st:en + 1], "inference call (second to last)")
