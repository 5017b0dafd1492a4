"""Instruction histogram of one kernel of a hipcc -S listing, whole body and its hottest loop (the largest backward branch):
   python isa_hist.py <file.s> <substring of the mangled kernel name>"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]) and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end + 1]
labels = {}
ins = []
for l in body:
    s = l.strip()
    m = re.match(r"^(\.LBB[0-9_]+):", s)
    if m: labels[m.group(1)] = len(ins); continue
    if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"): continue
    ins.append(s.split(";")[0].strip())
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop"): return op.split()[0]
    return "salu"
def hist(seq, title):
    c = Counter(cls(i.split()[0]) for i in seq)
    ops = Counter(i.split()[0] for i in seq)
    print(f"== {title}: {len(seq)} instructions", dict(c))
    print("   top:", ", ".join(f"{k} {v}" for k, v in ops.most_common(28)))
hist(ins, "whole kernel")
best = None
for idx, i in enumerate(ins):
    m = re.match(r"s_cbranch\S*\s+(\.LBB[0-9_]+)|s_branch\s+(\.LBB[0-9_]+)", i)
    if m:
        t = labels.get(m.group(1) or m.group(2))
        if t is not None and t < idx and (best is None or idx - t > best[1] - best[0]): best = (t, idx)
if best: hist(ins[best[0]:best[1] + 1], "largest loop")
