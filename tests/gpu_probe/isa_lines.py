"""VALU instructions of a kernel's hottest loop attributed to source lines (hipcc -gline-tables-only -S listing):
   python isa_lines.py <file.s> <substring of the mangled kernel name> [csrc dir]"""
import os, re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
csrc = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "inconsistencymasks_amd", "csrc")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and ": " in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
cur, labels, ins = None, {}, []
for l in lines[start:end]:
    s = l.strip()
    m = re.match(r"^(\.LBB[0-9_]+):", s)
    if m: labels[m.group(1)] = len(ins); continue
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if m: cur = (int(m.group(1)), int(m.group(2))); continue
    if not s or s.startswith((";", ".")) or s.endswith(":"): continue
    ins.append((s.split(";")[0].strip(), cur))
best = None
for idx, (i, _) in enumerate(ins):
    m = re.match(r"s_cbranch\S*\s+(\.LBB[0-9_]+)|s_branch\s+(\.LBB[0-9_]+)", i)
    if m:
        t = labels.get(m.group(1) or m.group(2))
        if t is not None and t < idx and (best is None or idx - t > best[1] - best[0]): best = (t, idx)
loop = ins[best[0]:best[1] + 1]
valu, salu = Counter(), Counter()
for i, loc in loop:
    op = i.split()[0]
    k = (files.get(loc[0], "?"), loc[1]) if loc else ("?", 0)
    if op.startswith("v_") and not op.startswith("v_mfma"): valu[k] += 1
    elif op.startswith("s_"): salu[k] += 1
print(f"loop: {len(loop)} instructions, VALU {sum(valu.values())}, SALU {sum(salu.values())}")
src = {}
for (f, ln), c in valu.most_common(45):
    text = ""
    try:
        if f not in src: src[f] = open(os.path.join(csrc, f)).read().split("\n")
        text = src[f][ln - 1].strip()[:120]
    except Exception:
        pass
    print(f"{c:4d} {f}:{ln}  {text}")
