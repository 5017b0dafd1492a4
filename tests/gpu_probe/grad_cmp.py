"""Compare two grad_dump.py files: per-layer relative L2 difference of the gradients, the parameters and the probabilities."""
import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
names, offs = a["names"], a["offs"]
for key in [k for k in a.files if k.startswith("g")]:
    ga, gb = a[key], b[key]
    print(key, "stats", a["st" + key[1:]], b["st" + key[1:]], "nonfinite", (~np.isfinite(ga)).sum(), (~np.isfinite(gb)).sum())
    for n, (ow, ob, kind, ks, ci, co) in zip(names, offs):
        nw = ks * ks * ci * co if kind == 0 else co
        for what, o, cnt in (("w", ow, nw), ("b", ob, co)):
            x, y = ga[o:o + cnt].astype(np.float64), gb[o:o + cnt].astype(np.float64)
            d = np.sqrt(((x - y) ** 2).sum() / max((y ** 2).sum(), 1e-30))
            if d > 1e-3 or not np.isfinite(d):
                print("   %-8s %s rel-L2 %.3e  max|a| %.3e max|b| %.3e" % (n, what, d, np.abs(x).max(), np.abs(y).max()))
print("params equal", np.array_equal(a["p"], b["p"]), "max |dp|", np.abs(a["p"] - b["p"]).max())
print("probs equal", np.array_equal(a["probs"], b["probs"]), "max |d|", np.abs(a["probs"] - b["probs"]).max(), "nonfinite", (~np.isfinite(a["probs"])).sum())
