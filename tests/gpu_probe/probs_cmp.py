import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
pa, pb = a["probs"], b["probs"]
d = pa != pb
print("shape", pa.shape, "differing", int(d.sum()), "params equal", np.array_equal(a["p"], b["p"]))
idx = np.argwhere(d)
for i in idx[:12]:
    print(tuple(i), pa[tuple(i)], pb[tuple(i)])
if len(idx):
    print("rows", sorted(set(idx[:, 1].tolist()))[:20], "cols", sorted(set(idx[:, 2].tolist()))[:20], "imgs", sorted(set(idx[:, 0].tolist())))
