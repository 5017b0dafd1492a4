import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
for k in a.files:
    if k not in b.files: continue
    x, y = a[k], b[k]
    d = x != y
    print(k, x.shape, "differing", int(d.sum()))
    for i in np.argwhere(d)[:8]:
        print("    ", tuple(int(v) for v in i), x[tuple(i)], y[tuple(i)])
