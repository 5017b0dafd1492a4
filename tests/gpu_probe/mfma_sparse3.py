"""GPU: the stem's case -- v_mfma_f32_16x16x32_f16 with only THREE non-zero products (an RGB pixel x a 3 -> 8 channel Conv1x1; C = 0).
Which model reproduces the matrix core, and how often does the stem-on-load's VALU form (fl(fl(fl(w0 x0) + w1 x1) + w2 x2) with fused
multiply-adds) differ from it?  Operands like the real ones: x = fp16(byte / 255), w = fp16(he_normal draw).
    python tests/gpu_probe/mfma_sparse3.py        (needs build/mfma_numerics_probe)"""
import os, subprocess, sys
from fractions import Fraction
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exe = os.path.join(ROOT, "build", "mfma_numerics_probe")
N = 256
rng = np.random.default_rng(7)
A = np.zeros((N, 16, 32), np.float16); B = np.zeros((N, 32, 16), np.float16); C = np.zeros((N, 16, 16), np.float32)
A[:, :, :3] = (rng.standard_normal((N, 16, 3)) * np.sqrt(2.0 / 3) / 0.8796).astype(np.float16)          # weights [co][k]
B[:, :3, :] = (rng.integers(0, 256, (N, 3, 16)).astype(np.float32) / 255.0).astype(np.float16)           # pixels [k][pixel]
tmp = "/tmp/mfma_s3"
os.makedirs(tmp, exist_ok=True)
with open(f"{tmp}/in.bin", "wb") as f:
    f.write(A.tobytes()); f.write(B.tobytes()); f.write(C.tobytes())
subprocess.check_call([exe, f"{tmp}/in.bin", f"{tmp}/out.bin", str(N)])
D = np.fromfile(f"{tmp}/out.bin", np.float32).reshape(N, 16, 16)
f32 = np.float32
def fl(x):
    if x == 0: return f32(0)
    z = f32(np.float64(x))
    cand = [z, np.nextafter(z, f32(np.inf)), np.nextafter(z, f32(-np.inf))]
    return f32(min(cand, key=lambda c: (abs(Fraction(float(c)) - x), int(np.frombuffer(f32(c).tobytes(), np.uint32)[0]) & 1)))
def fz(x):
    z = fl(x)
    if abs(Fraction(float(z))) > abs(x): z = np.nextafter(z, f32(0))
    return f32(z)
names = ("exact_rne", "exact_rtz", "seq012_rne (the VALU form)", "seq210_rne", "seq012_rtz")
hits = dict.fromkeys(names, 0)
total = differ_f16 = 0
for c in range(N):
    for i in range(16):
        for j in range(16):
            p = [Fraction(float(A[c, i, k])) * Fraction(float(B[c, k, j])) for k in range(3)]
            got = f32(D[c, i, j])
            seq = lambda order, r: r(Fraction(float(r(Fraction(float(r(p[order[0]]))) + p[order[1]]))) + p[order[2]])
            res = {"exact_rne": fl(sum(p)), "exact_rtz": fz(sum(p)), "seq012_rne (the VALU form)": seq((0, 1, 2), fl), "seq210_rne": seq((2, 1, 0), fl),
                   "seq012_rtz": seq((0, 1, 2), fz)}
            total += 1
            for kname, v in res.items():
                hits[kname] += int(v.tobytes() == got.tobytes())
            differ_f16 += int(np.float16(res["seq012_rne (the VALU form)"]) != np.float16(got))
print(f"{total} three-product sums; the matrix core's result is reproduced bit for bit by:")
for k, v in hits.items():
    print(f"   {k:32s} {v:6d}  ({100.0 * v / total:.2f} %)")
print(f"VALU form != matrix core after rounding to fp16 (before bias / ReLU): {differ_f16} of {total} ({100.0 * differ_f16 / total:.4f} %)")
