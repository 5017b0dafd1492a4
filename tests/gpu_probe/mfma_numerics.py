"""GPU: which accumulation model does v_mfma_f32_16x16x32_f16 follow?  Generates random fp16 operands with a wide exponent spread (and
zero-padded variants), runs build/mfma_numerics_probe, and counts the outputs each candidate model reproduces bit for bit:
  seq      : acc = C; for k = 0..31: acc = fl32(acc + a_k b_k)                       (products are exact in fp32)
  exact    : fl32(C + sum of all 32 products), one rounding
  grp8     : per 8-slot lane group an exact sum, then acc = fl32(acc + group) for g = 0..3
  grp8_c_last / grp4 / pair : variants (C added last; groups of 4; pairs);  *_rtz: the same with every rounding toward zero
Usage (from the repository root, on the GPU box): python tests/gpu_probe/mfma_numerics.py"""
import os, subprocess, sys
from fractions import Fraction
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exe = os.path.join(ROOT, "build", "mfma_numerics_probe")
N = 64
rng = np.random.default_rng(1)
def rnd16(shape, spread):
    return (rng.standard_normal(shape) * np.exp2(rng.integers(-spread, spread + 1, shape))).astype(np.float16)
SP = int(os.environ.get("SPREAD", 6))
A = rnd16((N, 16, 32), SP); B = rnd16((N, 32, 16), SP); C = (rng.standard_normal((N, 16, 16)) * 4).astype(np.float32)
C[N // 2:] = 0                                              # second half: C = 0 (the conv kernels' first k-step)
for c in range(N // 4):                                     # a quarter: only k-slots 0..2 non-zero (the stem's 3 products)
    A[4 * c, :, 3:] = 0
tmp = "/tmp/mfma_num"
os.makedirs(tmp, exist_ok=True)
with open(f"{tmp}/in.bin", "wb") as f:
    f.write(A.tobytes()); f.write(B.tobytes()); f.write(C.tobytes())
subprocess.check_call([exe, f"{tmp}/in.bin", f"{tmp}/out.bin", str(N)])
D = np.fromfile(f"{tmp}/out.bin", np.float32).reshape(N, 16, 16)
f32 = np.float32
def fl(x):                                                  # Fraction -> nearest float32 (ties to even) via float64 is NOT safe: do it exactly
    if x == 0: return f32(0)
    y = np.float64(x)                                       # float64 rounding first ...
    z = f32(y)
    # ... repair a possible double rounding: compare the exact distances of z's neighbours
    cand = [z, np.nextafter(z, f32(np.inf)), np.nextafter(z, f32(-np.inf))]
    best = min(cand, key=lambda c: (abs(Fraction(float(c)) - x), int(np.frombuffer(f32(c).tobytes(), np.uint32)[0]) & 1))
    return f32(best)
def fz(x):                                                  # Fraction -> float32 toward zero
    z = fl(x)
    if abs(Fraction(float(z))) > abs(x):
        z = np.nextafter(z, f32(0))
    return f32(z)
models = {k: 0 for k in ("seq", "exact", "grp8", "grp8_c_last", "grp4", "pair", "seq_rtz", "exact_rtz", "grp8_rtz", "grp4_rtz")}
total = 0
for c in range(N):
    for i in range(0, 16, 5):
        for j in range(0, 16, 5):
            p = [Fraction(float(A[c, i, k])) * Fraction(float(B[c, k, j])) for k in range(32)]
            c0 = Fraction(float(C[c, i, j]))
            got = D[c, i, j]
            def groups(w):
                acc = c0
                for g in range(0, 32, w):
                    acc = Fraction(float(fl(acc + sum(p[g:g + w]))))
                return f32(float(acc))
            def groups_z(w):
                acc = c0
                for g in range(0, 32, w):
                    acc = Fraction(float(fz(acc + sum(p[g:g + w]))))
                return f32(float(acc))
            res = {"seq": groups(1), "exact": fl(c0 + sum(p)), "grp8": groups(8), "grp4": groups(4), "pair": groups(2),
                   "seq_rtz": groups_z(1), "exact_rtz": fz(c0 + sum(p)), "grp8_rtz": groups_z(8), "grp4_rtz": groups_z(4)}
            acc = Fraction(0)
            for g in range(0, 32, 8):
                acc = Fraction(float(fl(acc + sum(p[g:g + 8]))))
            res["grp8_c_last"] = fl(acc + c0)
            total += 1
            for k, v in res.items():
                models[k] += int(v.tobytes() == f32(got).tobytes())
print(f"{total} outputs compared; reproduced bit for bit by:", {k: v for k, v in models.items()})
