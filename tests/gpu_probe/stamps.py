"""GPU: where do the latency-bound launches of a training step spend their time?  Builds nothing: run
`python tests/gpu_probe/stamps.py build` here (cross-compiles build/stamps/libimk_stamps.so with -DIMK_STAMPS), then on the
GPU box `IMK_LIB_PATH=build/stamps/libimk_stamps.so python tests/gpu_probe/stamps.py`.  One thread of the first workgroup of
every stamped kernel (conv_mfma_kernel, bn_finalize_kernel, bn_bwd_coef_kernel) records the shader clock at a few points
(csrc/imk_common.h: IMK_STAMP); printed per launch of the last of a few steps, in microseconds from kernel entry."""
import ctypes, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "build", "stamps", "libimk_stamps.so")

if len(sys.argv) > 1 and sys.argv[1] == "build":
    from inconsistencymasks_amd import build as B
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    objs, procs = [], []
    for s in sorted(glob.glob(os.path.join(B.CSRC, "*.hip")) + glob.glob(os.path.join(B.CSRC, "*.cpp"))):      # (.cpp: the host PNG codec / geometry)
        o = os.path.join(os.path.dirname(OUT), os.path.basename(s) + ".o")
        objs.append(o)
        procs.append(subprocess.Popen([B.HIPCC] + B.FLAGS + ["-DIMK_STAMPS"] + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", s, "-o", o]))
    assert all(p.wait() == 0 for p in procs)
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-lz"])
    print(OUT)
    sys.exit(0)

import numpy as np
import torch
from inconsistencymasks_amd import _lib
from inconsistencymasks_amd.unet import UNet
CFG = {"isic": (256, 256, 3, 1, 0.5, "sigmoid", 0), "hela": (256, 256, 1, 3, 1.0, "sigmoid", 0),
       "suim": (256, 256, 3, 9, 1.0, "softmax", 1), "city": (208, 416, 3, 35, 1.0, "softmax", 1)}
H, W, C, K, ALPHA, ACT, LOSS = CFG[os.environ.get("CONFIG", "isic")]
ALPHA = float(os.environ.get("ALPHA", ALPHA))
lib = _lib.lib
ROWS, COLS = 4096, 16
TUS = ("conv", "elem", "gemm", "bwd1", "wgemm", "headf")
BT = int(os.environ.get("B", 32))
x = torch.randint(0, 256, (BT, H, W, C), dtype=torch.uint8, device="cuda")
y = ((torch.rand((BT, H, W, K), device="cuda") > 0.7).to(torch.uint8) if LOSS == 0
     else torch.randint(0, K, (BT, H, W), dtype=torch.uint8, device="cuda"))
m = UNet(H, W, C, K, ALPHA, ACT, seed=3)
for _ in range(5):
    m.train_step(x, y, LOSS, 3e-3, 1e-4)
torch.cuda.synchronize()
tabs = {}
for tu in TUS:
    fn = getattr(lib, "imk_debug_stamps_" + tu)
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    fn(None, 1)
if os.environ.get("INFER"):
    xi = torch.randint(0, 256, (int(os.environ["INFER"]), H, W, C), dtype=torch.uint8, device="cuda")
    m.predict_device(xi)
else:
    m.train_step(x, y, LOSS, 3e-3, 1e-4)
torch.cuda.synchronize()
rows = []
for tu in TUS:
    buf = np.zeros((ROWS, COLS), dtype=np.uint64)
    n = getattr(lib, "imk_debug_stamps_" + tu)(buf.ctypes.data, 0)
    for r in buf[:n]:
        rows.append((int(r[2]), tu, r))
rows.sort(key=lambda e: e[0])
# shader clock per microsecond: every launch carries the 100 MHz counter at entry (col 2) and at its last stamp (col 15)
num = den = 0
for rt, tu, r in rows:
    st = [int(v) for v in r[3:15] if int(v)]
    if int(r[15]) > rt:
        num += st[-1] - st[0]; den += int(r[15]) - rt
mhz = num / den * 100.0
print(f"shader clock ~{mhz:.0f} MHz; {len(rows)} stamped launches; after the grid size: us from kernel entry at each stamp")
# kernel ids: 1 bn_finalize, 2 bn_bwd_coef, 10+m bn_bwd_prep<m>, 20 bn_bwd_prep_pool, 1xxxx conv_mfma, 3xxxx wgrad_mfma (side
# stream except the step's last one), 4xxxx conv_pipe, 5xxxx conv_wide, 6xxxx conv_gemm, 7xxxx bwd1x1, 8xxxx wgrad_gemm (side), 9000x
# fused heads, 21 wgf_stage1 (side but the last), 23 pack.  "idle" = time between the end of the previous MAIN-chain
# kernel's first workgroup and this kernel's entry (head+loss, AdamW, the re-pack and the split reductions carry no stamps).
prev_end = None
idle_sum = 0.0
for rt, tu, r in rows:
    st = [int(v) for v in r[3:15] if int(v)]
    rel = [(v - st[0]) / mhz for v in st]
    kid = int(r[0])
    side = (tu == "conv" and (30000 <= kid < 40000 or kid == 21)) or tu == "wgemm"   # (the step's last wgrad / reduction: main)
    t_us = (rt - rows[0][0]) / 100.0
    idle = ""
    if not side:
        if prev_end is not None:
            idle = f"{t_us - prev_end:7.1f}"
            if t_us - prev_end > 0: idle_sum += t_us - prev_end
        prev_end = t_us + rel[-1]
    print(f"{t_us:9.1f} us  {'side' if side else 'MAIN'} idle-before {idle:>7}  {tu}:{kid:6d} grid {int(r[1]):5d}  " +
          " ".join(f"{v:6.2f}" for v in rel[1:]))
print(f"sum of main-chain idle (incl. unstamped kernels' run time): {idle_sum:.1f} us")
