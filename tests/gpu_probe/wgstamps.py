"""GPU (stamps build): per-workgroup entry / exit times of one conv_pipe_kernel variant inside a training step or an inference call.
   IMK_LIB_PATH=build/stamps/libimk_stamps.so KID=41010 python tests/gpu_probe/wgstamps.py      (KID: 40000 + LM*1000 + WG*100 + CHAIN*10 + DYSTAT)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from inconsistencymasks_amd import _lib
from inconsistencymasks_amd.unet import UNet
lib = _lib.lib
kid = int(os.environ.get("KID", 41010))
x = torch.randint(0, 256, (int(os.environ.get("B", 32)), 256, 256, 3), dtype=torch.uint8, device="cuda")
y = (torch.rand((32, 256, 256, 1), device="cuda") > 0.7).to(torch.uint8)
m = UNet(256, 256, 3, 1, 0.5, "sigmoid", seed=3)
run = (lambda: m.predict_device(x)) if os.environ.get("INFER") else (lambda: m.train_step(x[:32], y, 0, 3e-3, 1e-4))
for _ in range(4):
    run()
lib.imk_debug_wgsel_conv.argtypes = [ctypes.c_uint]
lib.imk_debug_wgstamps_conv.argtypes = [ctypes.c_void_p]
assert lib.imk_debug_wgsel_conv(kid) == 0
run()
buf = np.zeros((4096, 2), dtype=np.uint64)
assert lib.imk_debug_wgstamps_conv(buf.ctypes.data) == 0
live = buf[:, 1] > 0
b, e = buf[live, 0].astype(np.int64), buf[live, 1].astype(np.int64)
t0 = b.min()
b, e = (b - t0) / 100.0, (e - t0) / 100.0
d = e - b
print(f"kid {kid}: {live.sum()} workgroups; entry min/median/max {b.min():.2f}/{np.median(b):.2f}/{b.max():.2f} us; exit min/median/p90/max "
      f"{e.min():.2f}/{np.median(e):.2f}/{np.percentile(e, 90):.2f}/{e.max():.2f} us; duration min/median/max {d.min():.2f}/{np.median(d):.2f}/{d.max():.2f}")
idx = np.nonzero(live)[0]
for g in range(8):
    sel = (idx % 8) == g
    print(f"  blocks = {g} mod 8: n {sel.sum():4d} entry median {np.median(b[sel]):6.2f} exit median {np.median(e[sel]):6.2f} max {e[sel].max():6.2f}")
order = np.argsort(e)
print("  exit-time deciles:", " ".join(f"{e[order[int(q * (len(e) - 1) / 10)]]:.1f}" for q in range(11)))
