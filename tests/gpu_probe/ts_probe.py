import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from inconsistencymasks_amd.unet import UNet
from inconsistencymasks_amd._lib import lib
lib.imk_debug_timestamps.restype = ctypes.c_int
lib.imk_debug_timestamps.argtypes = [ctypes.c_void_p]
x = torch.randint(0, 256, (32, 256, 256, 3), dtype=torch.uint8, device="cuda")
y = (torch.rand((32, 256, 256, 1), device="cuda") > 0.7).to(torch.uint8)
m = UNet(256, 256, 3, 1, 0.5, "sigmoid", seed=3)
for _ in range(3): m.train_step(x, y, 0, 3e-3, 1e-4)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
print("rc", lib.imk_debug_timestamps(buf))
t = list(buf); t0 = t[0]
names = {0: "start", 1: "aff staged", 2: "first issue", 40: "loop done", 41: "partial written"}
for p in range(3):
    names.update({3 + 8 * p: f"t{p} begin", 4 + 8 * p: f"t{p} staged", 5 + 8 * p: f"t{p} barrier", 6 + 8 * p: f"t{p} next issued",
                  7 + 8 * p: f"t{p} mfma done", 8 + 8 * p: f"t{p} barrier2"})
prev = t0
for i in sorted(names):
    if t[i] and t[i] >= t0:
        print(f"{names[i]:24s} +{t[i]-t0:8d}  (d {t[i]-prev:7d})"); prev = t[i]
