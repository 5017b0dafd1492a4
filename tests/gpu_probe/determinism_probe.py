"""GPU: is a training run bit-reproducible?  Trains a model N steps twice in one process (and can be started twice
concurrently: python determinism_probe.py & python determinism_probe.py) and prints a checksum per run.
CONFIG=isic|hela|suim|city, ALPHA overrides the width, STEPS the length."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from inconsistencymasks_amd.unet import UNet
CFG = {"isic": (256, 256, 3, 1, 0.5, "sigmoid", 0), "hela": (256, 256, 1, 3, 1.0, "sigmoid", 0),
       "suim": (256, 256, 3, 9, 1.0, "softmax", 1), "city": (208, 416, 3, 35, 1.0, "softmax", 1)}
H, W, C, K, ALPHA, ACT, LOSS = CFG[os.environ.get("CONFIG", "isic")]
ALPHA = float(os.environ.get("ALPHA", ALPHA))
steps = int(os.environ.get("STEPS", 60))
N = 96
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randint(0, 256, (N, H, W, C), dtype=torch.uint8, device="cuda", generator=g)
if LOSS == 0:
    y = (torch.rand((N, H, W, K), device="cuda", generator=g) > 0.6).to(torch.uint8)
else:
    y = torch.randint(0, K, (N, H, W), dtype=torch.uint8, device="cuda", generator=g)
def run():
    m = UNet(H, W, C, K, ALPHA, ACT, seed=1000)
    gi = torch.Generator(device="cuda").manual_seed(5)
    sums = []
    for it in range(steps):
        idx = torch.randint(0, N, (32,), device="cuda", generator=gi)
        m.train_step(x[idx].contiguous(), y[idx].contiguous(), LOSS, 3e-3, 1e-4)
        if it in (0, 1, 4, 19, steps - 1):
            sums.append(hashlib.sha1(m.params.cpu().numpy().tobytes()).hexdigest()[:10])
    return sums
a = run(); b = run()
print(os.environ.get("CONFIG", "isic"), ALPHA, "run 1", a); print("run 2", b); print("identical", a == b)
