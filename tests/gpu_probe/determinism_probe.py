"""GPU: is a training run bit-reproducible?  Trains the bench's ISIC-shaped model N steps twice in one process (and can be
started twice concurrently: python determinism_probe.py & python determinism_probe.py) and prints a checksum per run."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from inconsistencymasks_amd.unet import UNet
H = W = int(os.environ.get("SIZE", 256))
steps = int(os.environ.get("STEPS", 60))
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randint(0, 256, (259, H, W, 3), dtype=torch.uint8, device="cuda", generator=g)
y = (torch.rand((259, H, W, 1), device="cuda", generator=g) > 0.6).to(torch.uint8)
def run():
    m = UNet(H, W, 3, 1, 0.5, "sigmoid", seed=1000)
    gi = torch.Generator(device="cuda").manual_seed(5)
    sums = []
    for it in range(steps):
        idx = torch.randint(0, 259, (32,), device="cuda", generator=gi)
        m.train_step(x[idx].contiguous(), y[idx].contiguous(), 0, 3e-3, 1e-4)
        if it in (0, 1, 4, 19, steps - 1):
            sums.append(hashlib.sha1(m.params.cpu().numpy().tobytes()).hexdigest()[:10])
    return sums
a = run(); b = run()
print("run 1", a); print("run 2", b); print("identical", a == b)
