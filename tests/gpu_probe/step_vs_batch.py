"""GPU: training-step wall time against the batch size (ISIC shape): T(B) = fixed (the dependent launch chain) + B x per-image work."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from inconsistencymasks_amd.unet import UNet
H = W = 256
x = torch.randint(0, 256, (128, H, W, 3), dtype=torch.uint8, device="cuda")
y = (torch.rand((128, H, W, 1), device="cuda") > 0.7).to(torch.uint8)
for B in (2, 4, 8, 16, 32, 64, 128):
    m = UNet(H, W, 3, 1, 0.5, "sigmoid", seed=3)
    xs, ys = x[:B].contiguous(), y[:B].contiguous()
    for _ in range(5): m.train_step(xs, ys, 0, 3e-3, 1e-4)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40): m.train_step(xs, ys, 0, 3e-3, 1e-4)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 40 * 1e3)
    print(f"B={B:4d}: {sorted(ts)[2]:.3f} ms per step  ({sorted(ts)[2] / B * 1e3:.1f} us per image)", flush=True)
