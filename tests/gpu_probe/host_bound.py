"""GPU: how far ahead of the GPU does the host run while it enqueues training steps?  (enqueue time per step vs. time per step)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from inconsistencymasks_amd.unet import UNet
m = UNet(256, 256, 3, 1, 0.5, "sigmoid", seed=3)
x = torch.randint(0, 256, (32, 256, 256, 3), dtype=torch.uint8, device="cuda")
y = (torch.rand((32, 256, 256, 1), device="cuda") > 0.7).to(torch.uint8)
for _ in range(5): m.train_step(x, y, 0, 3e-3, 1e-4)
torch.cuda.synchronize()
for n in (40, 40, 40):
    t0 = time.perf_counter()
    for _ in range(n): m.train_step(x, y, 0, 3e-3, 1e-4)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3 * (t1 - t0) / n:.3f} ms per step; total {1e3 * (t2 - t0) / n:.3f} ms per step; GPU still busy for {1e3 * (t2 - t1):.2f} ms after the last enqueue")
