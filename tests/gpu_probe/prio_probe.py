"""GPU: training step time with the main chain on a high-priority stream (the weight-gradient side stream stays at the default)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from inconsistencymasks_amd.unet import UNet
m = UNet(256, 256, 3, 1, 0.5, "sigmoid", seed=3)
x = torch.randint(0, 256, (32, 256, 256, 3), dtype=torch.uint8, device="cuda")
y = (torch.rand((32, 256, 256, 1), device="cuda") > 0.7).to(torch.uint8)
def timeit(n=40, reps=5):
    for _ in range(5): m.train_step(x, y, 0, 3e-3, 1e-4)
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): m.train_step(x, y, 0, 3e-3, 1e-4)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(ts)[len(ts) // 2]
print("default stream: %.3f ms" % timeit())
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
for prio in (-1, 0):
    s = torch.cuda.Stream(priority=prio)
    with torch.cuda.stream(s):
        print("user stream priority %d: %.3f ms" % (prio, timeit()))
