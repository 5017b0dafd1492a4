"""GPU: is the training step's time data dependent?  Same loop as step_time.py on (a) noise images + random labels, (b) the
bench's synthetic images / labels, (c) as (b) but a different batch every step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from inconsistencymasks_amd.unet import UNet
name = os.environ.get("CONFIG", "suim")
cfg = bench.CONFIGS[name]
H, W, C, K, ALPHA, ACT, LOSS = cfg["h"], cfg["w"], cfg["c"], cfg["k"], cfg["alpha"], cfg["act"], cfg["loss"]
dev = torch.device("cuda:0")
xb, yb = bench.synth_images(cfg, 256, 5, dev)
if LOSS == 0:
    yb = (yb // 255).contiguous()
xn = torch.randint(0, 256, (32, H, W, C), dtype=torch.uint8, device=dev)
yn = torch.randint(0, K, (32, H, W), dtype=torch.uint8, device=dev) if LOSS else (torch.rand((32, H, W, K), device=dev) > 0.7).to(torch.uint8)
def timeit(fn, n, reps=5):
    for _ in range(3): fn(0)
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n): fn(i)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(ts)[len(ts) // 2]
for tag, fn in (("noise, one batch", lambda m: (lambda i: m.train_step(xn, yn, LOSS, 3e-3, 1e-4))),
                ("bench data, one batch", lambda m: (lambda i: m.train_step(xb[:32], yb[:32], LOSS, 3e-3, 1e-4))),
                ("bench data, 8 batches", lambda m: (lambda i: m.train_step(xb[32 * (i % 8):32 * (i % 8) + 32], yb[32 * (i % 8):32 * (i % 8) + 32], LOSS, 3e-3, 1e-4))),
                ("bench data, lr 0", lambda m: (lambda i: m.train_step(xb[:32], yb[:32], LOSS, 0.0, 0.0)))):
    m = UNet(H, W, C, K, ALPHA, ACT, seed=7)
    print("%-24s %.3f ms" % (tag, timeit(fn(m), 40)))
