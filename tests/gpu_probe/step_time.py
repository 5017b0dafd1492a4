"""GPU: wall time of a training step / an inference call (no profiling hooks), median of several runs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from inconsistencymasks_amd.unet import UNet
B = int(os.environ.get("INFER_B", 128))
x = torch.randint(0, 256, (B, 256, 256, 3), dtype=torch.uint8, device="cuda")
y = (torch.rand((32, 256, 256, 1), device="cuda") > 0.7).to(torch.uint8)
m = UNet(256, 256, 3, 1, 0.5, "sigmoid", seed=3)
xs = x[:32].contiguous()
def timeit(fn, n, reps=5):
    for _ in range(3): fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(ts)[len(ts) // 2]
print("env", {k: v for k, v in os.environ.items() if k.startswith("IMK_")})
print("train step B=32: %.3f ms" % timeit(lambda: m.train_step(xs, y, 0, 3e-3, 1e-4), 40))
print("inference B=%d: %.3f ms (%.2f us/img)" % (B, timeit(lambda: m.predict_device(x), 20), 0))
