"""GPU: wall time of an EvalNet training step / inference call at the HeLa IM++ shape (config.ini [HELA]: 256x256, brightfield
1 channel + 3 masks, ALPHA_EVALNET = 2, BATCH_SIZE_EVALNET = 32), median of several runs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from inconsistencymasks_amd.evalnet import get_evalnet_miou
ALPHA = float(os.environ.get("ALPHA", 2))
B = 32
m = get_evalnet_miou(256, 256, 1, 3, ALPHA, seed=1)
xa = torch.randint(0, 256, (B, 256, 256, 1), dtype=torch.uint8, device="cuda")
xb = (torch.rand((B, 256, 256, 3), device="cuda") > 0.7).to(torch.uint8)
y = torch.rand((B, 6), device="cuda")
def timeit(fn, n, reps=5):
    for _ in range(3): fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(ts)[len(ts) // 2]
print("evalnet alpha", ALPHA, "params", m.plan.n_total)
print("train step B=32: %.3f ms" % timeit(lambda: m.train_step(xa, xb, y, 3e-3, 1e-4), 20))
print("inference B=32: %.3f ms" % timeit(lambda: m.predict_device(xa, xb), 20))
