"""GPU: aggregate training throughput of k independent candidate models (the reference trains 5 per generation, one after
the other: ISIC_2018/09_ISIC_2018_IM.py:90) stepped side by side on k streams, against one model alone.  CONFIG=isic|suim."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from inconsistencymasks_amd.unet import UNet
CFG = {"isic": (256, 256, 3, 1, 0.5, "sigmoid", 0), "suim": (256, 256, 3, 9, 1.0, "softmax", 1)}
H, W, C, K, ALPHA, ACT, LOSS = CFG[os.environ.get("CONFIG", "isic")]
x = torch.randint(0, 256, (32, H, W, C), dtype=torch.uint8, device="cuda")
y = ((torch.rand((32, H, W, K), device="cuda") > 0.7).to(torch.uint8) if LOSS == 0
     else torch.randint(0, K, (32, H, W), dtype=torch.uint8, device="cuda"))
def run(k, steps=40):
    models = [UNet(H, W, C, K, ALPHA, ACT, seed=i) for i in range(k)]
    streams = [torch.cuda.Stream() for _ in range(k)]
    def sweep(n):
        for _ in range(n):
            for m, s in zip(models, streams):
                with torch.cuda.stream(s):
                    m.train_step(x, y, LOSS, 3e-3, 1e-4)
    sweep(5); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); sweep(steps); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[1]
    return t / steps * 1e3, k * steps / t
base = None
for k in (1, 2, 3, 5):
    ms, rate = run(k)
    base = base or rate
    print(f"{k} model(s) side by side: {ms:.3f} ms per sweep, {rate:.0f} model-steps/s ({rate / base:.2f}x)")

# the same with one host thread per model (ctypes releases the GIL inside libimk.so: the launches of different models overlap)
import threading
def run_threads(k, steps=40):
    models = [UNet(H, W, C, K, ALPHA, ACT, seed=i) for i in range(k)]
    streams = [torch.cuda.Stream() for _ in range(k)]
    def work(m, s, n):
        with torch.cuda.stream(s):
            for _ in range(n):
                m.train_step(x, y, LOSS, 3e-3, 1e-4)
    def sweep(n):
        th = [threading.Thread(target=work, args=(m, s, n)) for m, s in zip(models, streams)]
        [t.start() for t in th]; [t.join() for t in th]
    sweep(5); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); sweep(steps); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[1]
    return t / steps * 1e3, k * steps / t
for k in (2, 3, 5):
    ms, rate = run_threads(k)
    print(f"{k} model(s), one host thread each: {ms:.3f} ms per sweep, {rate:.0f} model-steps/s ({rate / base:.2f}x)")
