"""GPU: does capturing one training step (fwd_bwd + adamw_step, two streams inside) into a HIP graph change its wall time?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from inconsistencymasks_amd.unet import UNet
H, W, C, K, ALPHA = 256, 256, 3, 1, 0.5
m = UNet(H, W, C, K, ALPHA, "sigmoid", seed=3)
x = torch.randint(0, 256, (32, H, W, C), dtype=torch.uint8, device="cuda")
y = (torch.rand((32, H, W, K), device="cuda") > 0.7).to(torch.uint8)
def step():
    m.train_step(x, y, 0, 3e-3, 1e-4)
def timeit(fn, n, reps=5):
    for _ in range(3): fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e3)
    return sorted(ts)[len(ts) // 2]
for _ in range(5): step()
torch.cuda.synchronize()
print("eager step: %.3f ms" % timeit(step, 40))
p0 = m.params.clone()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        step()
    torch.cuda.synchronize()
    print("graph step: %.3f ms" % timeit(g.replay, 40))
    # same arithmetic? one eager step vs one replayed step from the same state
    st = m.train_state.clone(); m.params.copy_(p0)
    a = UNet(H, W, C, K, ALPHA, "sigmoid", seed=3)
    print("loss after replays:", float(m.stats[0]))
except Exception as e:
    print("capture failed:", repr(e)[:300])
