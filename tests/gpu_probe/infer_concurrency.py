"""GPU: does the inference stage of a generation gain from MORE concurrency than one model per stream?  2 335 images, 2 models:
(a) four sequential ensemble calls of 584 images (the bench's form), (b) two host threads with an ensemble each (own plans, own torch
stream), two calls each."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from inconsistencymasks_amd import functions as F
from inconsistencymasks_amd.unet import UNet
U, B = 2335, 584
x = torch.randint(0, 256, (U, 256, 256, 3), dtype=torch.uint8, device="cuda")
def ensemble():
    return F.EnsembleIM([UNet(256, 256, 3, 1, 0.5, "sigmoid", seed=1000 + j) for j in range(2)])
e0, e1 = ensemble(), ensemble()
chunks = [(i, min(i + B, U)) for i in range(0, U, B)]
def seq():
    for i, j in chunks: e0.run(x[i:j], 0.5, False, True, True)
def worker(e, mine, st):
    torch.cuda.set_device(0)
    with torch.cuda.stream(st):
        for i, j in mine: e.run(x[i:j], 0.5, False, True, True)
        st.synchronize()
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
def par():
    ts = [threading.Thread(target=worker, args=(e0, chunks[0::2], s0)), threading.Thread(target=worker, args=(e1, chunks[1::2], s1))]
    [t.start() for t in ts]; [t.join() for t in ts]
for name, fn in (("sequential calls", seq), ("two threads x two calls", par), ("sequential calls", seq), ("two threads x two calls", par)):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per 2 335 images")
