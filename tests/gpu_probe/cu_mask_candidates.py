"""GPU: k candidate models training side by side (one host thread and one stream each, single-stream plans: what
im_driver.train_candidates runs) on (a) ordinary streams and (b) streams with DISJOINT compute-unit masks
(hipExtStreamCreateWithCUMask): a candidate's full-resolution kernels no longer hold every CU while another candidate's
latency-bound deep-level chain waits for a slot.  MASK=contig: bit range [i * 256 / k, (i + 1) * 256 / k) per candidate;
MASK=xcd: bits j with (j % 8) in the candidate's set of XCDs (k must divide 8 or the sets are uneven).  CONFIG=isic|suim."""
import ctypes, os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from inconsistencymasks_amd.unet import UNet
CFG = {"isic": (256, 256, 3, 1, 0.5, "sigmoid", 0), "suim": (256, 256, 3, 9, 1.0, "softmax", 1)}
H, W, C, K, ALPHA, ACT, LOSS = CFG[os.environ.get("CONFIG", "isic")]
x = torch.randint(0, 256, (32, H, W, C), dtype=torch.uint8, device="cuda")
y = ((torch.rand((32, H, W, K), device="cuda") > 0.7).to(torch.uint8) if LOSS == 0
     else torch.randint(0, K, (32, H, W), dtype=torch.uint8, device="cuda"))
hip = ctypes.CDLL("libamdhip64.so")
NCU = torch.cuda.get_device_properties(0).multi_processor_count
print("compute units:", NCU)


def masked_stream(bits):
    words = (NCU + 31) // 32
    arr = (ctypes.c_uint32 * words)()
    for b in bits:
        arr[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def streams_for(k, mode):
    if mode == "none":
        return [torch.cuda.Stream() for _ in range(k)]
    if mode == "contig":
        return [masked_stream(range(i * NCU // k, (i + 1) * NCU // k)) for i in range(k)]
    if mode == "xcd":
        sets = [[x for x in range(8) if x % k == i] for i in range(k)]
        return [masked_stream([b for b in range(NCU) if (b % 8) in sets[i]]) for i in range(k)]
    if mode == "overlap":      # every candidate may use 2 / k of the chip: its own share and its neighbour's
        return [masked_stream([b % NCU for b in range(i * NCU // k, (i + 2) * NCU // k)]) for i in range(k)]
    raise ValueError(mode)


def run_threads(k, mode, steps=60):
    models = [UNet(H, W, C, K, ALPHA, ACT, seed=i) for i in range(k)]
    for m in models:
        m.debug(single_stream=True)
    streams = streams_for(k, mode)
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
    return k * steps / t


base = run_threads(1, "none")
print(f"1 model alone (single-stream plan): {base:.0f} model-steps/s = {1e3 / base:.3f} ms per step")
for k in (2, 3, 4, 5, 8):
    row = []
    for mode in ("none", "contig", "xcd", "overlap"):
        try:
            r = run_threads(k, mode)
            row.append(f"{mode} {r:.0f} ({1e3 / r:.3f} ms/model-step, {r / base:.2f}x)")
        except Exception as e:
            row.append(f"{mode} FAILED {type(e).__name__}: {e}")
    print(f"{k} side by side: " + " | ".join(row), flush=True)
