"""GPU: inference call time + a hash of the probabilities (A/B of environment switches across processes: the hashes must agree)."""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from inconsistencymasks_amd.unet import UNet
CFG = {"isic": (256, 256, 3, 1, 0.5, "sigmoid"), "hela": (256, 256, 1, 3, 1.0, "sigmoid"),
       "suim": (256, 256, 3, 9, 1.0, "softmax"), "city": (208, 416, 3, 35, 1.0, "softmax")}
H, W, C, K, ALPHA, ACT = CFG[os.environ.get("CONFIG", "isic")]
ALPHA = float(os.environ.get("ALPHA", ALPHA))
B = int(os.environ.get("INFER_B", 256))
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.randint(0, 256, (B, H, W, C), dtype=torch.uint8, device="cuda", generator=g)
m = UNet(H, W, C, K, ALPHA, ACT, seed=3)
p = m.predict_device(x)
h = hashlib.sha1(p.cpu().numpy().tobytes()).hexdigest()[:12]
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): m.predict_device(x)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e3)
print(f"{os.environ.get('CONFIG', 'isic')} alpha {ALPHA} B={B}: {sorted(ts)[2]:.3f} ms  probs sha {h}  env",
      {k: v for k, v in os.environ.items() if k.startswith("IMK_")})
