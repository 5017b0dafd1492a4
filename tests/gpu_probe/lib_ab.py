"""GPU: do two builds of the library (IMK_LIB_PATH=... for each run) compute the same bits?  Prints checksums of the inference
probabilities, of every layer's gradient after one forward/backward, and of the parameters after a step; diff the outputs."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from inconsistencymasks_amd.unet import UNet
CFG = {"isic": (256, 256, 3, 1, 0.5, "sigmoid", 0), "suim": (256, 256, 3, 9, 1.0, "softmax", 1),
       "city": (208, 416, 3, 35, 1.0, "softmax", 1)}
H, W, C, K, ALPHA, ACT, LOSS = CFG[os.environ.get("CONFIG", "isic")]
ALPHA = float(os.environ.get("ALPHA", ALPHA))
sha = lambda t: hashlib.sha1(t.detach().cpu().numpy().tobytes()).hexdigest()[:10]
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randint(0, 256, (32, H, W, C), dtype=torch.uint8, device="cuda", generator=g)
y = ((torch.rand((32, H, W, K), device="cuda", generator=g) > 0.6).to(torch.uint8) if LOSS == 0
     else torch.randint(0, K, (32, H, W), dtype=torch.uint8, device="cuda", generator=g))
m = UNet(H, W, C, K, ALPHA, ACT, seed=7)
print("probs", sha(m.predict_device(x)))
m.init_train_state()
m.fwd_bwd(x, y, LOSS)
torch.cuda.synchronize()
print("stats", m.stats.cpu().tolist())
for l in m.plan.layers:          # per layer: stored forward activation (conv layers) and the weight gradient
    try:
        print("act ", l["name"], sha(m.intermediate(l["name"], 32, 1)))
    except Exception:
        pass
    n = l["ksize"] * l["ksize"] * l["cin"] * l["cout"] if l["kind"] == 0 else l["cout"]
    print("grad", l["name"], sha(m.grads[l["off_w"]:l["off_w"] + n]))
print("grads", sha(m.grads))
m.adamw_step(3e-3, 1e-4)
print("params", sha(m.params))
