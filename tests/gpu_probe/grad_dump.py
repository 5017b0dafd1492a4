"""GPU: dump the gradient vector of one training step (and the loss statistics) to OUT (npz) for the case CASE=h,w,c,k,alpha,loss."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from inconsistencymasks_amd.unet import UNet
h, w, c, k, alpha, loss = os.environ.get("CASE", "64,80,3,1,0.5,0").split(",")
h, w, c, k, loss, alpha = int(h), int(w), int(c), int(k), int(loss), float(alpha)
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randint(0, 256, (6, h, w, c), dtype=torch.uint8, device="cuda", generator=g)
y = ((torch.rand((6, h, w, k), device="cuda", generator=g) > 0.6).to(torch.uint8) if loss == 0
     else torch.randint(0, k, (6, h, w), dtype=torch.uint8, device="cuda", generator=g))
m = UNet(h, w, c, k, alpha, "sigmoid" if loss == 0 else "softmax", seed=11)
out = {}
for s in range(int(os.environ.get("STEPS", 1))):
    m.fwd_bwd(x, y, loss)
    torch.cuda.synchronize()
    out["g%d" % s] = m.grads.cpu().numpy().copy()
    out["st%d" % s] = m.stats.cpu().numpy().copy()
    m.adamw_step(3e-3, 1e-4)
out["p"] = m.params.cpu().numpy()
out["probs"] = m.predict_device(x).cpu().numpy()
names, offs = [], []
for l in m.plan.layers:
    names.append(l["name"]); offs.append([l["off_w"], l["off_b"], l["kind"], l["ksize"], l["cin"], l["cout"]])
out["names"] = np.array(names); out["offs"] = np.array(offs)
np.savez(os.environ["OUT"], **out)
