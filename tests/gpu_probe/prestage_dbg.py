"""GPU: where do the pre-stage path (IMK_CONV_PRESTAGE=1, default) and the two-launch path differ?  Trains 3 steps (or loads
PARAMS), then dumps d8.c1 / d9.c1 (and with MAT=1 every stored layer) of one inference to OUT."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from inconsistencymasks_amd.unet import UNet
from inconsistencymasks_amd._lib import lib
h, w, c, k, alpha, loss = 64, 80, 3, 1, 0.5, 0
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randint(0, 256, (6, h, w, c), dtype=torch.uint8, device="cuda", generator=g)
y = (torch.rand((6, h, w, k), device="cuda", generator=g) > 0.6).to(torch.uint8)
m = UNet(h, w, c, k, alpha, "sigmoid", seed=11)
for _ in range(3):
    m.train_step(x, y, loss, 3e-3, 1e-4)
out = {"params": m.params.cpu().numpy()}
if os.environ.get("MAT") == "1":
    m.debug(materialize=True)
p = m.predict_device(x)
out["probs"] = p.cpu().numpy()
names = [l["name"] for l in m.plan.layers if l["kind"] == 0 and l["name"] != "out"]
for nme in (names if os.environ.get("MAT") == "1" else ["d8.c1", "d9.c1", "d7.c1"]):
    out[nme] = m.intermediate(nme, 6, 0).numpy()
np.savez(os.environ["OUT"], **out)
