"""GPU: a few training steps / inference calls for a rocprofv3 kernel trace.  CONFIG=isic|hela|suim|city (default isic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from inconsistencymasks_amd.unet import UNet
CFG = {"isic": (256, 256, 3, 1, 0.5, "sigmoid", 0), "hela": (256, 256, 1, 3, 1.0, "sigmoid", 0),
       "suim": (256, 256, 3, 9, 1.0, "softmax", 1), "city": (208, 416, 3, 35, 1.0, "softmax", 1)}
H, W, C, K, ALPHA, ACT, LOSS = CFG[os.environ.get("CONFIG", "isic")]
ALPHA = float(os.environ.get("ALPHA", ALPHA))
x = torch.randint(0, 256, (int(os.environ.get("INFER_B", 128)), H, W, C), dtype=torch.uint8, device="cuda")
if LOSS == 0:
    y = (torch.rand((32, H, W, K), device="cuda") > 0.7).to(torch.uint8)
else:
    y = torch.randint(0, K, (32, H, W), dtype=torch.uint8, device="cuda")
m = UNet(H, W, C, K, ALPHA, ACT, seed=3)
xs = x[:32].contiguous()
for _ in range(6):
    m.train_step(xs, y, LOSS, 3e-3, 1e-4)
torch.cuda.synchronize()
for _ in range(3):
    m.predict_device(x)
torch.cuda.synchronize()
