"""GPU: a few training steps / inference calls for a rocprofv3 kernel trace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from inconsistencymasks_amd.unet import UNet
x = torch.randint(0, 256, (128, 256, 256, 3), dtype=torch.uint8, device="cuda")
y = (torch.rand((32, 256, 256, 1), device="cuda") > 0.7).to(torch.uint8)
m = UNet(256, 256, 3, 1, 0.5, "sigmoid", seed=3)
xs = x[:32].contiguous()
for _ in range(6):
    m.train_step(xs, y, 0, 3e-3, 1e-4)
torch.cuda.synchronize()
for _ in range(3):
    m.predict_device(x)
torch.cuda.synchronize()
