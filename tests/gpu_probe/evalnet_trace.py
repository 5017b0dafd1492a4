"""GPU: a few EvalNet training steps / inference calls (HeLa IM++ shape) for a rocprofv3 kernel trace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from inconsistencymasks_amd.evalnet import get_evalnet_miou
m = get_evalnet_miou(256, 256, 1, 3, float(os.environ.get("ALPHA", 2)), seed=1)
B = 32
xa = torch.randint(0, 256, (B, 256, 256, 1), dtype=torch.uint8, device="cuda")
xb = (torch.rand((B, 256, 256, 3), device="cuda") > 0.7).to(torch.uint8)
y = torch.rand((B, 6), device="cuda")
for _ in range(5):
    m.train_step(xa, xb, y, 3e-3, 1e-4)
torch.cuda.synchronize()
