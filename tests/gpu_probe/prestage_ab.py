"""GPU: hashes of parameters (3 training steps) and probabilities for the shapes of test_decoder_first_stage_... (one process
per environment; compare the printed lines across IMK_CONV_PRESTAGE / IMK_CONV_GEMM / IMK_WGRAD_GEMM settings)."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from inconsistencymasks_amd.unet import UNet
g = torch.Generator(device="cuda").manual_seed(1)
for (h, w, c, k, alpha, act, loss) in [(64, 80, 3, 1, 0.5, "sigmoid", 0), (128, 96, 1, 3, 0.5, "softmax", 1), (48, 80, 1, 2, 1.25, "sigmoid", 0)]:
    x = torch.randint(0, 256, (6, h, w, c), dtype=torch.uint8, device="cuda", generator=g)
    y = ((torch.rand((6, h, w, k), device="cuda", generator=g) > 0.6).to(torch.uint8) if loss == 0
         else torch.randint(0, k, (6, h, w), dtype=torch.uint8, device="cuda", generator=g))
    m = UNet(h, w, c, k, alpha, act, seed=int(os.environ.get("SEED", 11)))
    hs = []
    for _ in range(3):
        m.train_step(x, y, loss, 3e-3, 1e-4)
        hs.append(hashlib.sha1(m.grads.cpu().numpy().tobytes()).hexdigest()[:8])
    p = m.predict_device(x)
    print((h, w, alpha), "grads", hs, "params", hashlib.sha1(m.params.cpu().numpy().tobytes()).hexdigest()[:8],
          "probs", hashlib.sha1(p.cpu().numpy().tobytes()).hexdigest()[:8])
