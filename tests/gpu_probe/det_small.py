"""GPU: run-to-run determinism of training and inference at small ragged sizes (the shapes of tests/test_gpu_unet.py's
chain / pre-stage tests): every case is run twice in this process; prints a hash of the parameters and of the probabilities."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from inconsistencymasks_amd.unet import UNet
CASES = [(64, 80, 3, 1, 0.5, "sigmoid", 0), (128, 96, 1, 3, 0.5, "softmax", 1), (48, 80, 1, 2, 1.25, "sigmoid", 0),
         (48, 64, 3, 9, 1.0, "softmax", 1), (32, 48, 3, 1, 2.0, "sigmoid", 0)]
ok = True
for (h, w, c, k, alpha, act, loss) in CASES:
    res = []
    for rep in range(3):
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randint(0, 256, (6, h, w, c), dtype=torch.uint8, device="cuda", generator=g)
        y = ((torch.rand((6, h, w, k), device="cuda", generator=g) > 0.6).to(torch.uint8) if loss == 0
             else torch.randint(0, k, (6, h, w), dtype=torch.uint8, device="cuda", generator=g))
        m = UNet(h, w, c, k, alpha, act, seed=11)
        hs = []
        for _ in range(3):
            m.train_step(x, y, loss, 3e-3, 1e-4)
            hs.append(hashlib.sha1(m.grads.cpu().numpy().tobytes()).hexdigest()[:8])
        p = m.predict_device(x)
        res.append((tuple(hs), hashlib.sha1(m.params.cpu().numpy().tobytes()).hexdigest()[:8], hashlib.sha1(p.cpu().numpy().tobytes()).hexdigest()[:8]))
    same = all(r == res[0] for r in res)
    ok = ok and same
    print((h, w, c, k, alpha), "identical" if same else "DIFFERENT", res if not same else res[0])
print("ALL IDENTICAL" if ok else "NON-DETERMINISTIC")
