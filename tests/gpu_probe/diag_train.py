"""GPU diagnostic: does the bench's ensemble pre-training learn the synthetic lesions?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from inconsistencymasks_amd.unet import UNet
from oracle import unet_oracle as U

dev = torch.device("cuda", 0)
x_lab, m_lab = bench.synth_images(259, 4242, dev)
y_lab = (m_lab // 255).contiguous()
print("fg fraction", float(y_lab.float().mean()))
m = UNet(256, 256, 3, 1, 0.5, "sigmoid", seed=1000, device=dev)
g = torch.Generator(device=dev).manual_seed(0)
for step in range(400):
    idx = torch.randint(0, 259, (32,), device=dev, generator=g)
    m.train_step(x_lab[idx].contiguous(), y_lab[idx].contiguous(), 0, 3e-3, 1e-4)
    if step % 50 == 0 or step == 399:
        st = m.stats.cpu().numpy()
        p = m.predict_device(x_lab[:32].contiguous())
        yy = y_lab[:32].bool()
        print(f"step {step} loss={st[0]:.4f} inf={st[1]} scale={st[2]} | inference: p_fg={float(p[yy].mean()):.3f} p_bg={float(p[~yy].mean()):.3f} "
              f"frac>0.5={float((p > 0.5).float().mean()):.3f}")
sd = m.state_dict()
for k in ("in.bn.mean", "in.bn.var", "e1.bn.mean", "e1.bn.var", "d9.bnb.mean", "d9.bnb.var", "d9.bnb.gamma", "out.w", "out.b"):
    print(k, sd[k].flatten()[:8].numpy().round(4))
# oracle in inference mode on the trained weights
ref = U.forward(sd, x_lab[:4].cpu().numpy(), 3, 1, 0.5, "sigmoid", emulate_fp16=True).numpy()
got = m.predict_device(x_lab[:4].contiguous()).cpu().numpy()
print("oracle vs gpu inference max|d|", np.abs(ref - got).max(), "oracle frac>0.5", (ref > 0.5).mean(), "true", float(y_lab[:4].float().mean()))
reft = U.forward(sd, x_lab[:32].cpu().numpy(), 3, 1, 0.5, "sigmoid", training=True, emulate_fp16=True).numpy()
print("oracle train-mode frac>0.5", (reft > 0.5).mean())
