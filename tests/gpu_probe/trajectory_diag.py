"""Where do a GPU training run and the oracle's drift apart?  Per BatchNorm layer (moving statistics) and per conv (weights) after
1, 2, 5, 10, 30 steps of tests/test_gpu_unet.py's trajectory batches, GPU vs oracle and oracle vs its own half-scale run.
    python tests/gpu_probe/trajectory_diag.py [isic|suim]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_unet as T
from oracle import unet_oracle as U
from inconsistencymasks_amd.unet import UNet

name = sys.argv[1] if len(sys.argv) > 1 else "suim"
cfg = T.CFGS[name]
c, k, alpha, act = cfg["c"], cfg["k"], cfg["alpha"], cfg["act"]
m = UNet(cfg["h"], cfg["w"], c, k, alpha, act, seed=61)
sd = {kk: v.clone() for kk, v in m.state_dict().items()}
sdy = {kk: v.clone() for kk, v in sd.items()}
opt, opty = U.new_opt_state(sd), U.new_opt_state(sdy)
kind = 0 if cfg["loss"] == "mse" else 1
m.init_train_state()
rl = lambda a, b: float(np.sqrt(((a.double() - b.double()) ** 2).sum() / max(float((b.double() ** 2).sum()), 1e-30)))
for s in range(30):
    x, y, tgt = T._trajectory_batch(cfg, s)
    scale = T._ctl(m)[0]
    m.train_step(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), kind, 3e-3, 1e-4)
    torch.cuda.synchronize()
    st = m.stats.cpu().numpy()
    ok = st[1] == 0.0
    ref = U.train_step(sd, opt, x, tgt, c, k, alpha, act, cfg["loss"], emulate_fp16=True, loss_scale=scale, apply=bool(ok))
    refy = U.train_step(sdy, opty, x, tgt, c, k, alpha, act, cfg["loss"], emulate_fp16=True, loss_scale=scale / 2, apply=bool(ok))
    if s + 1 in (1, 2, 3, 5, 10, 30):
        got = {kk: v.cpu() for kk, v in m.state_dict().items()}
        print(f"--- after step {s + 1}: loss gpu {st[0]:.5f} oracle {ref:.5f} yard {refy:.5f} scale {scale} ok {ok} oracle finite {opt['last_finite']}")
        rows = []
        for kk in sd:
            if kk.endswith((".mean", ".var", ".w", ".gamma", ".beta")):
                rows.append((rl(got[kk], sd[kk]), rl(sdy[kk], sd[kk]), kk, float(sd[kk].abs().max())))
        rows.sort(reverse=True)
        for r in rows[:10]:
            print(f"   {r[2]:12s} gpu-vs-oracle {r[0]:.3e}   oracle-vs-yardstick {r[1]:.3e}   max |v| {r[3]:.3e}")
