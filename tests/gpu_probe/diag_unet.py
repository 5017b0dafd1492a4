"""GPU diagnostic (not a pytest): per-layer error of the HIP U-Net against the torch-CPU oracle, for
inference, the training-mode forward and the gradients.  Usage: python tests/gpu_probe/diag_unet.py [cfg ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from inconsistencymasks_amd.unet import UNet  # noqa: E402
from oracle import unet_oracle as U  # noqa: E402

CFGS = {
    "isic": dict(h=64, w=64, c=3, k=1, alpha=0.5, act="sigmoid", loss="mse", b=3),
    "suim": dict(h=48, w=64, c=3, k=9, alpha=1.0, act="softmax", loss="cce", b=2),
    "hela": dict(h=32, w=32, c=1, k=3, alpha=1.0, act="sigmoid", loss="mse", b=2),
    "odd": dict(h=48, w=80, c=3, k=35, alpha=1.25, act="softmax", loss="cce", b=2),
    "alpha2": dict(h=32, w=48, c=3, k=1, alpha=2.0, act="sigmoid", loss="mse", b=2),
    "alpha15": dict(h=32, w=48, c=3, k=1, alpha=1.5, act="sigmoid", loss="mse", b=2),
}


def randomize_bn(sd, seed):
    g = torch.Generator().manual_seed(seed)
    for k in sd:
        if k.endswith(".gamma"):
            sd[k] = 0.5 + torch.rand(sd[k].shape, generator=g)
        elif k.endswith(".beta"):
            sd[k] = 0.2 * torch.randn(sd[k].shape, generator=g)
        elif k.endswith(".mean"):
            sd[k] = 0.3 + 0.2 * torch.randn(sd[k].shape, generator=g)
        elif k.endswith(".var"):
            sd[k] = 0.2 + 0.5 * torch.rand(sd[k].shape, generator=g)
        elif k.endswith(".b"):
            sd[k] = 0.05 * torch.randn(sd[k].shape, generator=g)
    return sd


def err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    d = np.abs(a - b)
    return d.max(), np.sqrt((d ** 2).sum() / max((b ** 2).sum(), 1e-30)), np.abs(b).max()


def run(name, cfg):
    print(f"=== {name}: {cfg}")
    torch.manual_seed(0)
    h, w, c, k, alpha, act, b = cfg["h"], cfg["w"], cfg["c"], cfg["k"], cfg["alpha"], cfg["act"], cfg["b"]
    ts = os.environ.get("DIAG_TESTSEEDS") == "1"
    m = UNet(h, w, c, k, alpha, act, seed=11 if ts else 1)
    if ts:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import test_gpu_unet as T
        sd = T.randomize_bn(m.state_dict(), 12)
    else:
        sd = randomize_bn(m.state_dict(), 2)
    m.load_state_dict(sd)
    rng = np.random.default_rng(13 if ts else 3)
    yy, xx = np.mgrid[0:h, 0:w]
    x = (127 + 80 * np.sin(xx / 7.0)[None, :, :, None] * np.cos(yy / 5.0)[None, :, :, None]
         + rng.integers(-30, 30, (b, h, w, c))).clip(0, 255).astype(np.uint8)
    xd = torch.from_numpy(x).cuda()
    # ---- inference
    probs = m.predict_device(xd).cpu().numpy()
    taps = {}
    ref = U.forward(sd, x, c, k, alpha, act, training=False, emulate_fp16=True, taps=taps).numpy()
    ref32 = U.forward(sd, x, c, k, alpha, act, training=False, emulate_fp16=False).numpy()
    for l in m.plan.layers:
        if l["kind"] == 0 and l["name"] != "out":
            got = m.intermediate(l["name"], b, 0).numpy()
            e = err(got, taps[l["name"]].numpy())
            print(f"  inf {l['name']:7s} max|d|={e[0]:.4g} relL2={e[1]:.3g} max|ref|={e[2]:.3g}")
    e = err(probs, ref)
    print(f"  inf probs vs f16-oracle max|d|={e[0]:.4g} relL2={e[1]:.3g};  vs fp32-oracle max|d|={err(probs, ref32)[0]:.4g}")
    # ---- training step
    if cfg["loss"] == "mse":
        y = (rng.random((b, h, w, k)) > 0.6).astype(np.uint8)
        tgt = y.astype(np.float32)
        kind = 0
    else:
        y = rng.integers(0, k, (b, h, w)).astype(np.uint8)
        tgt = np.eye(k, dtype=np.float32)[y]
        kind = 1
    m.init_train_state()
    if os.environ.get("DIAG_SCALE"):
        import struct
        sc = float(os.environ["DIAG_SCALE"])
        off = m.plan.state_bytes - 256
        m.train_state[off:off + 8] = torch.tensor(list(struct.pack("ff", sc, 1.0 / sc)), dtype=torch.uint8).cuda()
    m.fwd_bwd(xd, torch.from_numpy(y).cuda(), kind)
    torch.cuda.synchronize()
    stats = m.stats.cpu().numpy()
    sd_ref = {kk: v.clone() for kk, v in sd.items()}
    opt = U.new_opt_state(sd_ref)
    taps = {}
    stats_out = {}
    # oracle forward in train mode for the taps
    U.forward(sd, x, c, k, alpha, act, training=True, emulate_fp16=True, stats_out=stats_out, taps=taps)
    # gradients are compared with the oracle's forward VALUES pinned to the GPU's (straight-through), so that
    # fp16 rounding noise is not amplified through ReLU masks / pool arg-maxes
    ov = {l["name"]: m.intermediate(l["name"], b, 1) for l in m.plan.layers if l["kind"] == 0 and l["name"] != "out"}
    gt = {}
    loss_ref, grads_ref = U.train_step(sd_ref, opt, x, tgt, c, k, alpha, act, cfg["loss"], emulate_fp16=True,
                                       loss_scale=float(stats[2]), return_grads=True, override=ov, grad_taps=gt)
    if os.environ.get("DIAG_DA"):
        for l in m.plan.layers:   # pre-activation gradients of the 3x3 convs (materialised on the GPU): dh * [h > 0]
            if l["kind"] == 0 and l["name"].endswith(".c3"):
                got = m.intermediate(l["name"], b, 1, which=1).numpy()
                ref = (gt[l["name"]] * (ov[l["name"]] > 0)).numpy()
                e = err(got, ref)
                print(f"  dA {l['name']:7s} max|d|={e[0]:.4g} relL2={e[1]:.3g} max|ref|={e[2]:.3g}")
    print(f"  train loss got={stats[0]:.6f} ref={loss_ref:.6f} found_inf={stats[1]} scale={stats[2]}")
    for l in m.plan.layers:
        if l["kind"] == 0 and l["name"] != "out":
            got = m.intermediate(l["name"], b, 1).numpy()
            e = err(got, taps[l["name"]].numpy())
            print(f"  trn {l['name']:7s} max|d|={e[0]:.4g} relL2={e[1]:.3g}")
    g = m.grads.cpu()
    for l in m.plan.layers:
        n = l["name"]
        if l["kind"] == 0:
            kk, ci, co = l["ksize"], l["cin"], l["cout"]
            gw = g[l["off_w"]:l["off_w"] + kk * kk * ci * co].reshape(kk, kk, ci, co).numpy()
            gb = g[l["off_b"]:l["off_b"] + co].numpy()
            ew, eb = err(gw, grads_ref[n + ".w"].numpy()), err(gb, grads_ref[n + ".b"].numpy())
            print(f"  grad {n:7s} dW relL2={ew[1]:.3g} max|ref|={ew[2]:.3g} | db relL2={eb[1]:.3g} max|ref|={eb[2]:.3g}")
        else:
            cc = l["cout"]
            gg = g[l["off_w"]:l["off_w"] + cc].numpy()
            gb = g[l["off_b"]:l["off_b"] + cc].numpy()
            eg, eb = err(gg, grads_ref[n + ".gamma"].numpy()), err(gb, grads_ref[n + ".beta"].numpy())
            print(f"  grad {n:7s} dgamma relL2={eg[1]:.3g} max|ref|={eg[2]:.3g} | dbeta relL2={eb[1]:.3g} max|ref|={eb[2]:.3g}")
    # optimizer
    m.adamw_step(3e-3, 1e-4)
    torch.cuda.synchronize()
    new = m.state_dict()
    worst = max(float((new[kk] - sd_ref[kk]).abs().max()) for kk in new)
    worst_mv = max(float((new[kk] - sd_ref[kk]).abs().max()) for kk in new if kk.endswith(".mean") or kk.endswith(".var"))
    print(f"  after AdamW: max|param diff|={worst:.4g}  (moving stats {worst_mv:.4g}), lr=3e-3")


if __name__ == "__main__":
    names = sys.argv[1:] or list(CFGS)
    for n in names:
        try:
            run(n, CFGS[n])
        except Exception as ex:  # keep going: one call should tell us as much as possible
            import traceback
            traceback.print_exc()
            print(f"!!! {n} failed: {ex}")
