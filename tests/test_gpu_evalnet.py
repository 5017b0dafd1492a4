"""GPU parity of the HIP EvalNet (through the C ABI, imk_evalnet_*) against the torch-CPU oracle (oracle/evalnet_oracle.py,
parity unpinned: evalnet.py runs inside Keras).  Same tolerances as tests/test_gpu_unet.py:
  * stored conv outputs, layer by layer: relative L2 <= 1e-2 (inference), 3e-2 (training-mode batch statistics)
  * outputs: |dp| <= 2e-2;  losses 1e-3 relative with the oracle's forward values pinned to the GPU's
  * gradients with pinned forward values: relative L2 <= 2e-2 per tensor
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import evalnet_oracle as E  # noqa: E402
from tests.test_gpu_unet import randomize_bn, rel_l2  # noqa: E402

CFGS = {
    "isic": dict(h=64, w=64, ca=3, cb=1, k=1, alpha=1.0, two=False, na=True, nb=True, b=4),      # get_evalnet, config.ini:24
    "hela": dict(h=64, w=128, ca=1, cb=3, k=3, alpha=0.5, two=True, na=True, nb=False, b=3),     # get_evalnet_miou
    "alpha2": dict(h=64, w=64, ca=1, cb=3, k=3, alpha=2.0, two=True, na=True, nb=False, b=2),    # HeLa ALPHA_EVALNET = 2
    # SUIM: input B is the one-hot stack of a 9-class label map (functions.py:4978), passed as class ids
    "suim": dict(h=64, w=64, ca=3, cb=9, k=9, alpha=1.0, two=True, na=True, nb=False, b=3, onehot=True),
    # Cityscapes-like: 35 classes, sizes that are not multiples of 64 (104 -> 52 -> 26 -> 13 -> 6 -> 3 -> 1: MaxPooling2D drops
    # the odd last row / column twice per axis)
    "city": dict(h=104, w=208, ca=3, cb=35, k=35, alpha=0.5, two=True, na=True, nb=False, b=2, onehot=True),
}


def make(cfg, seed):
    from inconsistencymasks_amd.evalnet import EvalNet
    m = EvalNet(cfg["h"], cfg["w"], cfg["ca"], cfg["cb"], cfg["k"], cfg["alpha"], cfg["two"], cfg["na"], cfg["nb"], seed=seed,
                b_onehot=cfg.get("onehot", False))
    sd = randomize_bn(m.state_dict(), seed + 1)
    m.load_state_dict(sd)
    rng = np.random.default_rng(seed + 2)
    b, h, w = cfg["b"], cfg["h"], cfg["w"]
    yy, xx = np.mgrid[0:h, 0:w]
    xa = (127 + 80 * np.sin(xx / 7.0)[None, :, :, None] * np.cos(yy / 5.0)[None, :, :, None]
          + rng.integers(-30, 30, (b, h, w, cfg["ca"]))).clip(0, 255).astype(np.uint8)
    if cfg.get("onehot"):
        xb = rng.integers(0, cfg["cb"], (b, h // 8, w // 8, 1)).astype(np.uint8).repeat(8, 1).repeat(8, 2)
    else:
        xb = ((rng.random((b, h // 8, w // 8, cfg["cb"])) > 0.6).astype(np.uint8) * 255).repeat(8, 1).repeat(8, 2)
    y = rng.random((b, (2 if cfg["two"] else 1) * cfg["k"])).astype(np.float32)
    if cfg["two"]:
        y[:, cfg["k"]:] = (y[:, cfg["k"]:] > 0.5)
    return m, sd, xa, xb, y


def ob(cfg, xb):
    """input B as the oracle takes it"""
    return E.one_hot(xb, cfg["cb"]) if cfg.get("onehot") else xb


def test_param_count_and_order():
    from inconsistencymasks_amd.evalnet import get_evalnet, get_evalnet_miou
    m = get_evalnet(64, 64, 3, 1, alpha=1)
    assert (m.plan.n_total, m.plan.n_trainable) == E.count_params(3, 1, 1, 1, False)
    m = get_evalnet_miou(64, 64, 1, 3, alpha=2)
    assert (m.plan.n_total, m.plan.n_trainable) == E.count_params(1, 3, 3, 2, True)
    assert [l["name"] for l in m.plan.layers] == [t[0] for t in E.layer_table(1, 3, 3, 2, True)]


@pytest.mark.parametrize("name", list(CFGS))
def test_inference_parity(name):
    from inconsistencymasks_amd._lib import lib
    cfg = CFGS[name]
    m, sd, xa, xb, _ = make(cfg, 21)
    m.debug(materialize=True)
    try:
        out = m.predict([xa, xb])
    finally:
        m.debug(materialize=False)
    taps = {}
    ref, _ = E.forward(sd, xa, ob(cfg, xb), cfg["two"], cfg["na"], cfg["nb"], emulate_fp16=True, taps=taps)
    for n, t in taps.items():
        assert rel_l2(m.intermediate(n, cfg["b"], 0).numpy(), t.numpy()) <= 1e-2, n
    got = np.concatenate(out, 1) if cfg["two"] else out
    assert np.abs(got - ref.numpy()).max() <= 2e-2
    if cfg.get("onehot"):    # the reference's call sites pass the one-hot stack itself: same result
        oh = m.predict([xa, E.one_hot(xb, cfg["cb"])])
        assert np.array_equal(np.concatenate(oh, 1), got)
    # batch invariance of inference
    one = m.predict([xa[:1], xb[:1]])
    one = np.concatenate(one, 1) if cfg["two"] else one
    assert np.abs(one - got[:1]).max() <= 1e-6


@pytest.mark.parametrize("name", list(CFGS))
def test_train_pass_parity(name):
    cfg = CFGS[name]
    m, sd, xa, xb, y = make(cfg, 31)
    m.init_train_state()
    dev = lambda a: torch.from_numpy(a).cuda()
    for attempt in range(14):
        out = m.fwd_bwd(dev(xa), dev(xb), dev(y))
        torch.cuda.synchronize()
        stats = m.stats.cpu().numpy()
        assert stats[2] == 32768.0 / 2 ** attempt
        if stats[1] == 0.0:
            break
        before = m.params.clone()
        m.adamw_step(3e-3, 1e-4)
        assert torch.equal(before, m.params)
        m.load_state_dict(sd)
    assert stats[1] == 0.0
    g1 = m.grads.clone()
    moving_after_1 = m.state_dict()
    m.fwd_bwd(dev(xa), dev(xb), dev(y))
    assert torch.equal(g1, m.grads)          # deterministic
    dense = ("dense", "iou", "detection")
    ov = {l["name"]: m.intermediate(l["name"], cfg["b"], 1) for l in m.plan.layers if l["kind"] == 0 and l["name"] not in dense}
    taps = {}
    E.forward(sd, xa, ob(cfg, xb), cfg["two"], cfg["na"], cfg["nb"], training=True, emulate_fp16=True, taps=taps)
    for n, t in taps.items():
        assert rel_l2(ov[n].numpy(), t.numpy()) <= 3e-2, n
    losses, out_ref, grads_ref, bstats = E.grads(sd, xa, ob(cfg, xb), y, cfg["two"], cfg["na"], cfg["nb"], emulate_fp16=True,
                                                 loss_scale=float(stats[2]), override=ov)
    assert np.abs(out.cpu().numpy() - out_ref.numpy()).max() <= 2e-3
    assert abs(stats[0] - losses[0]) <= 1e-3 * max(1.0, abs(losses[0]))
    assert abs(stats[4] - losses[1]) <= 1e-3 * max(1.0, abs(losses[1]))
    assert abs(stats[5] - losses[2]) <= 1e-3 * max(1.0, abs(losses[2]))
    g = g1.cpu()
    errs = {}
    for l in m.plan.layers:
        n = l["name"]
        if l["kind"] == 0:
            kk, ci, co = l["ksize"], l["cin"], l["cout"]
            errs[n + ".w"] = rel_l2(g[l["off_w"]:l["off_w"] + kk * kk * ci * co].reshape(kk, kk, ci, co).numpy(), grads_ref[n + ".w"].numpy())
            errs[n + ".b"] = rel_l2(g[l["off_b"]:l["off_b"] + co].numpy(), grads_ref[n + ".b"].numpy())
        else:
            cc = l["cout"]
            errs[n + ".gamma"] = rel_l2(g[l["off_w"]:l["off_w"] + cc].numpy(), grads_ref[n + ".gamma"].numpy())
            errs[n + ".beta"] = rel_l2(g[l["off_b"]:l["off_b"] + cc].numpy(), grads_ref[n + ".beta"].numpy())
    bad = {k: round(v, 4) for k, v in errs.items() if not v <= 2e-2}
    assert not bad, f"gradient tensors off by more than 2e-2 rel-L2: {bad}; all: { {k: round(v, 4) for k, v in errs.items()} }"
    for name_, (mean, var) in bstats.items():       # moving = 0.99 * moving + 0.01 * batch
        assert torch.allclose(moving_after_1[name_ + ".mean"], 0.99 * sd[name_ + ".mean"] + 0.01 * mean, atol=1e-4, rtol=1e-3), name_
        assert torch.allclose(moving_after_1[name_ + ".var"], 0.99 * sd[name_ + ".var"] + 0.01 * var, atol=1e-4, rtol=1e-3), name_


def test_training_reduces_loss():
    cfg = CFGS["hela"]
    m, sd, xa, xb, y = make(cfg, 41)
    dev = lambda a: torch.from_numpy(a).cuda()
    losses = []
    for _ in range(60):
        m.train_step(dev(xa), dev(xb), dev(y), 3e-3, 1e-4)
        losses.append(float(m.stats[0]))
    assert np.isfinite(losses).all()
    assert np.mean(losses[-5:]) < 0.7 * np.mean(losses[:5])
    out = m.predict([xa, xb])
    assert out[0].shape == (cfg["b"], cfg["k"]) and out[1].shape == (cfg["b"], cfg["k"])


def test_full_size_hela_shape():
    """config.ini [HELA]: 256 x 256, brightfield + 3 masks, ALPHA_EVALNET = 2, batch 32 (the 32-channel towers run at full
    resolution on the per-tile conv kernel): finite, deterministic, the training-mode outputs match a second pass, inference is
    batch-invariant, and a few steps reduce the loss."""
    from inconsistencymasks_amd.evalnet import get_evalnet_miou
    m = get_evalnet_miou(256, 256, 1, 3, alpha=2, seed=5)
    g = torch.Generator(device="cuda").manual_seed(6)
    xa = torch.randint(0, 256, (32, 256, 256, 1), dtype=torch.uint8, device="cuda", generator=g)
    xb = (torch.rand((32, 256, 256, 3), device="cuda", generator=g) > 0.7).to(torch.uint8)
    y = torch.rand((32, 6), device="cuda", generator=g)
    y[:, 3:] = (y[:, 3:] > 0.5).float()
    m.init_train_state()
    for attempt in range(14):
        out1 = m.fwd_bwd(xa, xb, y).clone()
        if float(m.stats[1]) == 0.0:
            break
        m.adamw_step(3e-3, 1e-4)           # overflow: skipped, loss scale halved
    assert float(m.stats[1]) == 0.0 and torch.isfinite(m.grads).all() and torch.isfinite(out1).all()
    g1, l1 = m.grads.clone(), float(m.stats[0])
    # the BN moving statistics moved, the weights did not: a second pass gives the same outputs and gradients
    out2 = m.fwd_bwd(xa, xb, y)
    assert torch.equal(out1, out2) and torch.equal(g1, m.grads)
    assert abs(l1 - float(m.stats[4]) - float(m.stats[5])) < 1e-5
    losses = []
    for _ in range(12):
        m.train_step(xa, xb, y, 3e-3, 1e-4)
        losses.append(float(m.stats[0]))
    assert np.isfinite(losses).all() and losses[-1] < l1
    p32 = m.predict_device(xa, xb)
    p8 = m.predict_device(xa[8:16].contiguous(), xb[8:16].contiguous())
    assert torch.equal(p32[8:16], p8)
