"""GPU parity of the HIP U-Net (through the C ABI) against the torch-CPU oracle.

Tolerances (fp16 storage / fp32 accumulation on both sides; differences are summation order and 1-ulp
fp16 flips that propagate):
  * stored conv outputs, layer by layer:  relative L2 error <= 1e-2
  * probabilities vs the fp16-emulating oracle: max |dp| <= 3e-2 at default init;  vs the fp32 oracle the
    measured gap is reported (SURVEY §8c asks for |dp| <= 2e-3 "to be measured": it is the fp16 policy of the
    reference itself that does not meet that on deep random nets)
  * gradients, with the oracle's forward VALUES pinned to the GPU's (so ReLU masks / pool arg-maxes agree):
    relative L2 error <= 2e-2 per tensor;  loss <= 1e-4 relative
  * AdamW update given the GPU gradients: 1e-6;  BN moving statistics: 1e-5
"""
import struct

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import unet_oracle as U  # noqa: E402

CFGS = {
    "isic": dict(h=64, w=64, c=3, k=1, alpha=0.5, act="sigmoid", loss="mse", b=4),
    "suim": dict(h=48, w=64, c=3, k=9, alpha=1.0, act="softmax", loss="cce", b=2),
    "hela": dict(h=32, w=48, c=1, k=3, alpha=1.0, act="sigmoid", loss="mse", b=3),
    "odd": dict(h=48, w=80, c=3, k=35, alpha=1.25, act="softmax", loss="cce", b=2),
    "wide": dict(h=32, w=32, c=3, k=9, alpha=1.5, act="softmax", loss="cce", b=2),
    "alpha2": dict(h=32, w=48, c=3, k=1, alpha=2.0, act="sigmoid", loss="mse", b=2),   # 512-channel bottleneck: 2 K passes
    # 28 / 56 / ... channels: the wide persistent kernel with a partly filled second channel tile (28 of 32), half-resolution
    # rows that are not a multiple of the tile height (24), one input channel
    "alpha175": dict(h=48, w=32, c=1, k=2, alpha=1.75, act="sigmoid", loss="mse", b=2),
    "tiny16": dict(h=16, w=16, c=3, k=3, alpha=1.0, act="softmax", loss="cce", b=5),       # one tile per image, 1x1 pixels at the bottom
    # the one-pass softmax head (imk_headf.hip) at the channel strides the other cases miss: 8 (half a tile) and 32 with two class tiles
    "half_soft": dict(h=32, w=48, c=3, k=5, alpha=0.5, act="softmax", loss="cce", b=3),
    "alpha2_soft": dict(h=32, w=32, c=3, k=19, alpha=2.0, act="softmax", loss="cce", b=2),
}


@pytest.fixture(scope="module")
def UNet():
    assert torch.cuda.is_available()
    from inconsistencymasks_amd.unet import UNet as cls
    return cls


# ---- the INDEPENDENT oracle -----------------------------------------------------------------------------
# unet_oracle.forward(emulate_fp16=False) is a plain fp32 restatement of unet.py:4-67 that knows nothing about where the kernels
# round.  The fp16-emulating oracle has learnt the kernels' rounding points over the rounds (one-rounding BatchNorm affine, which
# gradients are stored); so that the two cannot drift together unnoticed, every forward parity test ALSO bounds the GPU against fp32:
#   (a) by 1.5 x what tests/gpu_probe/fp32_gap.py measured on the MI355X for that configuration and weight kind
#       (tests/measured/fp32_gap_r06.json: rel-L2, max |dp|, decision-flip rate; floors 1e-3 / 2e-3 / 1e-3), and
#   (b) by the emulation's own distance from fp32 on the same input: the GPU may not be further from fp32 than 1.5 x what fp16
#       storage alone costs (+ 5e-4) -- a change that moves kernels and emulation TOGETHER away from fp32 fails (a) or (b).
# Measured (round 6): GPU-vs-fp32 equals fp16emu-vs-fp32 within 10 % in every one of the 34 cases; default init rel-L2 <= 3.2e-3,
# max |dp| <= 5.8e-3; randomised BatchNorm parameters at full size up to rel-L2 1.3e-2, max |dp| 5.2e-2, flips 1.3 % -- the cost of
# the reference's own mixed_float16 policy on these nets (ISIC_2018/09_ISIC_2018_IM.py:16, unet.py:63), not of the kernels.
import json as _json
import os as _os
FP32_GAP = _json.load(open(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "measured", "fp32_gap_r06.json")))
FP32_FLOORS = (1e-3, 2e-3, 1e-3)


def three(p, r, act):
    flips = ((p.argmax(-1) != r.argmax(-1)) if act == "softmax" else ((p > 0.5) != (r > 0.5))).mean()
    return [rel_l2(p, r), float(np.abs(p - r).max()), float(flips)]


def check_against_fp32(key, probs, sd, x, cfg, ref16=None):
    """bounds (a) and (b) above for one (configuration, weight kind); returns the three measured numbers"""
    c, k, alpha, act = cfg["c"], cfg["k"], cfg["alpha"], cfg["act"]
    ref32 = U.forward(sd, x, c, k, alpha, act, emulate_fp16=False).numpy()
    if ref16 is None:
        ref16 = U.forward(sd, x, c, k, alpha, act, emulate_fp16=True).numpy()
    got, emu, frozen = three(probs, ref32, act), three(ref16, ref32, act), FP32_GAP[key]["gpu_vs_fp32"]
    for i, what in enumerate(("rel-L2", "max |dp|", "flip rate")):
        assert got[i] <= max(1.5 * frozen[i], FP32_FLOORS[i]), f"{key}: GPU vs fp32 oracle {what} {got[i]:.3g}, measured in round 6: {frozen[i]:.3g}"
    assert got[0] <= 1.5 * emu[0] + 5e-4, f"{key}: GPU is {got[0]:.3g} from fp32 (rel-L2), the fp16 emulation only {emu[0]:.3g}"
    assert got[1] <= 2.0 * emu[1] + 2e-3, f"{key}: GPU max |dp| {got[1]:.3g} vs fp32, the fp16 emulation {emu[1]:.3g}"
    return got


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum() / max((b ** 2).sum(), 1e-30)))


def make_input(cfg, seed=3):
    rng = np.random.default_rng(seed)
    h, w, c, b = cfg["h"], cfg["w"], cfg["c"], cfg["b"]
    yy, xx = np.mgrid[0:h, 0:w]
    x = (127 + 80 * np.sin(xx / 7.0)[None, :, :, None] * np.cos(yy / 5.0)[None, :, :, None]
         + rng.integers(-30, 30, (b, h, w, c))).clip(0, 255).astype(np.uint8)
    if cfg["loss"] == "mse":
        y = (rng.random((b, h, w, cfg["k"])) > 0.6).astype(np.uint8)
        tgt = y.astype(np.float32)
    else:
        y = rng.integers(0, cfg["k"], (b, h, w)).astype(np.uint8)
        tgt = np.eye(cfg["k"], dtype=np.float32)[y]
    return x, y, tgt


def randomize_bn(sd, seed):
    g = torch.Generator().manual_seed(seed)
    for k in sd:
        if k.endswith(".gamma"):
            sd[k] = 0.8 + 0.4 * torch.rand(sd[k].shape, generator=g)
        elif k.endswith(".beta"):
            sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
        elif k.endswith(".mean"):
            sd[k] = 0.4 + 0.1 * torch.randn(sd[k].shape, generator=g)
        elif k.endswith(".var"):
            sd[k] = 0.5 + 0.5 * torch.rand(sd[k].shape, generator=g)
        elif k.endswith(".b"):
            sd[k] = 0.05 * torch.randn(sd[k].shape, generator=g)
    return sd


@pytest.mark.parametrize("name", list(CFGS))
def test_inference_parity(UNet, name):
    cfg = CFGS[name]
    m = UNet(cfg["h"], cfg["w"], cfg["c"], cfg["k"], cfg["alpha"], cfg["act"], seed=1)
    sd = randomize_bn(m.state_dict(), 2)
    m.load_state_dict(sd)
    x, _, _ = make_input(cfg)
    from inconsistencymasks_amd._lib import lib
    xd = torch.from_numpy(x).cuda()
    taps = {}
    ref = U.forward(sd, x, cfg["c"], cfg["k"], cfg["alpha"], cfg["act"], emulate_fp16=True, taps=taps).numpy()
    # (1) the stored path: the plan's materialize switch keeps the on-chip intermediates of fused kernels (and the input block's
    #     output), so every layer can be compared
    m.debug(materialize=True)
    try:
        probs_stored = m.predict_device(xd).cpu().numpy()
        for l in m.plan.layers:
            if l["kind"] == 0 and l["name"] != "out":
                got = m.intermediate(l["name"], cfg["b"], 0).numpy()
                assert rel_l2(got, taps[l["name"]].numpy()) <= 1e-2, l["name"]
    finally:
        m.debug(materialize=False)
    # (2) the default path -- what bench.py and the writers run: the input block is computed on load by the first encoder
    #     conv (LM_STEM) where the widths allow it, Conv3x3 -> Conv1x1 pairs are chained.  Every tensor this path still
    #     stores (the block outputs *.c1 and the decoder's *.ca) is compared layer by layer, too.
    probs = m.predict_device(xd).cpu().numpy()
    for l in m.plan.layers:
        if l["kind"] == 0 and (l["name"].endswith(".c1") or l["name"].endswith(".ca")):
            got = m.intermediate(l["name"], cfg["b"], 0).numpy()
            assert rel_l2(got, taps[l["name"]].numpy()) <= 1e-2, "default path: " + l["name"]
    stem_on_load = cfg["c"] <= 4 and int(16 * cfg["alpha"]) <= 16       # imk_conv_stem_fusable
    if stem_on_load:    # same fp16 roundings, fp32 sums of <= 4 products in another order (test_fused_input_block_...)
        assert np.abs(probs - probs_stored).max() <= 2e-3
    else:               # only the chaining differs: the intermediate's fp16 rounding happens on chip, bit-identical
        assert np.array_equal(probs, probs_stored)
    assert np.abs(probs_stored - ref).max() <= 3e-2 and rel_l2(probs_stored, ref) <= 1e-2
    assert np.abs(probs - ref).max() <= 3e-2
    assert rel_l2(probs, ref) <= 1e-2
    if cfg["act"] == "softmax":
        assert np.allclose(probs.sum(-1), 1.0, atol=1e-5)
        flips = (probs.argmax(-1) != ref.argmax(-1)).mean()
    else:
        flips = ((probs > 0.5) != (ref > 0.5)).mean()
    assert flips <= 0.01, f"decision flip rate {flips}"
    # predict() (numpy in / numpy out, batches) is the same computation
    assert np.array_equal(m.predict([x], batch_size=3), probs)
    # the independent fp32 oracle: randomised BatchNorm parameters (this model) and default initialisation
    check_against_fp32(f"CFGS.{name}.random_bn", probs, sd, x, cfg, ref16=ref)
    m0 = UNet(cfg["h"], cfg["w"], cfg["c"], cfg["k"], cfg["alpha"], cfg["act"], seed=1)
    check_against_fp32(f"CFGS.{name}.default_init", m0.predict_device(xd).cpu().numpy(), m0.state_dict(), x, cfg)


def test_inference_batch_invariance(UNet):
    """Inference-mode outputs of one image do not depend on what else is in the batch (bit-exact): the
    property that makes IM masks identical for any sharding of the image set over GPUs."""
    cfg = CFGS["isic"]
    m = UNet(cfg["h"], cfg["w"], 3, 1, 0.5, "sigmoid", seed=5)
    x, _, _ = make_input(dict(cfg, b=7))
    xd = torch.from_numpy(x).cuda()
    full = m.predict_device(xd)
    for lo, hi in [(0, 1), (1, 4), (4, 7)]:
        assert torch.equal(m.predict_device(xd[lo:hi].contiguous()), full[lo:hi])


@pytest.mark.parametrize("name", list(CFGS))
def test_train_step_parity(UNet, name):
    cfg = CFGS[name]
    c, k, alpha, act, b = cfg["c"], cfg["k"], cfg["alpha"], cfg["act"], cfg["b"]
    m = UNet(cfg["h"], cfg["w"], c, k, alpha, act, seed=11)
    sd = randomize_bn(m.state_dict(), 12)
    m.load_state_dict(sd)
    x, y, tgt = make_input(cfg, 13)
    kind = 0 if cfg["loss"] == "mse" else 1
    m.init_train_state()
    for attempt in range(12):   # dynamic loss scaling: an overflowing step is skipped and the scale halved
        m.fwd_bwd(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), kind)
        torch.cuda.synchronize()
        stats = m.stats.cpu().numpy()
        assert stats[2] == 32768.0 / 2 ** attempt
        if stats[1] == 0.0:
            break
        before = m.params.clone()
        m.adamw_step(3e-3, 1e-4)
        assert torch.equal(before, m.params)
        m.load_state_dict(sd)          # undo the moving-statistics update of the skipped step
    assert stats[1] == 0.0
    g1 = m.grads.clone()
    # deterministic: a second pass from the same state gives bit-identical gradients
    moving_after_1 = m.state_dict()
    m.fwd_bwd(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), kind)
    assert torch.equal(g1, m.grads)

    ov = {l["name"]: m.intermediate(l["name"], b, 1) for l in m.plan.layers if l["kind"] == 0 and l["name"] != "out"}
    # training-mode forward parity (unpinned oracle)
    taps = {}
    U.forward(sd, x, c, k, alpha, act, training=True, emulate_fp16=True, taps=taps)
    for n, t in taps.items():   # batch statistics of tiny batches amplify 1-ulp flips: looser than inference
        assert rel_l2(ov[n].numpy(), t.numpy()) <= 3e-2, n
    sd_ref = {kk: v.clone() for kk, v in sd.items()}
    loss_ref, grads_ref = U.train_step(sd_ref, U.new_opt_state(sd_ref), x, tgt, c, k, alpha, act, cfg["loss"],
                                       emulate_fp16=True, loss_scale=float(stats[2]), return_grads=True, override=ov)
    assert abs(stats[0] - loss_ref) <= 1e-4 * max(1.0, abs(loss_ref))
    g = g1.cpu()
    errs = {}
    for l in m.plan.layers:
        n = l["name"]
        if l["kind"] == 0:
            kk, ci, co = l["ksize"], l["cin"], l["cout"]
            gw = g[l["off_w"]:l["off_w"] + kk * kk * ci * co].reshape(kk, kk, ci, co).numpy()
            gb = g[l["off_b"]:l["off_b"] + co].numpy()
            errs[n + ".w"] = rel_l2(gw, grads_ref[n + ".w"].numpy())
            errs[n + ".b"] = rel_l2(gb, grads_ref[n + ".b"].numpy())
        else:
            cc = l["cout"]
            errs[n + ".gamma"] = rel_l2(g[l["off_w"]:l["off_w"] + cc].numpy(), grads_ref[n + ".gamma"].numpy())
            errs[n + ".beta"] = rel_l2(g[l["off_b"]:l["off_b"] + cc].numpy(), grads_ref[n + ".beta"].numpy())
    # (The stem's bias gradient is the one cancellation-dominated tensor -- a sum over every pixel of the batch of ReLU-masked terms
    # whose unmasked sum is exactly zero behind the BatchNorm -- and was at 2.1e-2 while the oracle emulated the BatchNorm-on-load with
    # two roundings; the oracle now rounds once like the kernels (unet_oracle._AffineF16) and the one bound holds for every tensor.)
    bad = {k: round(v, 4) for k, v in errs.items() if not v <= 2e-2}
    assert not bad, f"gradient tensors off by more than 2e-2 rel-L2: {bad}; all: { {k: round(v, 4) for k, v in errs.items()} }"
    # BN moving statistics after one step (momentum 0.99)
    for kk in sd_ref:
        if kk.endswith(".mean") or kk.endswith(".var"):
            assert torch.allclose(moving_after_1[kk], sd_ref[kk], atol=1e-5, rtol=1e-5), kk


def test_adamw_matches_tfa_formula(UNet):
    cfg = CFGS["isic"]
    m = UNet(cfg["h"], cfg["w"], 3, 1, 0.5, "sigmoid", seed=21)
    x, y, _ = make_input(cfg, 22)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    m.init_train_state()
    n = m.plan.n_trainable
    p = m.params[:n].clone().double()
    mm = torch.zeros_like(p)
    vv = torch.zeros_like(p)
    lr, wd, b1, b2, eps = 3e-3, 1e-4, 0.9, 0.999, 1e-7
    for step in range(1, 4):
        m.fwd_bwd(xd, yd, 0)
        g = m.grads.clone().double()
        m.adamw_step(lr, wd)
        p = p * (1 - wd)
        mm = b1 * mm + (1 - b1) * g
        vv = b2 * vv + (1 - b2) * g * g
        lr_t = lr * (1 - b2 ** step) ** 0.5 / (1 - b1 ** step)
        p = p - lr_t * mm / (vv.sqrt() + eps)
        assert torch.allclose(m.params[:n].double(), p, atol=2e-6, rtol=1e-5), step
        p = m.params[:n].clone().double()      # re-sync to avoid accumulating fp32-vs-fp64 drift
        mm, vv = mm.float().double(), vv.float().double()


def test_loss_decreases_and_bn_stats_move(UNet):
    cfg = dict(CFGS["isic"], b=8)
    m = UNet(cfg["h"], cfg["w"], 3, 1, 0.5, "sigmoid", seed=31)
    x, _, _ = make_input(cfg, 32)
    y = (x[..., :1] > 140).astype(np.uint8)           # learnable target
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    losses = []
    for _ in range(40):
        m.train_step(xd, yd, 0, 3e-3, 1e-4)
        losses.append(float(m.stats[0]))
    assert losses[-1] < 0.5 * losses[0], losses[::8]
    sd = m.state_dict()
    assert float((sd["in.bn.mean"]).abs().max()) > 0


def test_overflow_skips_step_and_halves_scale(UNet):
    cfg = CFGS["isic"]
    m = UNet(cfg["h"], cfg["w"], 3, 1, 0.5, "sigmoid", seed=41)
    x, y, _ = make_input(cfg, 42)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    m.init_train_state()
    ctl_off = m.plan.state_bytes - 256
    huge = torch.tensor(list(struct.pack("ff", 2.0 ** 40, 2.0 ** -40)), dtype=torch.uint8).cuda()
    m.train_state[ctl_off:ctl_off + 8] = huge          # loss_scale, inv_loss_scale
    before = m.params.clone()
    m.fwd_bwd(xd, yd, 0)
    m.adamw_step(3e-3, 1e-4)
    torch.cuda.synchronize()
    st = m.stats.cpu().numpy()
    assert st[1] == 1.0, st
    assert torch.equal(before[:m.plan.n_trainable], m.params[:m.plan.n_trainable])
    scale = struct.unpack("f", bytes(m.train_state[ctl_off:ctl_off + 4].cpu().tolist()))[0]
    assert scale == 2.0 ** 39


def _ctl(m):
    """(loss_scale, inv_loss_scale, good_steps, step) of the model's optimizer state (ImkCtl, csrc/imk_elem.h)"""
    off = m.plan.state_bytes - 256
    return struct.unpack("ffii", bytes(m.train_state[off:off + 16].cpu().tolist()))


def test_loss_scale_doubles_after_2000_finite_steps(UNet):
    """Keras LossScaleOptimizer (dynamic, defaults): the scale grows by 2 after 2 000 consecutive finite steps and the count starts
    again.  1 999 good steps are written into the state buffer (as test_overflow_skips_step_and_halves_scale writes the scale), the
    2 000th is a real step."""
    cfg = CFGS["isic"]
    m = UNet(cfg["h"], cfg["w"], 3, 1, 0.5, "sigmoid", seed=43)
    x, y, _ = make_input(cfg, 44)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    m.init_train_state()
    off = m.plan.state_bytes - 256
    assert _ctl(m) == (32768.0, 1.0 / 32768.0, 0, 0)
    m.train_state[off + 8:off + 12] = torch.tensor(list(struct.pack("i", 1998)), dtype=torch.uint8).cuda()
    before = m.params.clone()
    m.train_step(xd, yd, 0, 3e-3, 1e-4)
    torch.cuda.synchronize()
    assert m.stats.cpu().numpy()[1] == 0.0
    assert _ctl(m) == (32768.0, 1.0 / 32768.0, 1999, 1)          # one short of the growth interval: unchanged
    m.train_step(xd, yd, 0, 3e-3, 1e-4)
    torch.cuda.synchronize()
    st = m.stats.cpu().numpy()
    assert st[1] == 0.0 and st[2] == 32768.0                        # the step itself still ran at the old scale ...
    assert _ctl(m) == (65536.0, 1.0 / 65536.0, 0, 2)               # ... the next one runs at twice that
    assert not torch.equal(before[:m.plan.n_trainable], m.params[:m.plan.n_trainable])
    m.fwd_bwd(xd, yd, 0)
    torch.cuda.synchronize()
    assert m.stats.cpu().numpy()[2] == 65536.0


def _trajectory_batch(cfg, step):
    """a learnable task: the target is a function of the image, so the loss falls over the steps"""
    x, _, _ = make_input(cfg, 1000 + step)
    if cfg["loss"] == "mse":
        y = (x[..., :1] > 140).astype(np.uint8)
        return x, y, y.astype(np.float32)
    y = np.minimum(x[..., 0].astype(np.int32) * cfg["k"] // 256, cfg["k"] - 1).astype(np.uint8)
    return x, y, np.eye(cfg["k"], dtype=np.float32)[y]


# Why the bounds below are what they are.  Thirty steps of this recipe on batches of 2-4 images are a sensitive map: batch statistics
# of a handful of images, ReLU masks and pool arg-maxes that flip on one fp16 ulp, and Adam (epsilon 1e-7, functions.py:215) dividing
# every gradient element by its own running magnitude, so that elements which are rounding noise receive full-size +-lr updates (after
# the first two applied steps of the suim case 15 % of the BatchNorm betas of the GPU and of the oracle already differ by 2 lr:
# tests/gpu_probe/trajectory_diag.py).  Yardstick, computed by the test itself on the very same batches: the fp16-EMULATING ORACLE
# AGAINST THE fp32 ORACLE -- two correct trainings that differ only in rounding.  Measured: loss curves 3.8e-2 (isic) / 1.1e-1 (suim)
# apart, weights 2.1e-1 / 3.5e-1 rel-L2.  The GPU stores what the fp16 emulation stores and must track it at least as closely as the
# emulation tracks fp32 (x 2: both numbers move from build to build -- another summation order is another trajectory -- and with the host CPU
# that runs the oracle), and stay under fixed caps in any case.  GPU values measured on the MI355X in round 5 (the test prints them), two
# builds: loss 2.6e-2 ... 3.6e-2 (isic) / 1.8e-2 ... 1.9e-2 (suim), weights 2.0e-1 / 2.8e-1, moving statistics 2.7e-1 / 5.6e-1 rel-L2; the
# yardstick on the GPU box's host: 5.2e-2 / 2.7e-2, 2.1e-1 / 2.9e-1, 2.9e-1 / 6.5e-1.
TRAJECTORY_CAP = dict(loss=1e-1, moving=1.0, weights=6e-1)


@pytest.mark.parametrize("name", ["isic", "suim"])
def test_training_trajectory_tracks_the_oracle(UNet, name):
    """SURVEY 8a' row a10: "loss after k steps within tolerance from identical init / batches" (functions.py:207-218).  30
    optimizer steps on 30 different batches, GPU (imk_unet_fwd_bwd + imk_unet_adamw_step, dynamic loss scale) against
    unet_oracle.train_step(emulate_fp16=True) fed the same batches and the same loss scale; both sides decide for themselves
    whether a step overflowed and must agree.  The loss of EVERY step, the final BatchNorm moving statistics and the final
    weights are compared.  No value of the GPU's is handed to the oracle (unlike the one-step gradient test, whose forward values
    are pinned): the end-to-end check that the two trainings stay together over a trajectory."""
    cfg = CFGS[name]
    c, k, alpha, act = cfg["c"], cfg["k"], cfg["alpha"], cfg["act"]
    m = UNet(cfg["h"], cfg["w"], c, k, alpha, act, seed=61)
    sd = {kk: v.clone() for kk, v in m.state_dict().items()}
    sd_y = {kk: v.clone() for kk, v in sd.items()}                  # the yardstick run: the same oracle in fp32
    opt, opt_y = U.new_opt_state(sd), U.new_opt_state(sd_y)
    kind = 0 if cfg["loss"] == "mse" else 1
    m.init_train_state()
    steps, skipped = 30, 0
    gpu_losses, ref_losses, y_losses = [], [], []
    for s in range(steps):
        x, y, tgt = _trajectory_batch(cfg, s)
        scale = _ctl(m)[0]
        m.train_step(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), kind, 3e-3, 1e-4)
        torch.cuda.synchronize()
        st = m.stats.cpu().numpy()
        assert st[2] == scale
        ok = st[1] == 0.0
        skipped += not ok
        # a step the GPU skipped is skipped here too; the reverse (the oracle's fp16 gradients overflow where the GPU's did not) cannot be
        # followed and fails the test
        ref = U.train_step(sd, opt, x, tgt, c, k, alpha, act, cfg["loss"], emulate_fp16=True, loss_scale=scale, apply=bool(ok))
        assert opt["last_finite"] or not ok, f"step {s}: the oracle's gradients overflow at scale {scale}, the GPU's did not"
        y_losses.append(U.train_step(sd_y, opt_y, x, tgt, c, k, alpha, act, cfg["loss"], emulate_fp16=False, apply=bool(ok)))
        gpu_losses.append(float(st[0])); ref_losses.append(ref)
    got = {kk: v.cpu() for kk, v in m.state_dict().items()}
    wn = [kk for kk in sd if kk.endswith(".w")]
    mn = [kk for kk in sd if kk.endswith(".mean") or kk.endswith(".var")]

    def gaps(losses, state):
        worst = max(abs(a - b) / max(abs(b), 1e-6) for a, b in zip(losses, ref_losses))
        mov = rel_l2(np.concatenate([state[kk].numpy().ravel() for kk in mn]), np.concatenate([sd[kk].numpy().ravel() for kk in mn]))
        wall = rel_l2(np.concatenate([state[kk].numpy().ravel() for kk in wn]), np.concatenate([sd[kk].numpy().ravel() for kk in wn]))
        return worst, mov, wall
    worst, mov, wall = gaps(gpu_losses, got)
    y_worst, y_mov, y_wall = gaps(y_losses, sd_y)
    print(f"trajectory {name}: GPU vs oracle: worst loss gap {worst:.3e}, final {abs(gpu_losses[-1] - ref_losses[-1]) / ref_losses[-1]:.3e}, moving "
          f"statistics {mov:.3e} rel-L2, weights {wall:.3e} rel-L2; fp16 oracle vs fp32 oracle: {y_worst:.3e} / {y_mov:.3e} / "
          f"{y_wall:.3e}; skipped {skipped}; gpu {np.round(gpu_losses[::5], 4).tolist()} oracle {np.round(ref_losses[::5], 4).tolist()}")
    assert skipped <= 4, f"{skipped} of {steps} steps overflowed"
    assert ref_losses[-1] < 0.8 * ref_losses[0], ref_losses[::5]                     # it did learn
    for what, got_v, yard in (("loss", worst, y_worst), ("moving", mov, y_mov), ("weights", wall, y_wall)):
        assert got_v <= TRAJECTORY_CAP[what], f"{what}: GPU vs oracle {got_v:.3e} above the cap {TRAJECTORY_CAP[what]:.1e} (fp16 vs fp32 oracle {yard:.3e})"
        assert got_v <= 2.0 * yard + 1e-3, f"{what}: GPU vs oracle {got_v:.3e}, more than 2 x what the oracle's own rounding costs ({yard:.3e})"


ENSEMBLE_CASES = {   # name -> (config, number of models)
    "isic": (CFGS["isic"], 3), "suim": (CFGS["suim"], 3), "hela": (CFGS["hela"], 3), "odd_k35": (CFGS["odd"], 2),
    "isic_256_n2": (dict(h=256, w=256, c=3, k=1, alpha=0.5, act="sigmoid", loss="mse", b=5), 2),
    "isic_n4": (dict(CFGS["isic"], b=3), 4),
    "cityscapes_208x416": (dict(h=208, w=416, c=3, k=35, alpha=1.0, act="softmax", loss="cce", b=2), 2),
}


@pytest.mark.parametrize("case", list(ENSEMBLE_CASES))
def test_ensemble_forward_im_matches_unfused(UNet, case):
    """imk_unet_forward_im (the head evaluated inside the IM kernel, no probability stack) against imk_unet_forward's
    probabilities pushed through the oracle's IM chain: every output bit-identical, i.e. the fused kernel computes the very
    same fp32 probabilities as head_kernel and votes on them like functions.py:3104-3137, 3157, 3187, 3225."""
    from inconsistencymasks_amd import functions as F
    from inconsistencymasks_amd import im as imk_im
    from oracle import im_oracle as O
    for name, (cfg, n_models) in [(case, ENSEMBLE_CASES[case])]:
        models = [UNet(cfg["h"], cfg["w"], cfg["c"], cfg["k"], cfg["alpha"], cfg["act"], seed=50 + j) for j in range(n_models)]
        if case in ("isic_256_n2", "isic_n4"):      # decisive, disagreeing predictions instead of ~0.5 everywhere
            for j, mm in enumerate(models):
                mm.load_state_dict(randomize_bn(mm.state_dict(), 70 + j))
        x, _, _ = make_input(cfg, 51)
        xd = torch.from_numpy(x).cuda()
        r = F.EnsembleIM(models).run(xd, 0.5, name == "hela", True, True, want_presence=True)
        probs = torch.stack([mm.predict_device(xd) for mm in models], 0)
        pn = probs.cpu().numpy()
        for i in range(cfg["b"]):
            if cfg["act"] == "sigmoid":
                e = O.im_binary(pn[:, i], 0.5, name == "hela")
                eimg, emasks = O.block(x[i], list(e["final"]), e["im"], True, True)
                assert np.array_equal(r["masks"][i].cpu().numpy(), np.stack(emasks))
                assert r["im_size"][i].cpu().tolist() == e["im_size_ch"].tolist()
                assert r["pred_size"][i].cpu().tolist() == e["pred_size_ch"].tolist()
            else:
                e = O.im_multiclass(pn[:, i])
                eimg, (ef,) = O.block(x[i], [e["final"]], e["im"], True, True)
                assert np.array_equal(r["masks"][i, 0].cpu().numpy(), ef)
                assert int(r["im_size"][i, 0]) == int(e["im_size"])
                assert np.array_equal(r["presence"][:, i].cpu().numpy(), e["presence"])
            assert np.array_equal(r["im"][i].cpu().numpy(), e["im"])
            assert np.array_equal(r["img_out"][i].cpu().numpy(), eimg)
        n_im = int(r["im_size"].sum())
        assert n_im > 0
        if case in ("isic_256_n2", "isic_n4"):
            assert n_im < cfg["b"] * cfg["h"] * cfg["w"] and int(r["pred_size"].sum()) > 0    # neither all-agree nor all-differ


BASELINE_SHAPES = {   # BASELINE.json configs at their real sizes (config.ini:18-26, 39-48, 61-69, 82-91 of the reference)
    "isic_256": dict(h=256, w=256, c=3, k=1, alpha=0.5, act="sigmoid", loss="mse", b=2),
    "hela_256": dict(h=256, w=256, c=1, k=3, alpha=1.0, act="sigmoid", loss="mse", b=2),
    "suim_256": dict(h=256, w=256, c=3, k=9, alpha=1.0, act="softmax", loss="cce", b=2),
    "cityscapes_208x416": dict(h=208, w=416, c=3, k=35, alpha=1.0, act="softmax", loss="cce", b=2),
    # BASELINE configs[3]: the Cityscapes IM+ width schedule alpha = 1 ... 2 (Cityscapes/11_Cityscapes_IM+.py:48), and the
    # literal 512 x 256 of the config's name
    "cityscapes_implus_a1.25": dict(h=208, w=416, c=3, k=35, alpha=1.25, act="softmax", loss="cce", b=2),
    "cityscapes_implus_a2": dict(h=208, w=416, c=3, k=35, alpha=2.0, act="softmax", loss="cce", b=2),
    "cityscapes_256x512": dict(h=256, w=512, c=3, k=35, alpha=1.0, act="softmax", loss="cce", b=2),
}


@pytest.mark.parametrize("name", list(BASELINE_SHAPES))
def test_baseline_shapes_full_size(UNet, name):
    """Forward parity at the real sizes of the four datasets + one finite training step."""
    cfg = BASELINE_SHAPES[name]
    c, k, alpha, act = cfg["c"], cfg["k"], cfg["alpha"], cfg["act"]
    m = UNet(cfg["h"], cfg["w"], c, k, alpha, act, seed=61)
    sd = randomize_bn(m.state_dict(), 62)
    m.load_state_dict(sd)
    x, y, _ = make_input(cfg, 63)
    xd = torch.from_numpy(x).cuda()
    probs = m.predict_device(xd).cpu().numpy()
    ref = U.forward(sd, x, c, k, alpha, act, emulate_fp16=True).numpy()
    assert rel_l2(probs, ref) <= 1e-2
    flips = ((probs.argmax(-1) != ref.argmax(-1)) if act == "softmax" else ((probs > 0.5) != (ref > 0.5))).mean()
    assert flips <= 0.01, f"decision flip rate {flips}"
    # the independent fp32 oracle at the real sizes, both weight kinds (see check_against_fp32)
    check_against_fp32(f"BASELINE_SHAPES.{name}.random_bn", probs, sd, x, cfg, ref16=ref)
    m0 = UNet(cfg["h"], cfg["w"], c, k, alpha, act, seed=61)
    check_against_fp32(f"BASELINE_SHAPES.{name}.default_init", m0.predict_device(xd).cpu().numpy(), m0.state_dict(), x, cfg)
    del m0
    before = m.params.clone()
    for _ in range(6):       # the dynamic loss scale may need a few halvings at these sizes
        m.train_step(xd, torch.from_numpy(y).cuda(), 0 if cfg["loss"] == "mse" else 1, 3e-3, 1e-4)
        if float(m.stats[1]) == 0.0:
            break
    st = m.stats.cpu().numpy()
    assert st[1] == 0.0 and np.isfinite(st[0]), st
    assert not torch.equal(before, m.params) and bool(torch.isfinite(m.params).all())


@pytest.mark.parametrize("alpha", [1.25, 1.5, 1.75, 2.0])
def test_cityscapes_im_plus_schedule_batch32(UNet, alpha):
    """The IM+ generations of BASELINE configs[3] at the reference's real size and batch: 208 x 416 x 3 -> 35 classes,
    alpha growing per generation (Cityscapes/11_Cityscapes_IM+.py:48 ALPHAS), batch 32 (config.ini:8): the workspace fits,
    a batch-32 forward is invariant to the batch composition, and training steps are finite and move the loss."""
    h, w, k, b = 208, 416, 35, 32
    m = UNet(h, w, 3, k, alpha, "softmax", seed=int(alpha * 100))
    ws_train, ws_inf = m.plan.workspace_bytes(b, 1), m.plan.workspace_bytes(b, 0)
    assert 0 < ws_inf < ws_train < 48 * 2 ** 30, (ws_inf, ws_train)       # a sixth of one MI355X's 288 GB at most
    rng = np.random.default_rng(7)
    yy, xx = np.mgrid[0:h, 0:w]
    y = ((yy // 26) * 5 + xx // 84).astype(np.uint8) % k                     # a coarse class layout
    y = np.broadcast_to(y, (b, h, w)).copy()
    x = (y[..., None] * 7 + rng.integers(0, 30, (b, h, w, 3))).clip(0, 255).astype(np.uint8)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    full = m.predict_device(xd)
    assert bool(torch.isfinite(full).all()) and torch.allclose(full.sum(-1), torch.ones_like(full[..., 0]), atol=1e-5)
    assert torch.equal(m.predict_device(xd[5:9].contiguous()), full[5:9])
    losses = []
    for _ in range(12):
        m.train_step(xd, yd, 1, 3e-3, 1e-4)
        st = m.stats.cpu().numpy()
        if st[1] == 0.0:
            losses.append(float(st[0]))
    assert len(losses) >= 8 and all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    assert bool(torch.isfinite(m.params).all())


_WGRAD_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, {root!r})
from inconsistencymasks_amd.unet import UNet
alpha = float(sys.argv[2])
h, w, k, b = 208, 416, 35, 32
g = torch.Generator(device="cuda").manual_seed(3)
x = torch.randint(0, 256, (b, h, w, 3), dtype=torch.uint8, device="cuda", generator=g)
y = torch.randint(0, k, (b, h, w), dtype=torch.uint8, device="cuda", generator=g)
m = UNet(h, w, 3, k, alpha, "softmax", seed=17)
m.fwd_bwd(x, y, 1)
torch.cuda.synchronize()
np.save(sys.argv[1], m.grads.cpu().numpy())
"""


@pytest.mark.parametrize("alpha,env", [(2.0, {"IMK_WGRAD_NFO2": "0"}), (1.25, {"IMK_WGRAD_GEMM_MIN": "33"})])
def test_full_resolution_weight_gradient_forms_agree(tmp_path, alpha, env):
    """The full-resolution 3x3 weight gradients of the wide configurations take kernel forms that only exist above 2 M pixels
    (wgrad_gemm_kernel<.., 2, 2>: two output tiles, 768 splits; the GEMM-class kernel from 24 channels): against the forms the
    small parity cases exercise (four output tiles / the per-pair kernel), on the same batch-32 Cityscapes step, every gradient
    tensor agrees to fp32 summation-order noise."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "wgrad_child.py"
    script.write_text(_WGRAD_CHILD.format(root=root))
    out = []
    for i, extra in enumerate(({}, env)):
        f = tmp_path / f"g{i}.npy"
        r = subprocess.run([sys.executable, str(script), str(f), str(alpha)], env={**os.environ, **extra}, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out.append(np.load(f))
    a, b = out
    assert np.isfinite(a).all() and np.abs(a).max() > 0
    assert np.linalg.norm(a - b) <= 1e-4 * np.linalg.norm(b), np.linalg.norm(a - b) / np.linalg.norm(b)


@pytest.mark.parametrize("name", ["isic", "hela"])
def test_fused_input_block_matches_stored_one(UNet, name):
    """Inference computes the input block (x/255 -> Conv1x1+ReLU -> BN) inside the first encoder conv's load (LM_STEM) instead of
    storing its output; the plan's materialize switch runs the stored path.  Same fp16 roundings, fp32 sums of <= 4 products in a
    different order: probabilities agree to 2e-3."""
    from inconsistencymasks_amd._lib import lib
    cfg = CFGS[name]
    m = UNet(cfg["h"], cfg["w"], cfg["c"], cfg["k"], cfg["alpha"], cfg["act"], seed=5)
    m.load_state_dict(randomize_bn(m.state_dict(), 6))
    x, _, _ = make_input(cfg, 7)
    fused = m.predict(x)
    m.debug(materialize=True)
    try:
        stored = m.predict(x)
    finally:
        m.debug(materialize=False)
    assert np.abs(fused - stored).max() <= 2e-3
    assert (np.abs(fused - stored) > 0).mean() < 0.5 or np.abs(fused - stored).max() <= 1e-3


def test_full_size_ensemble_im_properties(UNet):
    """BASELINE configs[1] at its FULL size (2 335 images of 256 x 256 x 3, 2 models, alpha 0.5) through the product
    path (`EnsembleIM.run`, batches of 256), checked by size-independent properties instead of the oracle (which would need
    minutes): the fused head + IM kernel equals `imk_unet_forward` + `imk_im_binary` bit for bit on every image; N = 2
    binary IM == XOR of the two votes; label map and IM are disjoint under output blocking; sizes are pixel counts; the
    result does not depend on how the set is cut into batches / rank shards (functions.shard_list blocks of 8 ranks)."""
    from inconsistencymasks_amd import functions as F
    from inconsistencymasks_amd import im as imk_im
    U_, H_, W_ = 2335, 256, 256
    models = [UNet(H_, W_, 3, 1, 0.5, "sigmoid", seed=90 + j) for j in range(2)]
    for j, mm in enumerate(models):
        mm.load_state_dict(randomize_bn(mm.state_dict(), 95 + j))
    g = torch.Generator(device="cuda").manual_seed(9)
    yy = torch.arange(H_, device="cuda").view(1, H_, 1, 1)
    x = (torch.randint(0, 64, (U_, H_, W_, 3), device="cuda", generator=g) + (yy // 2)).clamp(0, 255).to(torch.uint8)
    ens = F.EnsembleIM(models)
    def run(lo, hi):
        r = ens.run(x[lo:hi], 0.5, False, True, True)
        return {k: r[k].clone() for k in ("img_out", "masks", "im", "im_size", "pred_size")}
    whole = [run(i, min(i + 256, U_)) for i in range(0, U_, 256)]
    cat = {k: torch.cat([w[k] for w in whole], 0) for k in whole[0]}
    n_im, n_fg = int(cat["im_size"].sum()), int(cat["pred_size"].sum())
    assert 0 < n_im < U_ * H_ * W_ and n_fg > 0
    # (1) fused == unfused, on a sample of batches (the unfused route materialises 2 x 256 x 256 KB of probabilities)
    for lo in (0, 1024, 2304):
        hi = min(lo + 256, U_)
        probs = torch.stack([mm.predict_device(x[lo:hi]) for mm in models], 0)
        u = imk_im.im_binary(probs, 0.5, False, x[lo:hi], True, True)
        for k in ("img_out", "masks", "im", "im_size", "pred_size"):
            assert torch.equal(u[k], cat[k][lo:hi]), (k, lo)
        v = probs[..., 0] > 0.5
        assert torch.equal(cat["im"][lo:hi] > 0, v[0] ^ v[1])                      # (2) N = 2: the IM is the XOR of the votes
        assert torch.equal(cat["pred_size"][lo:hi, 0], (v[0] & v[1]).sum(dim=(1, 2)))
    # (3) structure
    assert not bool(((cat["masks"][:, 0] > 0) & (cat["im"] > 0)).any())
    assert torch.equal(cat["im_size"][:, 0], (cat["im"] > 0).sum(dim=(1, 2)))
    assert torch.equal(cat["img_out"], x * (cat["im"] == 0)[..., None])
    # (4) any cut of the set gives the same masks: the 8-rank shards of functions.shard_list
    for rank in range(8):
        lo, hi = (U_ * rank) // 8, (U_ * (rank + 1)) // 8
        part = run(lo, hi)
        for k in part:
            assert torch.equal(part[k], cat[k][lo:hi]), (k, rank)


def test_randomized_ensembles_fused_vs_unfused(UNet):
    """14 random ensembles (seeded): 2-5 models, alpha 0.25-1.5, 1-4 sigmoid maps or 2-40 classes, ragged (multiple-of-16)
    sizes, `>` and `>=`, every blocking combination -- `imk_unet_forward_im` (fused head + IM) against the probabilities
    of `imk_unet_forward` pushed through the oracle's IM chain, bit for bit."""
    from inconsistencymasks_amd import functions as F
    from oracle import im_oracle as O
    rng = np.random.default_rng(77)
    for case in range(14):
        n = int(rng.integers(2, 6))
        h, w = 16 * int(rng.integers(1, 5)), 16 * int(rng.integers(1, 6))
        c = int(rng.choice([1, 3]))
        alpha = float(rng.choice([0.25, 0.5, 1.0, 1.25, 1.5]))
        soft = bool(case % 2)
        k = int(rng.integers(2, 41)) if soft else int(rng.integers(1, 5))
        b = int(rng.integers(1, 4))
        ge, bi, bo = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        models = [UNet(h, w, c, k, alpha, "softmax" if soft else "sigmoid", seed=1000 * case + j) for j in range(n)]
        for j, mm in enumerate(models):
            mm.load_state_dict(randomize_bn(mm.state_dict(), 5000 + 10 * case + j))
        x = rng.integers(0, 256, (b, h, w, c)).astype(np.uint8)
        xd = torch.from_numpy(x).cuda()
        r = F.EnsembleIM(models).run(xd, 0.5, ge, bi, bo, want_presence=True)
        pn = torch.stack([mm.predict_device(xd) for mm in models], 0).cpu().numpy()
        for i in range(b):
            if soft:
                e = O.im_multiclass(pn[:, i])
                eimg, (ef,) = O.block(x[i], [e["final"]], e["im"], bi, bo)
                assert np.array_equal(r["masks"][i, 0].cpu().numpy(), ef), case
                assert int(r["im_size"][i, 0]) == int(e["im_size"]), case
                assert np.array_equal(r["presence"][:, i].cpu().numpy(), e["presence"]), case
            else:
                e = O.im_binary(pn[:, i], 0.5, ge)
                eimg, emasks = O.block(x[i], list(e["final"]), e["im"], bi, bo)
                assert np.array_equal(r["masks"][i].cpu().numpy(), np.stack(emasks)), case
                assert r["im_size"][i].cpu().tolist() == e["im_size_ch"].tolist(), case
                assert r["pred_size"][i].cpu().tolist() == e["pred_size_ch"].tolist(), case
            assert np.array_equal(r["im"][i].cpu().numpy(), e["im"]), case
            assert np.array_equal(r["img_out"][i].cpu().numpy(), eimg), case


def test_training_run_is_bit_reproducible(UNet):
    """Two runs of 80 training steps from the same seed give bit-identical parameters (no float atomics, fixed reduction
    orders) -- the property that makes the ranks of a multi-GPU run build identical ensembles.  At 256 x 256 on purpose:
    a missing barrier in the fused weight-gradient epilogue (found by bench.py's sharding check in round 2) only showed
    with many tiles per workgroup."""
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randint(0, 256, (96, 256, 256, 3), dtype=torch.uint8, device="cuda", generator=g)
    y = (torch.rand((96, 256, 256, 1), device="cuda", generator=g) > 0.6).to(torch.uint8)
    def run():
        m = UNet(256, 256, 3, 1, 0.5, "sigmoid", seed=1000)
        gi = torch.Generator(device="cuda").manual_seed(5)
        for _ in range(80):
            idx = torch.randint(0, 96, (32,), device="cuda", generator=gi)
            m.train_step(x[idx].contiguous(), y[idx].contiguous(), 0, 3e-3, 1e-4)
        return m.params.clone()
    a, b = run(), run()
    assert torch.equal(a, b)


_CHAIN_CHILD = r"""
import hashlib, sys, torch
sys.path.insert(0, {root!r})
from inconsistencymasks_amd.unet import UNet
g = torch.Generator(device="cuda").manual_seed(1)
out = []
for (h, w, c, k, alpha, act, loss) in [(64, 80, 3, 1, 0.5, "sigmoid", 0), (48, 64, 3, 9, 1.0, "softmax", 1), (48, 80, 1, 2, 1.25, "sigmoid", 0)]:
    x = torch.randint(0, 256, (6, h, w, c), dtype=torch.uint8, device="cuda", generator=g)
    y = ((torch.rand((6, h, w, k), device="cuda", generator=g) > 0.6).to(torch.uint8) if loss == 0
         else torch.randint(0, k, (6, h, w), dtype=torch.uint8, device="cuda", generator=g))
    m = UNet(h, w, c, k, alpha, act, seed=11)
    for _ in range(3):
        m.train_step(x, y, loss, 3e-3, 1e-4)
    p = m.predict_device(x)
    out.append(hashlib.sha1(m.params.cpu().numpy().tobytes() + p.cpu().numpy().tobytes()).hexdigest())
print("SHA", " ".join(out))
"""


def test_chained_per_tile_conv_is_bit_identical_to_two_launches(tmp_path):
    """conv_mfma_kernel<..., CHAIN> (Conv3x3+ReLU -> Conv1x1+ReLU of the mid / deep blocks in one launch) against the same
    kernel launched twice: same parameters after 3 training steps and same probabilities, bit for bit, at ragged sizes and
    at widths 0.5 / 1 / 1.25 (one process each: the switch is read once; the 17-32 channel kernel is off in both so that
    the second conv runs on the per-tile kernel either way; the GEMM-class kernel of the wide layers likewise: its
    BatchNorm statistics are summed in another order)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "chain_child.py"
    script.write_text(_CHAIN_CHILD.format(root=root))
    got = []
    for mode in ("0", "2"):
        env = {**os.environ, "IMK_CONV_WIDE": "0", "IMK_CONV_GEMM": "0", "IMK_CONV_CHAIN_TILE": mode}
        r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        got.append([l for l in r.stdout.splitlines() if l.startswith("SHA")][-1])
    assert got[0] == got[1]


_CHAIN3_CHILD = r"""
import hashlib, sys, numpy as np, torch
sys.path.insert(0, {root!r})
from inconsistencymasks_amd.unet import UNet
g = torch.Generator(device="cuda").manual_seed(2)
h = hashlib.sha256()
params = []
for (hh, ww, c, k, alpha, act) in [(64, 80, 3, 9, 1.0, "softmax"), (48, 64, 3, 5, 2.0, "softmax"), (80, 48, 1, 3, 1.0, "sigmoid"),
                                   (64, 64, 3, 4, 1.5, "softmax")]:
    x = torch.randint(0, 256, (5, hh, ww, c), dtype=torch.uint8, device="cuda", generator=g)
    m = UNet(hh, ww, c, k, alpha, act, seed=11)
    h.update(m.predict_device(x).cpu().numpy().tobytes())
    # training: the GEMM-class chain stores the intermediate and takes the BatchNorm statistics of the 1x1's output
    y = (torch.randint(0, k, (5, hh, ww), dtype=torch.uint8, device="cuda", generator=g) if act == "softmax"
         else (torch.rand((5, hh, ww, k), device="cuda", generator=g) > 0.6).to(torch.uint8))
    m.fwd_bwd(x, y, 1 if act == "softmax" else 0)      # (gradients, not parameters after Adam steps: Adam's first updates are
    torch.cuda.synchronize()                           #  lr * sign(g), which turns rounding noise of near-zero gradients into +-lr)
    params.append(m.grads.cpu().numpy())
print("SHA", h.hexdigest())
np.save(sys.argv[1], np.concatenate(params))
"""


def test_chained_wide_and_gemm_convs_are_bit_identical_to_two_launches(tmp_path):
    """Round 3's inference chains -- conv_wide_kernel<..., CHAIN2> (17-32 channel blocks) and conv_gemm_kernel<..., CH2> (blocks up
    to 128 channels): the block's Conv1x1 computed from the 3x3's output tile while it is on the chip, with the 1x1's regular
    pack in its regular k order -- against the same convs as two launches each: identical probabilities, bit for bit; and for the
    GEMM-class chain in TRAINING (intermediate stored, statistics of the 1x1's output) the same gradients, bit for bit.  Ragged sizes, widths 1 / 1.5 / 2 (one process each: the switches are read once)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "chain3_child.py"
    script.write_text(_CHAIN3_CHILD.format(root=root))
    got, par = [], []
    for i, (wide, gemm) in enumerate((("1", "1"), ("0", "0"), ("1", "0"))):
        # (the 17-32 channel chain in TRAINING writes its statistics rows on another launch grid than two launches do -- summation
        #  order, covered by the oracle parity cases -- so it is off here: the training comparison is about the GEMM-class chain)
        env = {**os.environ, "IMK_WIDE_CHAIN": wide, "IMK_GEMM_CHAIN": gemm, "IMK_WIDE_CHAIN_TRAIN": "0"}
        if gemm == "0":
            env["IMK_GEMM_OVER_CHAIN"] = "2"       # the 3x3 and the 1x1 as two GEMM-class launches (not the per-tile chain)
        f = tmp_path / f"params{i}.npy"
        r = subprocess.run([sys.executable, str(script), str(f)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        got.append([l for l in r.stdout.splitlines() if l.startswith("SHA")][-1])
        par.append(np.load(f))
    assert got[0] == got[1] == got[2], got
    # Training: the chain's BatchNorm statistics are per-workgroup sums over the same 128 pixels in the same order as the 1x1's own
    # launch makes them (a 64-wide 1x1 behind a 128-wide 3x3 sweeps its tile the 64-wide way): gradients bit for bit.
    assert np.array_equal(par[1], par[2]) and np.array_equal(par[0], par[1])


_PRE_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, {root!r})
from inconsistencymasks_amd.unet import UNet
g = torch.Generator(device="cuda").manual_seed(1)
out = {{}}
for i, (h, w, c, k, alpha, act, loss) in enumerate([(64, 80, 3, 1, 0.5, "sigmoid", 0), (128, 96, 1, 3, 0.5, "softmax", 1), (48, 80, 1, 2, 1.25, "sigmoid", 0)]):
    x = torch.randint(0, 256, (6, h, w, c), dtype=torch.uint8, device="cuda", generator=g)
    y = ((torch.rand((6, h, w, k), device="cuda", generator=g) > 0.6).to(torch.uint8) if loss == 0
         else torch.randint(0, k, (6, h, w), dtype=torch.uint8, device="cuda", generator=g))
    m = UNet(h, w, c, k, alpha, act, seed=int(sys.argv[2]))
    for _ in range(3):
        m.train_step(x, y, loss, 3e-3, 1e-4)
    out["params%d" % i] = m.params.cpu().numpy()
    out["probs%d" % i] = m.predict_device(x).cpu().numpy()
    out["last%d" % i] = m.intermediate("d9.c1", 6, 0).numpy()
np.savez(sys.argv[1], **out)
"""


def test_decoder_first_stage_matches_its_own_launch(tmp_path):
    """conv_pipe_kernel<..., PRE> (inference: a shallow decoder block's Conv1x1 on upsample + skip computed on the matrix
    cores inside the block's 3x3 launch) against the two launches (alpha = 0.5 widths, where the pair layout applies; ragged
    and full-tile sizes; one process per setting: the switch is read once).  Training is untouched (parameters bit-identical).
    BIT-IDENTICAL since round 4, on every seed.  Rounds 2-3 saw a handful of the last decoder block's outputs differ by one
    fp16 ulp on some seeds (12-16): the first stage's BatchNorm `fp16(z * sc + sh)` had been compiled to v_fma_mixlo_f16 -- ONE
    rounding of the exact sum -- in the fused launch, while the staged form of the two launches rounds to fp32 and then to fp16;
    1e-4 of the values land on an fp16 tie of the fp32 result and differ (tests/gpu_probe/pre_dump.py dumps both and names the
    pixels).  Every BatchNorm-on-load site now goes through one helper (imk_common.h: the one-rounding instruction written out)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "pre_child.py"
    script.write_text(_PRE_CHILD.format(root=root))
    for seed in (11, 13, 14, 15, 16):
        got = []
        for mode in ("0", "1"):
            out = tmp_path / f"pre_{seed}_{mode}.npz"
            r = subprocess.run([sys.executable, str(script), str(out), str(seed)], env={**os.environ, "IMK_CONV_PRESTAGE": mode},
                               capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            got.append(np.load(out))
        for i in range(3):
            assert np.array_equal(got[0]["params%d" % i], got[1]["params%d" % i])
            assert np.array_equal(got[0]["last%d" % i], got[1]["last%d" % i]), (seed, i, int((got[0]["last%d" % i] != got[1]["last%d" % i]).sum()))
            assert np.array_equal(got[0]["probs%d" % i], got[1]["probs%d" % i]), (seed, i)


@pytest.mark.parametrize("config,alpha", [("suim", "1"), ("city", "2")])
def test_conv_gemm_look_ahead_forms_agree(config, alpha):
    """The GEMM-class kernel has two forms of its k-loop (csrc/imk_gemm.hip: two fragment buffers for launches that fill the chip, a ring
    of three / four with a straight-line 9-step stage for launches of at most IMK_GEMM_AD3_WGS = 1 024 workgroups).  Same arithmetic,
    same k order: forced to one form and to the other (a child process each: the switch is read once), a batch-32 step at real size must
    give bit-identical probabilities, stored activations, every layer's gradient and the parameters after the step."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for wgs in ("0", "100000"):
        env = {**os.environ, "CONFIG": config, "ALPHA": alpha, "IMK_GEMM_AD3_WGS": wgs}
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_probe", "lib_ab.py")], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l.split() and l.split()[0] in ("probs", "act", "grad", "grads", "params", "stats")])
    assert len(outs[0]) > 40 and outs[0] == outs[1]
