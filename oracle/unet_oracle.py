"""CPU oracle for the tiny U-Net (forward, loss, backward, AdamW).  TEST INFRASTRUCTURE ONLY.

torch-CPU restatement of the network the reference builds with Keras, used as the checker
for the HIP kernels and as the "port" CPU baseline of bench.py.  Imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; the product path never touches it.

Parity status: **PARITY UNPINNED** for everything in this file.  The reference executes this
network inside TensorFlow/Keras (+ tensorflow_addons AdamW), none of which can be installed or
imported here, and the reference ships no test/golden vector for it.  What this
file follows instead:
  topology / layer order   unet.py:4-67   (Lambda x/255 -> Conv1x1+ReLU -> BN ; encoder: Conv3x3+ReLU ->
                            Conv1x1+ReLU -> BN -> MaxPool2 ; bottleneck same without pool ; decoder:
                            UpSampling2D(2) nearest + add skip -> Conv1x1+ReLU -> BN -> Conv3x3+ReLU ->
                            Conv1x1+ReLU -> BN ; head Conv1x1 fp32 with sigmoid/softmax)
  Keras defaults            Conv2D use_bias=True, padding='same'; BatchNormalization eps=1e-3, momentum=0.99,
                            gamma=1 beta=0 mean=0 var=1; he_normal = truncated normal, std=sqrt(2/fan_in)/0.8796
  numerics                  ISIC_2018/09_ISIC_2018_IM.py:16 mixed_float16: fp16 activations, fp32 variables,
                            fp32 head (unet.py:63).  `emulate_fp16=True` rounds weights and every stored
                            activation to fp16 (fp32 accumulation), which is what the HIP path computes.
  training recipe           functions.py:207-218: batch 32, tfa AdamW(lr=3e-3, weight_decay=1e-4), 'mse' or
                            CategoricalCrossentropy; config.ini:2-14
  tfa AdamW                 decoupled decay  var -= wd*var  (not scaled by lr), then Adam with eps=1e-7
                            (tensorflow_addons documented behaviour; module absent here)
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3
BN_MOMENTUM = 0.99


def layer_table(c_in, n_out, alpha):
    """Ordered list of (name, kind, k, cin, cout) following unet.py:46-67.  kind in {conv, bn}."""
    f = lambda v: int(v * alpha)
    c16, c32, c64, c128, c256 = f(16), f(32), f(64), f(128), f(256)
    t = [("in.c", "conv", 1, c_in, c16), ("in.bn", "bn", 0, c16, c16)]
    enc = [(c16, c16, c16), (c16, c32, c32), (c32, c64, c64), (c64, c128, c128)]
    for i, (ci, f1, f2) in enumerate(enc, start=1):
        t += [(f"e{i}.c3", "conv", 3, ci, f1), (f"e{i}.c1", "conv", 1, f1, f2), (f"e{i}.bn", "bn", 0, f2, f2)]
    t += [("b.c3", "conv", 3, c128, c256), ("b.c1", "conv", 1, c256, c128), ("b.bn", "bn", 0, c128, c128)]
    dec = [(6, c128, c128, c64), (7, c64, c64, c32), (8, c32, c32, c16), (9, c16, c16, c16)]
    for j, ci, f1, f2 in dec:
        t += [(f"d{j}.ca", "conv", 1, ci, f1), (f"d{j}.bna", "bn", 0, f1, f1),
              (f"d{j}.c3", "conv", 3, f1, f1), (f"d{j}.c1", "conv", 1, f1, f2), (f"d{j}.bnb", "bn", 0, f2, f2)]
    t += [("out", "conv", 1, c16, n_out)]
    return t


def count_params(c_in, n_out, alpha):
    total = trainable = 0
    for name, kind, k, ci, co in layer_table(c_in, n_out, alpha):
        if kind == "conv":
            n = k * k * ci * co + co
            total += n
            trainable += n
        else:
            total += 4 * co
            trainable += 2 * co
    return total, trainable


def he_normal_(shape, fan_in, gen):
    """Keras he_normal: truncated normal (|z|<2) scaled to std sqrt(2/fan_in) after truncation."""
    std = math.sqrt(2.0 / fan_in) / 0.87962566103423978
    w = torch.empty(shape, dtype=torch.float32)
    torch.nn.init.trunc_normal_(w, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=gen)
    return w


def init_weights(c_in, n_out, alpha, seed):
    """dict name -> tensor.  conv: '<n>.w' HWIO [k,k,ci,co], '<n>.b' [co]; bn: .gamma .beta .mean .var"""
    gen = torch.Generator().manual_seed(seed)
    w = {}
    for name, kind, k, ci, co in layer_table(c_in, n_out, alpha):
        if kind == "conv":
            w[name + ".w"] = he_normal_((k, k, ci, co), k * k * ci, gen)
            w[name + ".b"] = torch.zeros(co)
        else:
            w[name + ".gamma"] = torch.ones(co)
            w[name + ".beta"] = torch.zeros(co)
            w[name + ".mean"] = torch.zeros(co)
            w[name + ".var"] = torch.ones(co)
    return w


class _RoundF16(torch.autograd.Function):
    """Round to fp16 in forward; round the incoming gradient to fp16 in backward (the HIP path stores
    activations and activation-gradients as fp16)."""

    @staticmethod
    def forward(ctx, x):
        return x.half().float()

    @staticmethod
    def backward(ctx, g):
        return g.half().float()


class _RoundF16Fwd(torch.autograd.Function):
    """Round to fp16 in forward only: used for the fp16 copy of the fp32 master weights (their gradient
    is accumulated and kept in fp32 by the HIP path, like Keras keeps variable gradients in fp32)."""

    @staticmethod
    def forward(ctx, x):
        return x.half().float()

    @staticmethod
    def backward(ctx, g):
        return g


def _r(x, on):
    return _RoundF16.apply(x) if on else x


class _AffineF16(torch.autograd.Function):
    """y = fp16(x * scale + shift) with ONE rounding: the exact sum (formed in float64: an fp16 x times an fp32 scale has 35
    significant bits) goes to fp16 directly, numpy's float64 -> float16 conversion being correctly rounded.  This is what the HIP
    kernels' BatchNorm-on-load computes since round 4 (`v_fma_mixlo_f16`, csrc/imk_common.h imk_affine*); rounding the fp32
    product and the fp32 sum first differs from it by an fp16 ulp in ~1e-4 of the values.  Backward: the straight-through
    gradients of the affine map, the incoming gradient rounded to fp16 like every stored activation gradient."""

    @staticmethod
    def forward(ctx, x, scale, shift):
        ctx.save_for_backward(x, scale)
        y = x.double() * scale.double()[None, :, None, None] + shift.double()[None, :, None, None]
        return torch.from_numpy(y.numpy().astype(np.float16).astype(np.float32))

    @staticmethod
    def backward(ctx, g):
        x, scale = ctx.saved_tensors
        g = g.half().float()
        return g * scale[None, :, None, None], (g * x).sum(dim=(0, 2, 3)), g.sum(dim=(0, 2, 3))


def _conv(x, w, b, relu, f16, round_grad=True):
    """round_grad=False: the output is rounded to fp16, the gradient that comes back to it is not.  That is the HIP path's conv in
    front of a BatchNorm: the BatchNorm backward (dz = A dy + B z + C) is applied by the consumers while they load dy and z, so the
    gradient w.r.t. this output only ever exists in fp32 registers (Backward structure); the gradients that ARE stored
    -- dy of every BatchNorm, the ReLU-masked gradient between a Conv1x1 and the Conv3x3 in front of it -- are fp16."""
    k = w.shape[0]
    wt = w.permute(3, 2, 0, 1)
    if f16:
        wt = _RoundF16Fwd.apply(wt)
    y = F.conv2d(x, wt, b, padding=k // 2)
    if relu:
        y = F.relu(y)
    if not f16:
        return y
    return _RoundF16.apply(y) if round_grad else _RoundF16Fwd.apply(y)


def _bn(x, p, name, training, f16, stats_out):
    g, bta = p[name + ".gamma"], p[name + ".beta"]
    if training:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        if stats_out is not None:
            # Keras' fused BatchNormalization (the default for 4-D inputs) normalises with the biased batch variance but
            # feeds the moving average the Bessel-corrected one (`_bessels_correction_test_only`): factor n / (n - 1)
            n = x.numel() // x.shape[1]
            stats_out[name] = (mean.detach().clone(), var.detach().clone() * (n / max(n - 1, 1)))
    else:
        mean, var = p[name + ".mean"], p[name + ".var"]
    scale = g * torch.rsqrt(var + BN_EPS)
    shift = bta - mean * scale
    if f16:
        return _AffineF16.apply(x, scale, shift)
    return x * scale[None, :, None, None] + shift[None, :, None, None]


def forward(p, x_u8_nhwc, c_in, n_out, alpha, act_out, training=False, emulate_fp16=False,
            stats_out=None, taps=None, override=None, return_logits=False, grad_taps=None):
    """x uint8 [B,H,W,C] (numpy or torch) -> probabilities float32 [B,H,W,K] (torch).
    `taps`, if a dict, receives every stored intermediate in NHWC float32 (pre-BN conv outputs).
    `override`, if a dict name -> NHWC float32 tensor, replaces the VALUE of that conv output while
    keeping the autograd graph (straight-through): gradients can then be compared with an
    implementation whose forward values differ by fp16 rounding noise, without the chaotic
    amplification through ReLU masks / max-pool arg-maxes."""
    f16 = emulate_fp16
    x = torch.as_tensor(np.asarray(x_u8_nhwc)).float().permute(0, 3, 1, 2) / 255.0
    x = _r(x, f16)

    def tap(name, t):
        if taps is not None:
            taps[name] = t.detach().permute(0, 2, 3, 1).contiguous()

    def c(name, t, relu=True):
        feeds_bn = not name.endswith(".c3")      # unet.py: every conv but a block's 3x3 is followed by a BatchNormalization
        if override is not None and name in override:
            # value := the overriding tensor; gradient := that of (pre-activation * [override > 0]): the ReLU decision
            # follows the overriding values ONLY (not also this oracle's own pre-activation sign)
            ov = torch.as_tensor(override[name]).float().permute(0, 3, 1, 2)
            pre = _conv(t, p[name + ".w"], p[name + ".b"], False, f16)   # fp16 weights, gradient rounded like a stored tensor
            mask = (ov > 0).float() if relu else 1.0
            y = pre * mask + (ov - pre * mask).detach()
        else:
            y = _conv(t, p[name + ".w"], p[name + ".b"], relu, f16, round_grad=not feeds_bn)
        tap(name, y)
        if grad_taps is not None and y.requires_grad:   # gradient w.r.t. this conv's (post-ReLU) output, NHWC
            y.register_hook(lambda g, n=name: grad_taps.__setitem__(n, g.detach().permute(0, 2, 3, 1).contiguous()))
        return y

    def bn(name, t):
        return _bn(t, p, name, training, f16, stats_out)

    y = bn("in.bn", c("in.c", x))
    skips = []
    for i in range(1, 5):
        y = bn(f"e{i}.bn", c(f"e{i}.c1", c(f"e{i}.c3", y)))
        skips.append(y)
        y = F.max_pool2d(y, 2)
    y = bn("b.bn", c("b.c1", c("b.c3", y)))
    for j, skip in zip(range(6, 10), reversed(skips)):
        u = y.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3) + skip
        u = _r(u, f16)
        a = bn(f"d{j}.bna", c(f"d{j}.ca", u))
        y = bn(f"d{j}.bnb", c(f"d{j}.c1", c(f"d{j}.c3", a)))
    # head: fp32 weights, fp32 math on the (fp16-valued) input  (unet.py:63 dtype='float32')
    w = p["out.w"].permute(3, 2, 0, 1)
    logits = F.conv2d(y, w, p["out.b"])
    probs = torch.sigmoid(logits) if act_out == "sigmoid" else torch.softmax(logits, dim=1)
    probs = probs.permute(0, 2, 3, 1).contiguous()
    if return_logits:
        return probs, logits.permute(0, 2, 3, 1).contiguous()
    return probs


def loss_fn(probs, target, kind, logits=None):
    """'mse' : mean over all elements of (p - t)^2, t in {0,1} [B,H,W,K]   (Keras 'mse')
       'cce' : mean over pixels of -sum_k t_k log softmax(logits)_k, t one-hot [B,H,W,K].
               Keras' CategoricalCrossentropy() on the output of a softmax ACTIVATION goes back to the
               cached logits (backend.categorical_crossentropy, `_keras_logits`), i.e. no probability
               clipping and d loss / d logits = p - t exactly.  (Third-party behaviour, unpinned.)
               Without logits the clipped-probability form [1e-7, 1-1e-7] is used."""
    if kind == "mse":
        return ((probs - target) ** 2).mean()
    if logits is not None:
        return -(target * torch.log_softmax(logits, dim=-1)).sum(-1).mean()
    pc = probs.clamp(1e-7, 1 - 1e-7)
    return -(target * pc.log()).sum(-1).mean()


def trainable_names(p):
    return [k for k in p if not (k.endswith(".mean") or k.endswith(".var"))]


def new_opt_state(p):
    return {"step": 0, "m": {k: torch.zeros_like(p[k]) for k in trainable_names(p)},
            "v": {k: torch.zeros_like(p[k]) for k in trainable_names(p)}}


def train_step(p, opt, x_u8, target, c_in, n_out, alpha, act_out, loss_kind,
               lr=3e-3, wd=1e-4, b1=0.9, b2=0.999, eps=1e-7, emulate_fp16=False, loss_scale=1.0,
               return_grads=False, override=None, grad_taps=None, apply=True):
    """One step of forward(train) -> loss -> backward -> tfa-AdamW, in place on p / opt.
    apply=False: a step the dynamic loss scale skips (non-finite gradients) -- Keras' LossScaleOptimizer leaves the variables and
    the optimizer's slots alone, the BatchNorm moving statistics of the forward pass have moved all the same.
    Returns (loss, grads?)"""
    names = trainable_names(p)
    leaves = {k: p[k].clone().requires_grad_(True) for k in names}
    q = dict(p)
    q.update(leaves)
    stats = {}
    probs, logits = forward(q, x_u8, c_in, n_out, alpha, act_out, training=True, emulate_fp16=emulate_fp16,
                            stats_out=stats, override=override, return_logits=True, grad_taps=grad_taps)
    t = torch.as_tensor(np.asarray(target)).float()
    loss = loss_fn(probs, t, loss_kind, logits if act_out == "softmax" else None)
    (loss * loss_scale).backward()
    grads = {k: leaves[k].grad / loss_scale for k in names}
    # dynamic loss scaling: a step whose (scaled, fp16-rounded) gradients are not finite is skipped; the caller halves the scale
    finite = all(bool(torch.isfinite(g).all()) for g in grads.values())
    apply = apply and finite
    opt["last_finite"], opt["last_applied"] = finite, apply
    if apply:
        opt["step"] += 1
    s = max(opt["step"], 1)
    lr_t = lr * math.sqrt(1 - b2 ** s) / (1 - b1 ** s)
    with torch.no_grad():
        for k in (names if apply else []):
            g = grads[k]
            p[k].mul_(1 - wd)                                   # decoupled decay, not scaled by lr
            opt["m"][k].mul_(b1).add_(g, alpha=1 - b1)
            opt["v"][k].mul_(b2).addcmul_(g, g, value=1 - b2)
            p[k].addcdiv_(opt["m"][k], opt["v"][k].sqrt() + eps, value=-lr_t)
        for name, (mean, var) in stats.items():                # moving = moving*m + batch*(1-m)
            p[name + ".mean"].mul_(BN_MOMENTUM).add_(mean, alpha=1 - BN_MOMENTUM)
            p[name + ".var"].mul_(BN_MOMENTUM).add_(var, alpha=1 - BN_MOMENTUM)
    lv = float(loss.detach())
    return (lv, grads) if return_grads else lv


def predict_batch1(p, images_u8, c_in, n_out, alpha, act_out):
    """Reference-structured inference: one batch-1 forward per image (functions.py:2844-2854, 3155-3158)."""
    outs = []
    with torch.no_grad():
        for i in range(len(images_u8)):
            outs.append(forward(p, images_u8[i:i + 1], c_in, n_out, alpha, act_out).numpy())
    return np.concatenate(outs, 0)
