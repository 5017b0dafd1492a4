"""CPU oracle for EvalNet (forward, losses, backward).  TEST INFRASTRUCTURE ONLY.

torch-CPU restatement of the network the reference builds in evalnet.py, used as the checker for the HIP path
(imk_evalnet_*).  Imported only by tests/ and __graft_entry__.smoke(); the product path never touches it.

Parity status: **PARITY UNPINNED** -- like the U-Net, the reference runs this network inside TensorFlow/Keras, which
cannot be imported here, and ships no test or golden vector for it.  What this file follows:
  topology / layer order   evalnet.py:24-47 (get_evalnet) and :49-73 (get_evalnet_miou):
                            input_block  = [Lambda x/255 if normalize] -> Conv1x1+ReLU -> BN           (evalnet.py:4-11)
                            conv_block   = Conv3x3+ReLU -> Conv1x1+ReLU -> BN -> MaxPooling2D(2,2)     (evalnet.py:14-21)
                            a = conv_block(input_block(A));  b = conv_block(input_block(B));  c = concatenate([a, b])
                            c = conv_block x5 with 16a, 32a, 64a, 128a, 256a filters; GlobalAvgPool2D;
                            Dense(1, sigmoid)                                     (get_evalnet)
                            Dense(Cb, sigmoid, 'iou') and Dense(Cb, sigmoid, 'detection')   (get_evalnet_miou)
  Keras defaults            as in unet_oracle.py; Dense: glorot_uniform kernel, zero bias
  losses                    functions.py:4492 'mean_squared_error' (get_evalnet);
                            functions.py:4708 loss=['mse', 'binary_crossentropy'] (get_evalnet_miou), summed with weight 1;
                            Keras takes the sigmoid ACTIVATION's cached logits for binary_crossentropy (no clipping)
  numerics                  mixed_float16 policy in every EvalNet script (HeLa/14_HeLa_aug_IM++.py:19): fp16 activations,
                            fp32 variables; the HIP path computes GAP + Dense + losses in fp32 from the fp16 map
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import unet_oracle as U


def layer_table(ca, cb, n_out, alpha, two_heads):
    """Ordered (name, kind, k, cin, cout), Keras creation order of evalnet.py:24-45."""
    f = lambda v: int(v * alpha)
    F0 = f(16)
    t = []
    for tw, cin in (("a", ca), ("b", cb)):
        t += [(f"{tw}.in.c", "conv", 1, cin, F0), (f"{tw}.in.bn", "bn", 0, F0, F0),
              (f"{tw}.c3", "conv", 3, F0, F0), (f"{tw}.c1", "conv", 1, F0, F0), (f"{tw}.bn", "bn", 0, F0, F0)]
    prev = 2 * F0
    for i, v in enumerate((16, 32, 64, 128, 256), start=1):
        t += [(f"m{i}.c3", "conv", 3, prev, f(v)), (f"m{i}.c1", "conv", 1, f(v), f(v)), (f"m{i}.bn", "bn", 0, f(v), f(v))]
        prev = f(v)
    if two_heads:
        t += [("iou", "conv", 1, prev, n_out), ("detection", "conv", 1, prev, n_out)]
    else:
        t += [("dense", "conv", 1, prev, n_out)]
    return t


def count_params(ca, cb, n_out, alpha, two_heads):
    total = trainable = 0
    for name, kind, k, ci, co in layer_table(ca, cb, n_out, alpha, two_heads):
        if kind == "conv":
            total += k * k * ci * co + co
            trainable += k * k * ci * co + co
        else:
            total += 4 * co
            trainable += 2 * co
    return total, trainable


def init_weights(ca, cb, n_out, alpha, two_heads, seed):
    gen = torch.Generator().manual_seed(seed)
    w = {}
    for name, kind, k, ci, co in layer_table(ca, cb, n_out, alpha, two_heads):
        if kind == "conv":
            if name in ("dense", "iou", "detection"):      # Keras Dense default: glorot_uniform
                lim = math.sqrt(6.0 / (ci + co))
                w[name + ".w"] = (torch.rand((1, 1, ci, co), generator=gen) * 2 - 1) * lim
            else:
                w[name + ".w"] = U.he_normal_((k, k, ci, co), k * k * ci, gen)
            w[name + ".b"] = torch.zeros(co)
        else:
            w[name + ".gamma"] = torch.ones(co)
            w[name + ".beta"] = torch.zeros(co)
            w[name + ".mean"] = torch.zeros(co)
            w[name + ".var"] = torch.ones(co)
    return w


def one_hot(class_ids, num_classes):
    """functions.py:4978 / 5990: np.stack([(mask == cls) for cls in range(num_classes)], axis=-1), as uint8"""
    ids = np.asarray(class_ids)
    ids = ids[..., 0] if ids.ndim == 4 else ids
    return np.stack([(ids == c) for c in range(num_classes)], axis=-1).astype(np.uint8)


def forward(p, xa_u8, xb_u8, two_heads, normalize_a=True, normalize_b=True, training=False, emulate_fp16=False,
            stats_out=None, taps=None, override=None):
    """xa [B,H,W,Ca], xb [B,H,W,Cb] uint8 (for the multiclass nets: one_hot(class ids), normalize_b False)
    -> (outputs [B, n_heads*K] float32, logits)."""
    f16 = emulate_fp16

    def tap(name, t):
        if taps is not None:
            taps[name] = t.detach().permute(0, 2, 3, 1).contiguous()

    def c(name, t):
        if override is not None and name in override:
            ov = torch.as_tensor(override[name]).float().permute(0, 3, 1, 2)
            pre = U._conv(t, p[name + ".w"], p[name + ".b"], False, f16)
            mask = (ov > 0).float()
            y = pre * mask + (ov - pre * mask).detach()
        else:
            y = U._conv(t, p[name + ".w"], p[name + ".b"], True, f16)
        tap(name, y)
        return y

    def bn(name, t):
        return U._bn(t, p, name, training, f16, stats_out)

    def tower(tw, x_u8, normalize):
        x = torch.as_tensor(np.asarray(x_u8)).float().permute(0, 3, 1, 2)
        if normalize:
            x = x / 255.0
        x = U._r(x, f16)
        y = bn(f"{tw}.in.bn", c(f"{tw}.in.c", x))
        y = bn(f"{tw}.bn", c(f"{tw}.c1", c(f"{tw}.c3", y)))
        return F.max_pool2d(y, 2)

    y = torch.cat([tower("a", xa_u8, normalize_a), tower("b", xb_u8, normalize_b)], dim=1)
    for i in range(1, 6):
        y = F.max_pool2d(bn(f"m{i}.bn", c(f"m{i}.c1", c(f"m{i}.c3", y))), 2)
    feat = y.mean(dim=(2, 3))                       # GlobalAvgPool2D, fp32
    heads = ("iou", "detection") if two_heads else ("dense",)
    logits = torch.cat([feat @ p[h + ".w"][0, 0] + p[h + ".b"] for h in heads], dim=1)
    return torch.sigmoid(logits), logits


def loss_fn(out, logits, y, two_heads):
    """Returns (total, head-0 mse, head-1 bce).  y [B, n_heads*K]."""
    y = torch.as_tensor(np.asarray(y)).float()
    if not two_heads:
        l0 = ((out - y) ** 2).mean()
        return l0, l0, torch.zeros(())
    k = y.shape[1] // 2
    l0 = ((out[:, :k] - y[:, :k]) ** 2).mean()
    l1 = F.binary_cross_entropy_with_logits(logits[:, k:], y[:, k:])
    return l0 + l1, l0, l1


def grads(p, xa, xb, y, two_heads, normalize_a=True, normalize_b=True, emulate_fp16=False, loss_scale=1.0, override=None):
    """One training-mode forward + backward.  Returns (losses tuple of floats, outputs, grads dict, batch stats)."""
    names = U.trainable_names(p)
    leaves = {k: p[k].clone().requires_grad_(True) for k in names}
    q = dict(p)
    q.update(leaves)
    stats = {}
    out, logits = forward(q, xa, xb, two_heads, normalize_a, normalize_b, training=True, emulate_fp16=emulate_fp16,
                          stats_out=stats, override=override)
    total, l0, l1 = loss_fn(out, logits, y, two_heads)
    (total * loss_scale).backward()
    g = {k: leaves[k].grad / loss_scale for k in names}
    return (float(total.detach()), float(l0.detach()), float(l1.detach())), out.detach(), g, stats
