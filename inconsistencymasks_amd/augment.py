"""Noisy-Student augmentation on the GPU: the host side of imk_augment (csrc/imk_aug.hip).

Mirrors augment_image_and_mask / augment_image_and_masks (functions.py:2725-2826): the random draws happen here in the
reference's order and with its distributions, the pixel work runs in one fused kernel per batch."""
import ctypes
import random

import numpy as np
import torch

from . import _lib
from ._lib import check, lib


def draw_params(n, brightness_range_alpha=(0.5, 1.5), brightness_range_beta=(-25, 25), max_blur=3, max_noise=25,
                free_rotation=True, rng=None, np_rng=None):
    """n draws of the reference's per-image choices (functions.py:2795-2826, :1494) as an AugParams array.
    rng: random.Random (default: the global stream, like the reference); np_rng: numpy Generator/RandomState."""
    rng = rng or random
    np_rng = np_rng or np.random
    arr = (_lib.AugParams * n)()
    for q in arr:
        q.flip_v = int(free_rotation and rng.randint(0, 1) == 1)
        q.flip_h = int(rng.randint(0, 1) == 1)
        q.rot = rng.randint(0, 3) if free_rotation else 0
        q.alpha = float(np_rng.uniform(brightness_range_alpha[0], brightness_range_alpha[1]))
        q.beta = float(np_rng.uniform(brightness_range_beta[0], brightness_range_beta[1]))
        q.bright_on = int(rng.randint(0, 1) == 1)
        b = rng.randint(0, max_blur)
        q.blur_k = 2 * b + 1 if 1 <= b <= 3 else 0
        q.noise_max = int(max_noise) if max_noise > 0 else 0
        q.seed = rng.getrandbits(32)
    return arr


def augment_batch(images, masks, params):
    """images [B,H,W,C] u8 cuda, masks [B,H,W,Cm] u8 cuda or None, params: AugParams array of length B."""
    assert images.is_cuda and images.dtype == torch.uint8 and images.is_contiguous() and images.dim() == 4
    B, H, W, C = images.shape
    assert len(params) == B
    quarter = int(any(q.rot in (1, 3) for q in params))
    prm = torch.frombuffer(bytearray(bytes(params)), dtype=torch.uint8).to(images.device)
    out = torch.empty_like(images)
    m_out = None
    cm = 0
    if masks is not None:
        assert masks.is_cuda and masks.dtype == torch.uint8 and masks.is_contiguous() and masks.shape[:3] == images.shape[:3]
        cm = masks.shape[3]
        m_out = torch.empty_like(masks)
    rc = lib.imk_augment(images.data_ptr(), masks.data_ptr() if masks is not None else None, B, H, W, C, cm,
                         prm.data_ptr(), out.data_ptr(), m_out.data_ptr() if m_out is not None else None, quarter,
                         torch.cuda.current_stream().cuda_stream)
    check(rc, "imk_augment")
    return out, m_out


def augment_image_and_mask(image, mask, brightness_range_alpha=(0.5, 1.5), brightness_range_beta=(-25, 25), max_blur=3,
                           max_noise=25, free_rotation=True):
    """functions.py:2779-2826 for one numpy image/mask pair ([H,W,C] u8)."""
    prm = draw_params(1, brightness_range_alpha, brightness_range_beta, max_blur, max_noise, free_rotation)
    m3 = mask if mask.ndim == 3 else mask[..., None]
    i3 = image if image.ndim == 3 else image[..., None]
    o, m = augment_batch(torch.from_numpy(np.ascontiguousarray(i3))[None].cuda(),
                         torch.from_numpy(np.ascontiguousarray(m3))[None].cuda(), prm)
    o, m = o[0].cpu().numpy(), m[0].cpu().numpy()
    return (o if image.ndim == 3 else o[..., 0]), (m if mask.ndim == 3 else m[..., 0])


def augment_image_and_masks(image, masks, brightness_range_alpha=(0.5, 1.5), brightness_range_beta=(-25, 25), max_blur=3,
                            max_noise=25, free_rotation=True):
    """functions.py:2725-2776: one image, a list of masks (HeLa: alive / dead / position)."""
    stack = np.concatenate([m if m.ndim == 3 else m[..., None] for m in masks], axis=2)
    o, m = augment_image_and_mask(image, stack, brightness_range_alpha, brightness_range_beta, max_blur, max_noise,
                                  free_rotation)
    outs, c0 = [], 0
    for src in masks:
        c = src.shape[2] if src.ndim == 3 else 1
        part = m[..., c0:c0 + c]
        outs.append(part if src.ndim == 3 else part[..., 0])
        c0 += c
    return o, outs
