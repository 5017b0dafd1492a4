"""GPU parity of imk_augment against oracle/aug_oracle.py (bit-exact: integer / byte work)."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(n, seed, **kw):
    from inconsistencymasks_amd import augment
    return augment.draw_params(n, rng=random.Random(seed), np_rng=np.random.RandomState(seed), **kw)


@pytest.mark.parametrize("shape,cm,free", [((6, 64, 64, 3), 1, True), ((5, 48, 80, 3), 1, False), ((4, 32, 32, 1), 3, True),
                                           ((3, 7, 7, 3), 1, True)])
def test_augment_matches_oracle(shape, cm, free):
    from inconsistencymasks_amd import augment
    from oracle import aug_oracle as A
    rng = np.random.default_rng(0)
    B, H, W, C = shape
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    msk = rng.integers(0, 256, (B, H, W, cm), dtype=np.uint8)
    for rep in range(4):                       # several draws so that every branch is visited
        prm = _params(B, 100 + rep, free_rotation=free, max_blur=3, max_noise=25)
        o, m = augment.augment_batch(torch.from_numpy(img).cuda(), torch.from_numpy(msk).cuda(), prm)
        o, m = o.cpu().numpy(), m.cpu().numpy()
        for b, q in enumerate(prm):
            wi, wm = A.augment(img[b], msk[b], q.flip_v, q.flip_h, q.rot, q.bright_on, q.alpha, q.beta, q.blur_k,
                               q.noise_max, q.seed)
            assert np.array_equal(o[b], wi), (rep, b, [getattr(q, f[0]) for f in q._fields_])
            assert np.array_equal(m[b], wm)


def test_augment_each_stage_alone():
    from inconsistencymasks_amd import _lib, augment
    from oracle import aug_oracle as A
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (1, 40, 40, 3), dtype=np.uint8)
    base = dict(flip_v=0, flip_h=0, rot=0, bright_on=0, alpha=1.0, beta=0.0, blur_k=0, noise_max=0, seed=7)
    cases = [dict(flip_v=1), dict(flip_h=1), dict(rot=1), dict(rot=2), dict(rot=3), dict(bright_on=1, alpha=1.5, beta=-25.0),
             dict(bright_on=1, alpha=0.5, beta=25.0), dict(blur_k=3), dict(blur_k=5), dict(blur_k=7), dict(noise_max=5),
             dict(noise_max=25), dict()]
    for c in cases:
        kw = dict(base, **c)
        prm = (_lib.AugParams * 1)()
        for k, v in kw.items():
            setattr(prm[0], k, v)
        o, _ = augment.augment_batch(torch.from_numpy(img).cuda(), None, prm)
        want, _ = A.augment(img[0], None, **kw)
        assert np.array_equal(o[0].cpu().numpy(), want), c
    # identity parameters return the input
    prm = (_lib.AugParams * 1)()
    prm[0].alpha = 1.0
    o, _ = augment.augment_batch(torch.from_numpy(img).cuda(), None, prm)
    assert np.array_equal(o.cpu().numpy(), img)


def test_augment_rejects_quarter_turn_of_non_square():
    from inconsistencymasks_amd import _lib, augment
    prm = (_lib.AugParams * 1)()
    prm[0].rot = 1
    with pytest.raises(_lib.ImkError):
        augment.augment_batch(torch.zeros((1, 8, 16, 3), dtype=torch.uint8, device="cuda"), None, prm)


@pytest.mark.gpu
@pytest.mark.parametrize("planes,hw_shape,c", [(1, (16, 16), 3), (3, (24, 40), 1), (1, (5, 7), 3)])
def test_gather_pairs_is_an_indexed_copy_with_the_parsers_mask_arithmetic(planes, hw_shape, c):
    """imk_gather_pairs = x[idx] and (mask[idx] // 255 * mul) moved to NHWC: the shuffle + parse of the reference's tf.data
    pipeline over a device-resident set (functions.py:207-209, 975, 1001-1011), bit for bit against torch indexing."""
    import torch
    from inconsistencymasks_amd import functions as F
    g = torch.Generator(device="cuda").manual_seed(5)
    H, W = hw_shape
    x = torch.randint(0, 256, (37, H, W, c), dtype=torch.uint8, device="cuda", generator=g)
    m = (torch.randint(0, 2, (37, planes, H, W), dtype=torch.uint8, device="cuda", generator=g) * 255)
    idx = torch.randperm(37, device="cuda", generator=g)[:29]
    mul = torch.tensor([1, 1, 3][:planes], dtype=torch.uint8, device="cuda") if planes == 3 else None
    gx, gm = F.gather_pairs(idx, img=x, mask_planar=m, div255=True, mul=mul)
    want_m = m[idx].permute(0, 2, 3, 1) // 255
    if mul is not None:
        want_m = want_m * mul
    assert torch.equal(gx, x[idx]) and torch.equal(gm, want_m.contiguous())
    # class-id maps: no arithmetic, one plane
    ids = torch.randint(0, 35, (37, 1, H, W), dtype=torch.uint8, device="cuda", generator=g)
    _, gi = F.gather_pairs(idx, mask_planar=ids)
    assert torch.equal(gi[..., 0], ids[idx][:, 0])


@pytest.mark.gpu
def test_gather_pairs_more_rows_than_one_launch_takes():
    """imk_gather_pairs carries the row index on gridDim.y (<= 65535 rows per call); the Python wrapper cuts longer index lists."""
    import torch
    from inconsistencymasks_amd import functions as F
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.randint(0, 256, (300, 4, 4, 1), dtype=torch.uint8, device="cuda", generator=g)
    idx = torch.randint(0, 300, (70001,), device="cuda", generator=g)
    gx, _ = F.gather_pairs(idx, img=x)
    assert torch.equal(gx, x[idx])
