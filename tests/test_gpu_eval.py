"""GPU parity of imk_eval_binary / imk_eval_multiclass: integer counts vs numpy, metric values vs the numpy metric
functions (which tests/test_cpu_api.py pins to the reference's golden values)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bc(gt, pr):
    gn, p, gh = gt != 0, pr != 0, gt >= 128
    return [int((gn & p).sum()), int((gn | p).sum()), int(gh.sum()), int(p.sum()), int((gh & p).sum())]


@pytest.mark.parametrize("shape", [(3, 256, 256), (5, 17, 23), (2, 1, 1), (4, 208, 416)])
@pytest.mark.parametrize("ge", [False, True])
def test_eval_binary(shape, ge):
    from inconsistencymasks_amd import evaluate as E, functions as F
    rng = np.random.default_rng(1)
    probs = rng.random(shape, dtype=np.float32)
    probs.ravel()[::7] = 0.5                                   # the threshold itself
    gt = rng.choice(np.array([0, 1, 127, 128, 255], np.uint8), shape)
    gt[0] = 0                                                   # empty ground truth
    pred, counts = E.eval_binary(torch.from_numpy(probs).cuda(), torch.from_numpy(gt).cuda(), 0.5, ge)
    want = ((probs >= 0.5) if ge else (probs > 0.5)).astype(np.uint8) * 255
    assert np.array_equal(pred.cpu().numpy(), want)
    for b in range(shape[0]):
        assert counts[b].tolist() == _bc(gt[b], want[b])
        iou, dice = E.iou_dice_from_counts(counts[b])
        assert iou == F.get_IoU_binary(gt[b], want[b]) and float(dice) == float(F.dice_score_numpy_binary(gt[b], want[b]))


@pytest.mark.parametrize("shape,K", [((3, 64, 64), 9), ((2, 208, 416), 35), ((4, 5, 7), 3), ((1, 32, 32), 1)])
def test_eval_multiclass(shape, K):
    from inconsistencymasks_amd import evaluate as E, functions as F
    rng = np.random.default_rng(2)
    probs = rng.random(shape + (K,), dtype=np.float32)
    probs[..., 0][probs[..., 0] > 0.9] = 2.0
    if K > 2:
        probs[0, :, :, 2] = probs[0, :, :, 1]                  # ties: the first maximum wins
    gt = rng.integers(0, K + 1, shape).astype(np.uint8)        # includes an id the net never predicts
    gt[-1] = 255 if K < 255 else 0
    pred, counts = E.eval_multiclass(torch.from_numpy(probs).cuda(), torch.from_numpy(gt).cuda())
    want = probs.argmax(-1).astype(np.uint8)
    assert np.array_equal(pred.cpu().numpy(), want)
    for b in range(shape[0]):
        assert np.array_equal(counts[b, 0], np.bincount(gt[b].ravel(), minlength=256))
        assert np.array_equal(counts[b, 1], np.bincount(want[b].ravel(), minlength=256))
        pa, iou = E.pa_iou_from_counts(counts[b], gt[b].size)
        assert pa == F.pixel_accuracy(want[b], gt[b]) and iou == F.get_IoU_multi_unique(want[b], gt[b])


def test_eval_golden_metrics(golden_dir):
    """the reference's own metric values (tests/golden/metrics.npz) through the GPU kernels"""
    import os
    from inconsistencymasks_amd import evaluate as E
    g = np.load(os.path.join(golden_dir, "metrics.npz"))
    for k in g["cases"]:
        gt, pr = g[k + "_gt"], g[k + "_pr"]
        if gt.ndim != 2 or not set(np.unique(pr)) <= {0, 255}:
            continue
        probs = (pr > 0).astype(np.float32)
        _, c = E.eval_binary(torch.from_numpy(probs[None]).cuda(), torch.from_numpy(gt.astype(np.uint8)[None]).cuda())
        iou, dice = E.iou_dice_from_counts(c[0])
        assert iou == float(g[k + "_iou"][0]) and float(dice) == float(g[k + "_dice"][0])
    for k in g["mc_cases"]:
        gt, pr = g[k + "_gt"].astype(np.uint8), g[k + "_pr"].astype(np.uint8)
        K = int(max(gt.max(), pr.max())) + 1
        probs = np.eye(K, dtype=np.float32)[pr]
        _, c = E.eval_multiclass(torch.from_numpy(probs[None]).cuda(), torch.from_numpy(gt[None]).cuda())
        pa, iou = E.pa_iou_from_counts(c[0], gt.size)
        assert pa == float(g[k + "_pa"][0]) and iou == float(g[k + "_iou"][0])


def test_soft_sums_and_mean_iou_monitor():
    """imk_eval_soft_sums against numpy in float64, and the MeanIoU monitor built on it against the reference's formula
    (functions.py:75-90: per class intersection / (sum_true + sum_pred - intersection), mean over classes, mean over batches)."""
    from inconsistencymasks_amd import evaluate as E
    from inconsistencymasks_amd import functions as F
    rng = np.random.default_rng(4)
    for (b, h, w, k) in ((3, 32, 48, 9), (2, 16, 16, 35), (5, 64, 64, 3), (1, 16, 16, 64)):
        p = rng.random((b, h, w, k)).astype(np.float32)
        p /= p.sum(-1, keepdims=True)
        gt = rng.integers(0, k, (b, h, w)).astype(np.uint8)
        gt[gt == k - 1] = 0                                   # one class absent from the ground truth
        s = E.soft_sums(torch.from_numpy(p).cuda(), torch.from_numpy(gt).cuda(), 0)
        oh = np.eye(k)[gt]
        want = np.stack([(oh * p).sum((0, 1, 2)), oh.sum((0, 1, 2)), p.astype(np.float64).sum((0, 1, 2))])
        assert np.allclose(s, want, rtol=2e-6, atol=1e-6)
        assert np.array_equal(s[1], oh.sum((0, 1, 2)))        # counts are exact
        again = E.soft_sums(torch.from_numpy(p).cuda(), torch.from_numpy(gt).cuda(), 0)
        assert np.array_equal(s, again)                       # deterministic
        m = F.MeanIoU(k)
        m.update_state(torch.from_numpy(gt).cuda(), torch.from_numpy(p).cuda())
        m.update_state(torch.from_numpy(gt[:1]).cuda(), torch.from_numpy(p[:1]).cuda())
        def ref(pp, gg):
            o = np.eye(k)[gg]
            inter = (o * pp).sum((0, 1, 2))
            return float(np.mean(inter / (o.sum((0, 1, 2)) + pp.sum((0, 1, 2)) - inter)))
        assert m.result() == pytest.approx((ref(p, gt) + ref(p[:1], gt[:1])) / 2, rel=1e-5)
    y = rng.integers(0, 4, (4, 32, 32, 3)).astype(np.uint8)
    p = rng.random((4, 32, 32, 3)).astype(np.float32)
    sq = E.soft_sums(torch.from_numpy(p).cuda(), torch.from_numpy(y).cuda(), 1)
    assert sq == pytest.approx(float(((p.astype(np.float64) - y) ** 2).sum()), rel=2e-6)
