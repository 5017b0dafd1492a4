"""Host side of the evaluation kernels (csrc/imk_eval.hip): integer pixel counts on the GPU, the reference's float
expressions on the host (functions.py:1767-1861), so every metric value is bit-identical to the numpy loop."""
import numpy as np
import torch

from ._lib import check, lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def eval_binary(probs, gt, thr=0.5, cmp_ge=False, want_pred=True):
    """probs [B,H,W] or [B,H,W,1] f32 cuda, gt [B,H,W] u8 cuda -> (pred u8 {0,255} cuda | None, counts [B,5] int64 numpy)."""
    if probs.dim() == 4:
        assert probs.shape[3] == 1
        probs = probs[..., 0]
    probs, gt = probs.contiguous(), gt.contiguous()
    assert probs.is_cuda and probs.dtype == torch.float32 and gt.dtype == torch.uint8 and gt.shape == probs.shape
    B, H, W = probs.shape
    pred = torch.empty((B, H, W), dtype=torch.uint8, device=probs.device) if want_pred else None
    counts = torch.empty((B, 5), dtype=torch.int64, device=probs.device)
    check(lib.imk_eval_binary(probs.data_ptr(), float(thr), int(bool(cmp_ge)), gt.data_ptr(), B, H, W,
                              pred.data_ptr() if want_pred else None, counts.data_ptr(), _stream()), "imk_eval_binary")
    return pred, counts.cpu().numpy()


def iou_dice_from_counts(c, smooth=1):
    """get_IoU_binary (functions.py:1767-1788) and dice_score_numpy_binary (:1837-1861) from one image's counts."""
    inter, union, g, p, gp = (int(v) for v in c)
    iou = np.int64(inter) / (np.int64(union) + 1e-7)
    f = np.float32
    dice = (2.0 * f(gp) + smooth) / ((f(g) + f(p)) + smooth)     # float32 sums of 0/1 arrays are exact integers
    return iou, dice


def eval_multiclass(probs, gt, want_pred=True):
    """probs [B,H,W,K] f32 cuda, gt [B,H,W] u8 cuda -> (pred u8 class ids cuda | None, counts [B,4,256] int64 numpy)."""
    probs, gt = probs.contiguous(), gt.contiguous()
    assert probs.is_cuda and probs.dtype == torch.float32 and gt.dtype == torch.uint8 and gt.shape == probs.shape[:3]
    B, H, W, K = probs.shape
    pred = torch.empty((B, H, W), dtype=torch.uint8, device=probs.device) if want_pred else None
    counts = torch.empty((B, 4, 256), dtype=torch.int64, device=probs.device)
    check(lib.imk_eval_multiclass(probs.data_ptr(), gt.data_ptr(), B, H, W, K, pred.data_ptr() if want_pred else None,
                                  counts.data_ptr(), _stream()), "imk_eval_multiclass")
    return pred, counts.cpu().numpy()


def pa_iou_from_counts(c, n_pix):
    """pixel_accuracy (functions.py:1820-1834) and get_IoU_multi_unique (:1791-1816) from one image's counts."""
    n_gt, n_pr, n_both = c[0], c[1], c[2]
    pa = np.int64(c[3, 0]) / np.int64(n_pix)
    classes = np.nonzero(n_gt)[0]            # np.unique(gt), ascending
    total = 0.0
    for i in classes:
        total += np.int64(n_both[i]) / ((np.int64(n_gt[i]) + np.int64(n_pr[i]) - np.int64(n_both[i])) + 1e-7)
    return pa, total / len(classes)


def soft_sums(probs, gt, mode):
    """Validation-monitor reductions on the device (imk_eval_soft_sums), deterministic.
    mode 0: probs [..., K] f32, gt [...] u8 class ids -> float64 numpy [3, K] = (sum [gt == k] p_k, sum [gt == k], sum p_k)
    mode 1: probs [..., K] f32, gt [..., K] u8 targets -> float64 sum of squared errors"""
    probs, gt = probs.contiguous(), gt.contiguous()
    K = probs.shape[-1]
    n_pix = probs.numel() // K
    assert probs.is_cuda and probs.dtype == torch.float32 and gt.dtype == torch.uint8
    assert gt.numel() == (n_pix if mode == 0 else n_pix * K)
    out = torch.empty(lib.imk_eval_soft_out_doubles(K), dtype=torch.float64, device=probs.device)
    check(lib.imk_eval_soft_sums(probs.data_ptr(), gt.data_ptr(), n_pix, K, int(mode), out.data_ptr(), _stream()),
          "imk_eval_soft_sums")
    res = out[:3 * K].cpu().numpy()
    return res.reshape(3, K) if mode == 0 else float(res[0])
