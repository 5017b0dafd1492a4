"""CPU oracle for the Inconsistency-Mask arithmetic.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain numpy, what the reference computes between "N models'
probability maps" and "pseudo-label / inconsistency mask / blocked image".  It is the
checker for the HIP kernels in inconsistencymasks_amd/csrc/imk_im.hip and is imported only
by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the
product path (inconsistencymasks_amd/*), which fails loudly when libimk.so is missing.

Parity status: PINNED.  tests/test_oracle_golden.py checks every function below against
tests/golden/*.npz, which tests/golden/make_golden.py produced by importing the real
reference (functions.py) in the build container.

Reference lines followed (relative to /root/reference):
  binary core      functions.py:3104-3120  pred_masks_to_im_binary
  binary caller    functions.py:3155-3160  get_im_prediction_binary   (strict  `> thr`)
  HeLa caller      functions.py:3183-3200  get_im_prediction_hela     (`>= thr`, max of IMs, sum of sizes)
  multiclass core  functions.py:3123-3137  pred_masks_to_im_multiclass
  multiclass call  functions.py:3222-3236  get_im_prediction_multiclass (argmax, first max wins; unique-set filter)
  blocking         functions.py:2867-2874  image[im>0]=0 ; mask[im>0]=0
  ISIC keep rule   functions.py:2878-2886  write pair iff pred_size > im_size and pred_size > 0
  mean IM size     functions.py:2889       round(sum / count, 0)
  morphology       functions.py:2858-2864, 3075-3100 (cv2.erode/dilate with a k x k ones kernel) -- UNPINNED:
                   OpenCV is not importable here; restated from its documented behaviour (anchor at k//2,
                   border pixels outside the image never win: +inf for erode, -inf for dilate).
"""
import numpy as np


def threshold_votes(preds, thr=0.5, cmp_ge=False):
    """preds float32 [N, ...] -> int64 votes [N, ...].  NaN compares false under both operators."""
    preds = np.asarray(preds, dtype=np.float32)
    with np.errstate(invalid="ignore"):
        hit = (preds >= np.float32(thr)) if cmp_ge else (preds > np.float32(thr))
    return hit.astype(np.int64)


def im_binary_from_votes(votes):
    """votes int [N, H, W] -> (final u8 {0,255}, im u8 {0,255}, im_size, pred_size).

    A pixel is pseudo-label foreground when every model voted 1, inconsistent when the vote
    count is strictly between 0 and N (functions.py:3107-3115).
    """
    votes = np.asarray(votes)
    n = votes.shape[0]
    s = votes.sum(axis=0)
    agree_fg = s == n
    mixed = (s > 0) & (s < n)
    final = np.where(agree_fg, 255, 0).astype(np.uint8)
    im = np.where(mixed, 255, 0).astype(np.uint8)
    return final, im, np.int64(mixed.sum()), np.int64(agree_fg.sum())


def im_binary(preds, thr=0.5, cmp_ge=False):
    """preds float32 [N, H, W, Kb] -> dict with per-channel masks and the combined IM.

    Kb = 1 is the ISIC case (functions.py:3155-3160); Kb = 3 with cmp_ge=True is HeLa
    (functions.py:3183-3200): three independent binary IMs, combined IM = elementwise max,
    im_size = sum of the three channel sizes (overlaps counted once per channel).
    """
    preds = np.asarray(preds, dtype=np.float32)
    n, h, w, kb = preds.shape
    votes = threshold_votes(preds, thr, cmp_ge)
    finals = np.zeros((kb, h, w), np.uint8)
    ims = np.zeros((kb, h, w), np.uint8)
    im_sizes = np.zeros(kb, np.int64)
    pred_sizes = np.zeros(kb, np.int64)
    for c in range(kb):
        finals[c], ims[c], im_sizes[c], pred_sizes[c] = im_binary_from_votes(votes[..., c])
    return {
        "final": finals,              # [Kb,H,W] u8 {0,255}
        "im_ch": ims,                 # [Kb,H,W] u8 {0,255}
        "im": ims.max(axis=0),        # [H,W]
        "im_size_ch": im_sizes,
        "pred_size_ch": pred_sizes,
        "im_size": np.int64(im_sizes.sum()),
        "pred_size": np.int64(pred_sizes.sum()),
    }


def argmax_first(probs):
    """argmax over the last axis, lowest index on ties (numpy semantics the reference relies on,
    functions.py:3225).  NaN is treated as the maximum by numpy; callers must keep inputs finite."""
    return np.argmax(np.asarray(probs), axis=-1)


def im_multiclass_from_labels(labels):
    """labels int [N, H, W] -> (final u8 class ids, im u8 {0,255}, im_size).

    Pixels where all models equal model 0 keep that class id (NOT x255); the rest become class 0
    and are flagged in the IM (functions.py:3128-3135).
    """
    labels = np.asarray(labels)
    agree = np.all(labels == labels[0:1], axis=0)
    final = np.where(agree, labels[0], 0).astype(np.uint8)
    im = np.where(agree, 0, 255).astype(np.uint8)
    return final, im, np.int64((~agree).sum())


def im_multiclass(probs, filter_unequal_class_pred=False):
    """probs float32 [N, H, W, K] -> dict(final, im, im_size, lists_equal, presence[N,K])."""
    probs = np.asarray(probs, dtype=np.float32)
    n, h, w, k = probs.shape
    labels = argmax_first(probs)
    final, im, im_size = im_multiclass_from_labels(labels)
    presence = np.zeros((n, k), np.uint8)
    for m in range(n):
        presence[m, np.unique(labels[m])] = 1
    if filter_unequal_class_pred:  # functions.py:3231-3234: compare the SETS of predicted classes
        lists_equal = bool(np.all(presence == presence[0:1]))
    else:
        lists_equal = True
    return {"final": final, "im": im, "im_size": im_size, "lists_equal": lists_equal, "presence": presence}


def block(image, masks, im, block_input=True, block_output=True):
    """Zero the IM pixels in the image (all channels) and in each mask (functions.py:2867-2874).
    `image` [H,W,C] or [H,W]; `masks` list of [H,W] / [H,W,3] arrays.  Returns copies."""
    hit = np.asarray(im) > 0
    image = np.array(image, copy=True)
    masks = [np.array(m, copy=True) for m in masks]
    if block_input:
        image[hit] = 0
    if block_output:
        for m in masks:
            m[hit] = 0
    return image, masks


def keep_isic(pred_size, im_size, filter_bad_predictions=True):
    """functions.py:2878-2886: sizes are the ones taken BEFORE any morphology."""
    if not filter_bad_predictions:
        return True
    return bool(pred_size > im_size and pred_size > 0)


def mean_im_size(im_sizes):
    """functions.py:2889 -- python round() to 0 decimals (banker's rounding on exact halves)."""
    im_sizes = list(im_sizes)
    return round(sum(int(s) for s in im_sizes) / len(im_sizes), 0)


# --------------------------------------------------------------------------------------------
# Morphology (UNPINNED: no OpenCV here).  k x k all-ones structuring element, anchor k//2,
# out-of-image neighbours ignored (OpenCV's default BORDER_CONSTANT with the morphology default
# border value behaves as "identity element").
# --------------------------------------------------------------------------------------------
def _morph(mask, k, op):
    mask = np.asarray(mask)
    if k <= 0:
        return mask.copy()
    h, w = mask.shape
    a = k // 2                      # anchor
    lo, hi = a, k - 1 - a           # taps cover [-a, k-1-a]
    ident = 255 if op == "erode" else 0
    pad = np.full((h + k - 1, w + k - 1), ident, mask.dtype)
    pad[lo:lo + h, lo:lo + w] = mask
    out = np.full((h, w), ident, mask.dtype)
    red = np.minimum if op == "erode" else np.maximum
    for dy in range(k):
        for dx in range(k):
            out = red(out, pad[dy:dy + h, dx:dx + w])
    del hi
    return out


def erode(mask, k):
    return _morph(mask, k, "erode")


def dilate(mask, k):
    return _morph(mask, k, "dilate")


def dilate_mask_per_class(mask, k=3):
    """functions.py:3075-3100: dilate each non-zero class separately in ascending class order;
    later (higher) classes overwrite earlier ones where dilations overlap."""
    mask = np.asarray(mask)
    out = np.zeros_like(mask)
    for u in np.unique(mask):
        if u == 0:
            continue
        d = dilate((mask == u).astype(np.uint8), k)
        out[d == 1] = u
    return out
