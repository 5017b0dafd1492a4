#!/usr/bin/env python3
"""Generate golden vectors for the IM hot path by importing the REAL reference.

Run ONLY in the build container (needs /root/reference):

    python tests/golden/make_golden.py

The reference module `functions.py` imports cv2 / tensorflow / tensorflow_addons at
the top (functions.py:7,10,11), none of which exist here.  We inject empty stub
modules so that the module body executes; every function exercised below is pure
numpy in the reference (functions.py:3104-3238, 2832-2891, 2988-3070) and runs
unmodified.  Nothing from the reference is copied: the outputs are *data* (inputs and
expected outputs) written to tests/golden/*.npz.

What is pinned (SURVEY.md §8 a'):
  im_binary.npz      pred_masks_to_im_binary / get_im_prediction_binary   (a4, a5)
  im_hela.npz        get_im_prediction_hela                               (a7)
  im_multiclass.npz  pred_masks_to_im_multiclass / get_im_prediction_multiclass (a6)
  writer_isic.npz    create_pseudo_labels_im_ISIC_2018, EK=DK=0           (a8)
  writer_multi.npz   create_pseudo_labels_im_multiclass, EK=DK=0          (a9)
  metrics.npz        get_IoU_binary / dice_score_numpy_binary             (eval helpers)
  evalnet_labels.npz compute_classwise_IoU / compute_classwise_detection_im / compute_classwise_detection
                     (functions.py:4328-4358, 4400-4459): the label arithmetic of the multiclass EvalNet training data
  augment.npz        add_noise (functions.py:1463-1478): the only numpy-only piece of the augmentation chain.  The
                     noise field the reference drew is recorded next to its output (numpy's global stream re-seeded),
                     so the add / clip / dtype arithmetic is pinned; the draw itself is not reproducible by design.
"""
import hashlib
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("IMK_GOLDEN_OUT", HERE)       # tests/test_golden_regenerate.py writes to a scratch directory and compares
REF = "/root/reference"


def _stub(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


def import_reference():
    cv2 = _stub("cv2")
    tf = _stub("tensorflow")
    _stub("tensorflow_addons")
    k = _stub("tensorflow.keras")
    tf.keras = k
    _stub("tensorflow.keras.preprocessing")
    kpi = _stub("tensorflow.keras.preprocessing.image")
    ku = _stub("tensorflow.keras.utils")
    kpi.load_img = kpi.img_to_array = None
    ku.to_categorical = None

    class _Base:  # real (empty) classes so `class MeanIoU(tf.keras.metrics.Metric)` executes
        pass

    k.metrics = types.SimpleNamespace(Metric=_Base)
    k.losses = types.SimpleNamespace(Loss=_Base)
    # one-line shims for the two cv2 calls the pure-numpy functions make
    cv2.split = lambda a: [a[..., i] for i in range(a.shape[-1])]
    cv2.COLOR_BGR2RGB = 4
    cwd = os.getcwd()
    os.chdir(REF)  # functions.py:24 reads config.ini relative to CWD
    sys.path.insert(0, REF)
    import functions as F  # noqa

    os.chdir(cwd)
    return F, cv2


class FixedModel:
    """Fake Keras model: .predict([x]) returns a fixed [1,H,W,K] float32 array."""

    def __init__(self, arr):
        self.arr = arr

    def predict(self, x):
        return self.arr


class LookupModel:
    """Fake model for the directory writers: prediction looked up by image content."""

    def __init__(self, table):
        self.table = table

    def predict(self, x):
        key = hashlib.sha1(np.ascontiguousarray(x[0]).tobytes()).hexdigest()
        return self.table[key]


def adversarial_probs(rng, shape):
    """Random probabilities salted with the edge values SURVEY §8a' lists."""
    p = rng.random(shape, dtype=np.float32)
    flat = p.reshape(-1)
    specials = np.array(
        [0.5, np.nextafter(np.float32(0.5), np.float32(1)), np.nextafter(np.float32(0.5), np.float32(0)),
         0.0, 1.0, np.nan, -0.0, 0.49999997, 0.50000006, np.inf, -np.inf, 1e-38, 0.25, 0.75],
        dtype=np.float32)
    n = max(1, flat.size // 5)
    idx = rng.choice(flat.size, size=n, replace=False)
    flat[idx] = specials[rng.integers(0, len(specials), size=n)]
    return p


def gen_binary(F):
    out = {}
    cases = []
    rng = np.random.default_rng(1234)
    cid = 0
    for (H, W) in [(4, 4), (16, 24), (7, 5), (64, 64), (256, 256)]:
        for N in (2, 3, 4):
            for kind in ("random", "adversarial", "correlated"):
                if (H, W) == (256, 256) and (N != 2 or kind != "correlated"):
                    continue
                if kind == "random":
                    preds = rng.random((N, 1, H, W, 1), dtype=np.float32)
                elif kind == "adversarial":
                    preds = adversarial_probs(rng, (N, 1, H, W, 1))
                else:  # models mostly agree: shared field + small per-model noise, quantised (compressible)
                    base = rng.random((1, 1, H, W, 1), dtype=np.float32)
                    preds = base + (rng.random((N, 1, H, W, 1), dtype=np.float32) - 0.5) * 0.2
                    preds = (np.round(np.clip(preds, 0, 1) * 64) / 64).astype(np.float32)
                models = [FixedModel(preds[n]) for n in range(N)]
                x = np.zeros((1, H, W, 3), np.uint8)
                final, im, im_size, pred_size = F.get_im_prediction_binary(models, x, 0.5)
                # also the bare numpy core on explicit int stacks (a5)
                masks = [(preds[n][0] > 0.5).astype(int) for n in range(N)]
                f2, i2, s2, p2 = F.pred_masks_to_im_binary(masks)
                assert np.array_equal(final, f2) and np.array_equal(im, i2) and s2 == im_size and p2 == pred_size
                k = f"c{cid}"
                out[k + "_preds"] = preds
                out[k + "_final"] = final
                out[k + "_im"] = im
                out[k + "_sizes"] = np.array([im_size, pred_size], np.int64)
                cases.append(k)
                cid += 1
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "im_binary.npz"), **out)
    print("im_binary:", len(cases), "cases")


def gen_hela(F):
    out = {}
    cases = []
    rng = np.random.default_rng(4321)
    cid = 0
    for (H, W) in [(4, 4), (16, 24), (64, 64)]:
        for N in (2, 3):
            for kind in ("random", "adversarial"):
                preds = (rng.random((N, 1, H, W, 3), dtype=np.float32) if kind == "random"
                         else adversarial_probs(rng, (N, 1, H, W, 3)))
                models = [FixedModel(preds[n]) for n in range(N)]
                x = np.zeros((1, H, W, 1), np.uint8)
                alive, dead, pos, cim, im_size = F.get_im_prediction_hela(models, x)
                k = f"c{cid}"
                out[k + "_preds"] = preds
                out[k + "_alive"] = alive
                out[k + "_dead"] = dead
                out[k + "_pos"] = pos
                out[k + "_im"] = cim
                out[k + "_sizes"] = np.array([im_size], np.int64)
                cases.append(k)
                cid += 1
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "im_hela.npz"), **out)
    print("im_hela:", len(cases), "cases")


def softmax_like(rng, shape, ties=False):
    p = rng.random(shape, dtype=np.float32)
    if ties:  # quantise hard so that exact ties between classes are common
        p = (np.round(p * 4) / 4).astype(np.float32)
    s = p.sum(-1, keepdims=True)
    s[s == 0] = 1
    return (p / s).astype(np.float32) if not ties else p


def gen_multiclass(F):
    out = {}
    cases = []
    rng = np.random.default_rng(99)
    cid = 0
    for (H, W) in [(4, 4), (16, 24), (13, 26), (64, 64)]:
        for K in (3, 9, 35):
            for N in (2, 3):
                for kind in ("random", "ties", "agree", "differ", "mostly"):
                    if (H, W) == (64, 64) and not (K == 9 and N == 3 and kind in ("random", "mostly")):
                        continue
                    if (H, W) == (13, 26) and kind in ("agree", "differ") and K != 35:
                        continue
                    if kind == "random":
                        preds = softmax_like(rng, (N, 1, H, W, K))
                    elif kind == "ties":
                        preds = softmax_like(rng, (N, 1, H, W, K), ties=True)
                    elif kind == "agree":
                        one = softmax_like(rng, (1, 1, H, W, K))
                        preds = np.repeat(one, N, axis=0)
                        preds[:, :, : H // 2] = 0
                        preds[:, :, : H // 2, :, 0] = 1  # agree on class 0 => final 0, im 0
                    elif kind == "differ":
                        preds = np.zeros((N, 1, H, W, K), np.float32)
                        for n in range(N):
                            preds[n, ..., (n + 1) % K] = 1.0
                    else:  # mostly agree: shared field plus noise
                        base = rng.random((1, 1, H, W, K), dtype=np.float32)
                        preds = (base + 0.15 * rng.random((N, 1, H, W, K), dtype=np.float32)).astype(np.float32)
                    k = f"c{cid}"
                    out[k + "_preds"] = preds
                    for filt in (False, True):
                        models = [FixedModel(preds[n]) for n in range(N)]
                        x = np.zeros((1, H, W, 3), np.uint8)
                        final, im, im_size, lists_equal = F.get_im_prediction_multiclass(models, x, filt)
                        if not filt:
                            out[k + "_final"] = final
                            out[k + "_im"] = im
                            out[k + "_sizes"] = np.array([im_size], np.int64)
                        else:
                            assert np.array_equal(out[k + "_final"], final) and np.array_equal(out[k + "_im"], im)
                            out[k + "_lists_equal"] = np.array([int(bool(lists_equal))], np.int64)
                    cases.append(k)
                    cid += 1
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "im_multiclass.npz"), **out)
    print("im_multiclass:", len(cases), "cases")


def lesion_image(rng, H, W, C):
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    cy, cx = rng.uniform(0.3, 0.7) * H, rng.uniform(0.3, 0.7) * W
    ry, rx = rng.uniform(0.15, 0.35) * H, rng.uniform(0.15, 0.35) * W
    ell = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2) < 1
    img = rng.integers(90, 200, size=(H, W, C)).astype(np.int32)
    img[ell] -= 70
    return np.clip(img, 0, 255).astype(np.uint8), ell


def run_writer(F, cv2, which, images, models, h, w, c, flags):
    """Drive a reference writer over an in-memory directory through a recording cv2 shim."""
    written = {}
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "src")
        dst = os.path.join(td, "dst")
        os.makedirs(src)
        for name in images:
            open(os.path.join(src, name), "wb").close()
        cv2.imread = lambda p, *a: images[os.path.basename(p)].copy()
        cv2.cvtColor = lambda a, code: a[..., ::-1]
        cv2.imwrite = lambda p, a: written.__setitem__(
            os.path.relpath(p, dst).replace(os.sep, "/"), np.array(a, copy=True))
        if which == "isic":
            mean = F.create_pseudo_labels_im_ISIC_2018(models, h, w, c, src, dst, *flags)
        else:
            mean = F.create_pseudo_labels_im_multiclass(models, h, w, c, src, dst, *flags)
    return written, float(mean)


def gen_writers(F, cv2):
    rng = np.random.default_rng(7)
    H, W, C = 32, 48, 3
    N = 2
    # ---- ISIC writer --------------------------------------------------------
    images, tables = {}, [dict() for _ in range(N)]
    out = {}
    names = []
    for i in range(6):
        name = f"ISIC_{i:07d}.png"
        bgr, ell = lesion_image(rng, H, W, C)
        images[name] = bgr
        names.append(name)
        rgb = bgr[..., ::-1]
        key = hashlib.sha1(np.ascontiguousarray(rgb).tobytes()).hexdigest()
        for n in range(N):
            if i == 4:    # empty prediction -> pred_size == 0 -> filtered out
                p = rng.random((1, H, W, 1), dtype=np.float32) * 0.4
            elif i == 5:  # models disagree almost everywhere -> im_size > pred_size -> filtered out
                p = (rng.random((1, H, W, 1), dtype=np.float32) * 0.2 + (0.7 if n == 0 else 0.1)).astype(np.float32)
                p[0, :2, :2, 0] = 0.9
            else:
                p = (ell[None, :, :, None] * 0.6 + 0.1 + 0.35 * rng.random((1, H, W, 1))).astype(np.float32)
            tables[n][key] = p
            out[f"pred_{i}_{n}"] = p
        out[f"img_{i}"] = bgr
    out["names"] = np.array(names)
    models = [LookupModel(t) for t in tables]
    combos = []
    for bi in (True, False):
        for bo in (True, False):
            for filt in (True, False):
                written, mean = run_writer(F, cv2, "isic", images, models, H, W, C, (True, 0, 0, bi, bo, filt))
                tag = f"bi{int(bi)}_bo{int(bo)}_f{int(filt)}"
                combos.append(tag)
                out[tag + "_mean"] = np.array([mean])
                out[tag + "_files"] = np.array(sorted(written))
                for p, a in written.items():
                    out[tag + "/" + p] = a
    out["combos"] = np.array(combos)
    np.savez_compressed(os.path.join(OUT, "writer_isic.npz"), **out)
    print("writer_isic:", len(combos), "flag combos")

    # ---- multiclass writer ----------------------------------------------------
    K, N = 9, 3
    images, tables = {}, [dict() for _ in range(N)]
    out = {}
    names = []
    for i in range(5):
        name = f"d_r_{i}_.png"
        bgr, ell = lesion_image(rng, H, W, C)
        images[name] = bgr
        names.append(name)
        rgb = bgr[..., ::-1]
        key = hashlib.sha1(np.ascontiguousarray(rgb).tobytes()).hexdigest()
        base = rng.random((1, H, W, K), dtype=np.float32)
        for n in range(N):
            p = (base + 0.2 * rng.random((1, H, W, K), dtype=np.float32)).astype(np.float32)
            if i == 3 and n == 1:
                p[..., 8] = 0  # model 1 never predicts class 8 on this image -> unique-set filter may trip
            tables[n][key] = p
            out[f"pred_{i}_{n}"] = p
        out[f"img_{i}"] = bgr
    out["names"] = np.array(names)
    models = [LookupModel(t) for t in tables]
    combos = []
    for bi in (True, False):
        for bo in (True, False):
            for filt in (False, True):
                written, mean = run_writer(F, cv2, "multi", images, models, H, W, C, (True, 0, 0, bi, bo, filt))
                tag = f"bi{int(bi)}_bo{int(bo)}_f{int(filt)}"
                combos.append(tag)
                out[tag + "_mean"] = np.array([mean])
                out[tag + "_files"] = np.array(sorted(written))
                for p, a in written.items():
                    out[tag + "/" + p] = a
    out["combos"] = np.array(combos)
    np.savez_compressed(os.path.join(OUT, "writer_multi.npz"), **out)
    print("writer_multi:", len(combos), "flag combos")


def gen_metrics(F):
    rng = np.random.default_rng(5)
    out = {}
    cases = []
    for i, (H, W) in enumerate([(8, 8), (32, 48), (64, 64)]):
        for j in range(3):
            gt = (rng.random((H, W)) > 0.5).astype(np.uint8) * 255
            pr = (rng.random((H, W)) > (0.3 + 0.2 * j)).astype(np.uint8) * 255
            if j == 2 and i == 0:
                gt[:] = 0
                pr[:] = 0
            k = f"c{i}_{j}"
            out[k + "_gt"] = gt
            out[k + "_pr"] = pr
            out[k + "_iou"] = np.array([F.get_IoU_binary(gt, pr)], np.float64)
            out[k + "_dice"] = np.array([F.dice_score_numpy_binary(gt, pr)], np.float64)
            cases.append(k)
    # multi-class helpers (functions.py:1791-1834) and the HeLa distance helper (functions.py:6221-6252)
    mc = []
    for i, K in enumerate((3, 9, 35)):
        gt = rng.integers(0, K, (24, 40)).astype(np.uint8)
        pr = np.where(rng.random((24, 40)) > 0.3, gt, rng.integers(0, K, (24, 40))).astype(np.uint8)
        k = f"m{i}"
        out[k + "_gt"], out[k + "_pr"] = gt, pr
        out[k + "_iou"] = np.array([F.get_IoU_multi_unique(pr, gt)], np.float64)
        out[k + "_pa"] = np.array([F.pixel_accuracy(pr, gt)], np.float64)
        mc.append(k)
    out["mc_cases"] = np.array(mc)
    pts = rng.integers(0, 256, (12, 2))
    out["dist_pts"] = pts
    out["dist_min"] = np.array([F.get_min_dist(tuple(p), [tuple(q) for q in pts]) for p in pts], np.float64)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), **out)
    print("metrics:", len(cases), "cases")


def gen_augment(F):
    rng = np.random.default_rng(9)
    out = {}
    cases = []
    for i, (shape, m) in enumerate([((16, 24, 3), 25), ((32, 32, 3), 5), ((8, 8, 1), 15), ((12, 20, 3), 1)]):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        img[0, :4] = 0
        img[1, :4] = 255                       # both clip edges
        np.random.seed(100 + i)
        got = F.add_noise(img.copy(), m)
        np.random.seed(100 + i)
        noise = np.random.randint(m * -1, m, size=img.shape)     # the same draw the call above made
        k = f"n{i}"
        out[k + "_img"], out[k + "_noise"], out[k + "_out"], out[k + "_m"] = img, noise.astype(np.int16), got, np.array([m])
        cases.append(k)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "augment.npz"), **out)
    print("augment:", len(cases), "cases")


def gen_evalnet_labels(F):
    """the numpy-only label helpers of create_training_data_evalnet_miou_im_multiclass (functions.py:3826-3838)"""
    rng = np.random.default_rng(11)
    out = {}
    cases = []
    for i, (K, shape) in enumerate([(3, (16, 16)), (9, (32, 48)), (9, (64, 64)), (35, (40, 80)), (9, (24, 24))]):
        gt = rng.integers(0, K, shape).astype(np.uint8)
        gt[: shape[0] // 2, : shape[1] // 2] = rng.integers(0, K)                   # a dominant class
        if i == 4:
            gt[:] = 4                                                              # single-class ground truth
        pred = np.where(rng.random(shape) > 0.35, gt, rng.integers(0, K, shape)).astype(np.uint8)
        im = (rng.random(shape) > 0.8)
        pred[im] = 0                                                               # blocked by the IM
        if i == 1:
            pred[pred == 0] = 1                                                    # no class-0 pixel in the prediction
        counts = np.zeros(K)
        bins = np.bincount(gt.ravel(), minlength=K)
        counts[:len(bins)] += bins
        gt_blocked = gt.copy()
        gt_blocked[im] = 0
        k = f"e{i}"
        out[k + "_gt"], out[k + "_pred"], out[k + "_im"], out[k + "_K"] = gt, pred, im.astype(np.uint8), np.array([K])
        out[k + "_iou"] = np.array(F.compute_classwise_IoU(pred, gt, K), np.float64)
        out[k + "_det_im"] = np.array(F.compute_classwise_detection_im(gt_blocked, K, counts, 0.3), np.int64)
        out[k + "_det"] = np.array(F.compute_classwise_detection(pred, K), np.int64)
        cases.append(k)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "evalnet_labels.npz"), **out)
    print("evalnet_labels:", len(cases), "cases")


if __name__ == "__main__":
    F, cv2 = import_reference()
    only = set(sys.argv[1:])          # e.g. `make_golden.py evalnet_labels` regenerates one file
    gens = [("binary", lambda: gen_binary(F)), ("hela", lambda: gen_hela(F)), ("multiclass", lambda: gen_multiclass(F)),
            ("writers", lambda: gen_writers(F, cv2)), ("metrics", lambda: gen_metrics(F)), ("augment", lambda: gen_augment(F)),
            ("evalnet_labels", lambda: gen_evalnet_labels(F))]
    with np.errstate(all="ignore"):
        for name, fn in gens:
            if not only or name in only:
                fn()
