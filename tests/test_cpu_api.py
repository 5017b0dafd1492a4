"""CPU-side checks: the C-ABI library builds, loads and exports every symbol include/imk.h declares; the
plan/parameter layout agrees with the oracle's reading of unet.py; host logic (sharding, reductions)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="session")
def built_lib():
    from inconsistencymasks_amd.build import build_lib
    return build_lib()


def test_header_symbols_exported(built_lib):
    hdr = open(os.path.join(ROOT, "include", "imk.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(imk_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    from inconsistencymasks_amd import _lib
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = ctypes.CDLL(built_lib)
    for name in declared:
        getattr(lib, name)
    assert _lib.lib.imk_version() == 100
    assert _lib.lib.imk_error_string(-1) == b"invalid argument"


def test_missing_library_fails_loudly(tmp_path):
    code = ("import importlib.util, sys\n"
            f"spec = importlib.util.spec_from_file_location('x', r'{ROOT}/inconsistencymasks_amd/_lib.py')\n"
            "m = importlib.util.module_from_spec(spec)\n"
            f"import os; os.path.exists = lambda p, _e=os.path.exists: False if p.endswith('libimk.so') else _e(p)\n"
            "try:\n    spec.loader.exec_module(m)\n    print('LOADED')\nexcept RuntimeError as e:\n    print('RAISED', 'no CPU fallback' in str(e))\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True).stdout
    assert "RAISED True" in out


CFGS = [(256, 256, 3, 1, 0.5, "sigmoid"), (256, 256, 1, 3, 1.0, "sigmoid"), (256, 256, 3, 9, 1.0, "softmax"),
        (208, 416, 3, 35, 1.0, "softmax"), (256, 256, 3, 9, 2.0, "softmax"), (64, 64, 3, 1, 1.25, "sigmoid")]


@pytest.mark.parametrize("cfg", CFGS)
def test_plan_matches_oracle_layer_table(built_lib, cfg):
    from inconsistencymasks_amd.unet import Plan
    from oracle import unet_oracle as U
    h, w, c, k, alpha, act = cfg
    p = Plan(h, w, c, k, alpha, act)
    total, trainable = U.count_params(c, k, alpha)
    assert (p.n_total, p.n_trainable) == (total, trainable)
    table = U.layer_table(c, k, alpha)
    assert [l["name"] for l in p.layers] == [t[0] for t in table]
    off = 0
    for l, t in zip(p.layers, table):
        assert (l["kind"], l["ksize"], l["cin"], l["cout"]) == ({"conv": 0, "bn": 1}[t[1]], t[2], t[3], t[4])
        assert l["off_w"] == off
        if l["kind"] == 0:
            off += t[2] * t[2] * t[3] * t[4]
            assert l["off_b"] == off
            off += t[4]
        else:
            assert l["off_b"] == off + t[4]
            off += 2 * t[4]
    assert off == trainable
    assert p.workspace_bytes(8, 1) > p.workspace_bytes(8, 0) > 0
    assert p.workspace_bytes(16, 0) > p.workspace_bytes(8, 0)
    assert p.packed_bytes > 0 and p.state_bytes >= 8 * trainable


def test_published_param_counts(built_lib):
    """README.md:25 of the reference: 0.17 M ... 2.72 M parameters."""
    from inconsistencymasks_amd.unet import Plan
    assert Plan(256, 256, 3, 1, 0.5, "sigmoid").n_total == 171561
    assert Plan(256, 256, 3, 9, 2.0, "softmax").n_total == 2717865


def test_plan_rejects_bad_config(built_lib):
    from inconsistencymasks_amd._lib import ImkError
    from inconsistencymasks_amd.unet import Plan
    with pytest.raises(ImkError):
        Plan(250, 256, 3, 1, 0.5, "sigmoid")      # not a multiple of 16
    with pytest.raises(ValueError):
        Plan(256, 256, 3, 1, 0.5, "tanh")
    # include/imk.h: images of 2^24 pixels or more are refused (the conv kernels' 24-bit tile-offset arithmetic) ...
    with pytest.raises(ImkError, match="-2"):
        Plan(4096, 4096, 3, 1, 0.5, "sigmoid")
    with pytest.raises(ImkError, match="-2"):
        Plan(16, 65536, 3, 1, 0.5, "sigmoid")
    Plan(4096, 4080, 3, 1, 0.5, "sigmoid")        # ... one row less is a plan


def test_shard_list_partitions():
    from inconsistencymasks_amd.functions import shard_list
    names = [f"img_{i:04d}.png" for i in np.random.default_rng(0).permutation(2335)]
    for world in (1, 2, 4, 8):
        parts = [shard_list(names, r, world) for r in range(world)]
        assert sum(parts, []) == sorted(names)                       # contiguous blocks of the sorted list
        assert max(map(len, parts)) - min(map(len, parts)) <= 1


def test_metric_helpers_golden(golden_dir):
    from inconsistencymasks_amd import functions as F
    g = np.load(os.path.join(golden_dir, "metrics.npz"))
    for k in g["cases"]:
        assert F.get_IoU_binary(g[k + "_gt"], g[k + "_pr"]) == pytest.approx(float(g[k + "_iou"][0]), abs=1e-12)
        assert F.dice_score_numpy_binary(g[k + "_gt"], g[k + "_pr"]) == pytest.approx(float(g[k + "_dice"][0]), abs=1e-7)


def test_multiclass_and_hela_helpers_golden(golden_dir):
    from inconsistencymasks_amd import functions as F
    g = np.load(os.path.join(golden_dir, "metrics.npz"))
    for k in g["mc_cases"]:
        assert F.get_IoU_multi_unique(g[k + "_pr"], g[k + "_gt"]) == pytest.approx(float(g[k + "_iou"][0]), abs=1e-12)
        assert F.pixel_accuracy(g[k + "_pr"], g[k + "_gt"]) == pytest.approx(float(g[k + "_pa"][0]), abs=1e-12)
    pts = [tuple(int(v) for v in p) for p in g["dist_pts"]]
    for p, d in zip(pts, g["dist_min"]):
        assert F.get_min_dist(p, pts) == pytest.approx(float(d), abs=1e-9)


def test_hela_position_postprocessing_properties():
    """The contour/circle step is unpinned (needs OpenCV in the reference): property checks only."""
    from inconsistencymasks_amd import functions as F
    m = np.zeros((96, 96), np.uint8)
    centres = [(20, 20), (60, 30), (40, 70), (80, 80)]
    yy, xx = np.mgrid[0:96, 0:96]
    for cx, cy in centres:
        m[(yy - cy) ** 2 + (xx - cx) ** 2 <= 16] = 255
    pos = F.get_pos_contours(m)
    assert len(pos) == len(centres)
    for (px, py) in pos:
        assert min(abs(px - 1 - cx) + abs(py - 1 - cy) for cx, cy in centres) <= 1      # centre (+1 offset of the reference)
    out = F.mod_pos_size(m)
    assert len(F.get_pos_contours(out)) == len(centres)                                  # one disc per blob
    from scipy import ndimage
    lab, n = ndimage.label(out > 0)
    areas = ndimage.sum(out > 0, lab, range(1, n + 1))
    assert all(np.pi * 2.5 ** 2 <= a <= np.pi * 8.6 ** 2 for a in areas)                # radius clamped to [3, 8]
    assert F.get_cell_count(pos, m, np.zeros_like(m)) == (4, 0, 0)


def test_parse_mask_rule(tmp_path):
    """functions.py:975: uint8(mask/255) keeps only 255."""
    from inconsistencymasks_amd import functions as F
    os.makedirs(tmp_path / "images")
    os.makedirs(tmp_path / "masks")
    ramp = np.arange(256, dtype=np.uint8).reshape(16, 16)
    F.write_png(str(tmp_path / "masks" / "a.png"), ramp)
    F.write_png(str(tmp_path / "images" / "a.png"), np.stack([ramp] * 3, -1))
    img, mask = F.parse_image_ISIC_2018(str(tmp_path / "images" / "a.png"))
    assert img.shape == (16, 16, 3) and np.array_equal(img[..., 1], ramp)
    assert mask.shape == (16, 16, 1) and mask.sum() == 1 and mask.reshape(-1)[255] == 1


def _params(n, seed, **kw):
    import random
    from inconsistencymasks_amd import augment
    return augment.draw_params(n, rng=random.Random(seed), np_rng=np.random.RandomState(seed), **kw)


def test_augment_distribution_of_draws():
    """the host draws follow the reference's distributions (functions.py:2795-2826, :1494)"""
    prm = _params(4000, 5, free_rotation=True, max_blur=2, max_noise=10, brightness_range_alpha=(0.8, 1.2),
                  brightness_range_beta=(-10, 10))
    rot = np.bincount([q.rot for q in prm], minlength=4) / 4000
    assert np.all(np.abs(rot - 0.25) < 0.04)
    assert abs(np.mean([q.flip_h for q in prm]) - 0.5) < 0.04 and abs(np.mean([q.bright_on for q in prm]) - 0.5) < 0.04
    ks = sorted(set(q.blur_k for q in prm))
    assert ks == [0, 3, 5]
    assert all(0.8 <= q.alpha <= 1.2 and -10 <= q.beta <= 10 and q.noise_max == 10 for q in prm)
    prm = _params(200, 6, free_rotation=False)
    assert all(q.rot == 0 and q.flip_v == 0 for q in prm)


def _binary_counts(gt, pr):
    gn, p, gh = gt != 0, pr != 0, gt >= 128
    return [int((gn & p).sum()), int((gn | p).sum()), int(gh.sum()), int((pr >= 128).sum()), int((gh & (pr >= 128)).sum())]


def _multi_counts(gt, pr):
    c = np.zeros((4, 256), np.int64)
    c[0] = np.bincount(gt.ravel(), minlength=256)
    c[1] = np.bincount(pr.ravel(), minlength=256)
    c[2] = np.bincount(gt.ravel()[gt.ravel() == pr.ravel()], minlength=256)
    c[3, 0] = int((gt == pr).sum())
    return c


def test_metrics_from_integer_counts_match_reference_goldens_exactly(golden_dir):
    """the evaluation kernels return integer counts; the host formulas must give the reference's floats bit for bit"""
    from inconsistencymasks_amd import evaluate as E
    g = np.load(os.path.join(golden_dir, "metrics.npz"))
    for k in g["cases"]:
        iou, dice = E.iou_dice_from_counts(_binary_counts(g[k + "_gt"], g[k + "_pr"]))
        assert iou == float(g[k + "_iou"][0]) and float(dice) == float(g[k + "_dice"][0])
    for k in g["mc_cases"]:
        gt, pr = g[k + "_gt"], g[k + "_pr"]
        pa, iou = E.pa_iou_from_counts(_multi_counts(gt.astype(np.uint8), pr.astype(np.uint8)), gt.size)
        assert pa == float(g[k + "_pa"][0]) and iou == float(g[k + "_iou"][0])


@pytest.mark.parametrize("cfg", [(256, 256, 1, 3, 3, 2.0, True), (256, 256, 3, 1, 1, 1.0, False), (64, 128, 1, 3, 3, 0.5, True)])
def test_evalnet_plan_matches_oracle_layer_table(built_lib, cfg):
    """evalnet.py:24-73: Keras creation order (tower A, tower B, five trunk blocks, Dense head(s)) and parameter layout"""
    from inconsistencymasks_amd.evalnet import EvalPlan
    from oracle import evalnet_oracle as E
    h, w, ca, cb, k, alpha, two = cfg
    p = EvalPlan(h, w, ca, cb, k, alpha, two, True, not two)
    assert (p.n_total, p.n_trainable) == E.count_params(ca, cb, k, alpha, two)
    table = E.layer_table(ca, cb, k, alpha, two)
    assert [l["name"] for l in p.layers] == [t[0] for t in table]
    off = 0
    for l, t in zip(p.layers, table):
        assert (l["kind"], l["ksize"], l["cin"], l["cout"]) == ({"conv": 0, "bn": 1}[t[1]], t[2], t[3], t[4])
        assert l["off_w"] == off
        off += t[2] * t[2] * t[3] * t[4] + t[4] if l["kind"] == 0 else 2 * t[4]
    assert off == p.n_trainable
    assert p.workspace_bytes(8, 1) > p.workspace_bytes(8, 0) > 0 and p.packed_bytes > 0


def test_evalnet_param_count_by_hand(built_lib):
    """get_evalnet_miou(256, 256, 1, 3, alpha=2) (HeLa/14_HeLa_aug_IM++.py:106, config.ini:46), Keras count by hand:
    towers 10 624 + 10 688, trunk 19 648 + 22 912 + 90 880 + 361 984 + 1 444 864, two Dense(3) heads 2 x 1 539"""
    from inconsistencymasks_amd.evalnet import EvalPlan
    assert EvalPlan(256, 256, 1, 3, 3, 2.0, True, True, False).n_total == 1964678


def test_evalnet_plan_rejects_bad_config(built_lib):
    from inconsistencymasks_amd._lib import ImkError
    from inconsistencymasks_amd.evalnet import EvalPlan
    with pytest.raises(ImkError):
        EvalPlan(255, 256, 3, 1, 1, 1.0, False, True, True)      # the towers' pooled outputs are concatenated: even sizes
    with pytest.raises(ImkError):
        EvalPlan(48, 256, 3, 1, 1, 1.0, False, True, True)       # six poolings need at least 64 rows
    with pytest.raises(ImkError):
        EvalPlan(256, 256, 3, 9, 9, 1.0, True, True, False)      # a 9-channel uint8 mask stack (not one-hot): not built
    p = EvalPlan(208, 416, 3, 35, 35, 2.0, True, True, False, True)   # Cityscapes: floor pooling 13 -> 6 -> 3, 35 classes
    assert p.n_total > 0 and p.workspace_bytes(4, 1) > 0


def test_evalnet_oracle_losses_and_aug_count():
    """loss forms of functions.py:4708 (mse + bce on cached logits == clipped-probability bce away from saturation) and the
    augmentation count rule of functions.py:5921-5930 (known answers by definition)"""
    torch = pytest.importorskip("torch")
    from inconsistencymasks_amd.evalnet_functions import num_augs_from_miou
    from oracle import evalnet_oracle as E
    g = torch.Generator().manual_seed(0)
    logits = torch.randn((5, 6), generator=g)
    y = torch.rand((5, 6), generator=g)
    y[:, 3:] = (y[:, 3:] > 0.5).float()
    out = torch.sigmoid(logits)
    total, l0, l1 = E.loss_fn(out, logits, y, True)
    pc = out[:, 3:].clamp(1e-7, 1 - 1e-7)
    bce = -(y[:, 3:] * pc.log() + (1 - y[:, 3:]) * (1 - pc).log()).mean()
    assert abs(float(l1) - float(bce)) < 1e-6 and abs(float(l0) - float(((out[:, :3] - y[:, :3]) ** 2).mean())) < 1e-7
    assert abs(float(total) - float(l0 + l1)) < 1e-7
    lo, hi = 0.59, 0.62                                  # config.ini:54-55
    assert [num_augs_from_miou(v, lo, hi) for v in (0.0, 0.59, 0.5901, 0.597, 0.603, 0.609, 0.615, 0.62, 0.6201, 1.0)] == \
        [1, 1, 1, 2, 3, 4, 5, 5, 5, 5]


def test_classwise_label_helpers_golden(golden_dir):
    """compute_classwise_IoU / compute_classwise_detection_im / compute_classwise_detection (functions.py:4328-4459): the label
    arithmetic of the multiclass EvalNet training data, bit-identical to the reference's own outputs"""
    from inconsistencymasks_amd import evalnet_functions as EF
    g = np.load(os.path.join(golden_dir, "evalnet_labels.npz"))
    assert len(g["cases"]) == 5
    for k in g["cases"]:
        K = int(g[k + "_K"][0])
        gt, pred, im = g[k + "_gt"], g[k + "_pred"], g[k + "_im"] > 0
        assert np.array_equal(np.array(EF.compute_classwise_IoU(pred, gt, K), np.float64), g[k + "_iou"]), k
        counts = np.zeros(K)
        bins = np.bincount(gt.ravel(), minlength=K)
        counts[:len(bins)] += bins
        blocked = gt.copy()
        blocked[im] = 0
        assert np.array_equal(np.array(EF.compute_classwise_detection_im(blocked, K, counts, 0.3)), g[k + "_det_im"]), k
        assert np.array_equal(np.array(EF.compute_classwise_detection(pred, K)), g[k + "_det"]), k


def test_unseeded_models_differ(built_lib):
    """get_unet() without a seed must give a fresh initialisation per call, like Keras (unet.py:46);
    torch.manual_seed() makes the sequence repeatable."""
    import torch
    from inconsistencymasks_amd.unet import get_unet
    a = get_unet(64, 64, 3, 1, 0.5, "relu", "sigmoid", device="cpu")
    b = get_unet(64, 64, 3, 1, 0.5, "relu", "sigmoid", device="cpu")
    assert not torch.equal(a.params, b.params)
    torch.manual_seed(5)
    c = get_unet(64, 64, 3, 1, 0.5, "relu", "sigmoid", device="cpu")
    torch.manual_seed(5)
    d = get_unet(64, 64, 3, 1, 0.5, "relu", "sigmoid", device="cpu")
    assert torch.equal(c.params, d.params)
    e = get_unet(64, 64, 3, 1, 0.5, "relu", "sigmoid", seed=3, device="cpu")
    f = get_unet(64, 64, 3, 1, 0.5, "relu", "sigmoid", seed=3, device="cpu")
    assert torch.equal(e.params, f.params)


def test_reference_module_names_resolve(built_lib):
    """`from functions import ...`, `from unet import get_unet`, `import paths`, and the five TensorFlow names
    the reference's scripts touch (inconsistencymasks_amd/compat) resolve to this implementation, with the reference's
    positional signatures (ISIC_2018/09_ISIC_2018_IM.py:5-16, 75-78, 90-120)."""
    import inspect
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from functions import train_ISIC_2018, create_pseudo_labels_im_ISIC_2018, dice_loss, train_multiclass, train_hela\n"
            "from functions import create_pseudo_labels_im_multiclass, create_pseudo_labels_im_hela, BATCH_SIZE, dilate_mask\n"
            "from unet import get_unet\nfrom evalnet import get_evalnet, get_evalnet_miou\nimport paths\n"
            "import tensorflow as tf\nfrom tensorflow.keras import mixed_precision\n"
            "mixed_precision.set_global_policy('mixed_float16')\n"
            "with tf.device('/gpu:0'):\n    pass\n"
            "assert callable(tf.keras.models.load_model) and callable(tf.keras.backend.clear_session)\n"
            "tf.keras.losses.CategoricalCrossentropy()\n"
            "print(paths.ISIC_2018_TRAIN_UNLABELED_IMAGES_DIR, paths.HELA_VAL_BRIGHTFIELD_DIR, paths.CITYSCAPES_MODEL_DIR)\n"
            % (ROOT, os.path.join(ROOT, "inconsistencymasks_amd", "compat")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "train_unlabeled" in r.stdout and "brightfield" in r.stdout
    from inconsistencymasks_amd import functions as F
    sig = lambda f: list(inspect.signature(f).parameters)
    assert sig(F.create_pseudo_labels_im_ISIC_2018) == ["models", "h", "w", "c", "images_path", "main_output_path", "rgb",
                                                        "erode_kernel", "dilate_kernel", "block_input", "block_output",
                                                        "filter_bad_predictions"]
    d = inspect.signature(F.create_pseudo_labels_im_multiclass).parameters
    assert d["erode_kernel"].default == 5 and d["dilate_kernel"].default == 5      # the reference's defaults


def test_color_tables_follow_the_public_palettes():
    """SUIM/SUIM_class_mapping.py:4-14 and Cityscapes/Cityscapes_class_mapping.py:43-80, generated by rule."""
    from inconsistencymasks_amd.im_driver import color_mapping
    s = color_mapping("SUIM", 9)
    assert s[(211, 211, 211)] == 0 and s[(0, 0, 0)] == 1 and s[(0, 0, 255)] == 2 and s[(255, 0, 0)] == 5 and s[(255, 255, 255)] == 8
    assert len(s) == 9
    c = color_mapping("Cityscapes", 35)
    assert c[(0, 0, 0)] == 0 and c[(0, 0, 128)] == 1 and c[(128, 128, 128)] == 7 and c[(0, 0, 64)] == 8
    assert c[(64, 0, 0)] == 32 and c[(64, 128, 0)] == 34 and c[(192, 192, 192)] == -1 and len(c) == 36
    assert len(color_mapping("SUIM", 3)) == 3          # toy configurations fall back to a generated palette


def test_keras_checkpoint_converter_mapping(built_lib, tmp_path):
    """tools/keras_h5_to_safetensors.py (runs offline where h5py exists): its layer table is the plan's, Keras'
    auto-numbered layer names map to ours by creation order whatever the session's counter offset, HWIO kernels pass through
    unchanged, and the `model.get_weights()`-order export round-trips through load_model."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import keras_h5_to_safetensors as K
    import torch
    from inconsistencymasks_amd import functions as F
    from inconsistencymasks_amd.unet import Plan, UNet
    from oracle import unet_oracle as U
    for c, k, alpha in ((3, 1, 0.5), (1, 3, 1.0), (3, 35, 1.25)):
        table = K.layer_table(c, k, alpha)
        assert table == U.layer_table(c, k, alpha)
        p = Plan(64, 64, c, k, alpha, "sigmoid")
        assert [(l["name"], {0: "conv", 1: "bn"}[l["kind"]], l["ksize"], l["cin"], l["cout"]) for l in p.layers] == table
    table = K.layer_table(3, 1, 0.5)
    # a model built as the 3rd in a Keras session: conv2d_48 ... conv2d_70, batch_normalization_28 ... _41, 'out'
    m = UNet(64, 64, 3, 1, 0.5, "sigmoid", seed=5, device="cpu")
    sd = {kk: v.numpy() for kk, v in m.state_dict().items()}
    weights_of, ci, bi = {}, 48, 28
    for name, kind, *_ in table:
        if kind == "conv":
            kn = "out" if name == "out" else f"conv2d_{ci}"
            ci += name != "out"
            weights_of[kn] = {"kernel": sd[name + ".w"], "bias": sd[name + ".b"]}
        else:
            weights_of[f"batch_normalization_{bi}"] = {"gamma": sd[name + ".gamma"], "beta": sd[name + ".beta"],
                                                       "moving_mean": sd[name + ".mean"], "moving_variance": sd[name + ".var"]}
            bi += 1
    names = K.match_keras_layers(list(weights_of), table)
    assert names["in.c"] == "conv2d_48" and names["e1.c3"] == "conv2d_49" and names["d9.c1"] == "conv2d_70" and names["out"] == "out"
    assert names["in.bn"] == "batch_normalization_28" and names["d9.bnb"] == "batch_normalization_41"
    back = K.state_dict_from_keras(weights_of, table)
    assert set(back) == set(sd) and all(np.array_equal(back[kk], sd[kk]) for kk in sd)
    assert K.infer_config(sd["in.c.w"].shape, sd["out.w"].shape) == (3, 1, 0.5)
    with pytest.raises(ValueError):
        K.match_keras_layers([n for n in weights_of if n != "conv2d_60"], table)
    # safetensors -> get_weights() order -> safetensors, through the CLI and load_model
    src, npz, dst = str(tmp_path / "a.h5"), str(tmp_path / "w.npz"), str(tmp_path / "b.h5")
    F.save_model(m, src)
    assert K.main(["--to-keras-npz", src, npz]) == 0
    d = np.load(npz)
    assert len(d.files) == 2 * 24 + 4 * 14 and d["w000"].shape == (1, 1, 3, 8)       # Keras: kernel, bias, gamma, beta, mean, var ...
    assert K.main(["--from-keras-npz", npz, dst, "--height", "64", "--width", "64", "--act-out", "sigmoid"]) == 0
    m2 = F.load_model(dst, device="cpu")
    assert torch.equal(m2.params, m.params) and (m2.plan.alpha, m2.plan.act_out) == (0.5, "sigmoid")
    # the file a TensorFlow user writes with np.savez(path, *model.get_weights()): arrays named arr_0 ... arr_103 (numeric order)
    npz2, dst2 = str(tmp_path / "arr.npz"), str(tmp_path / "c.h5")
    np.savez(npz2, *[d[k] for k in sorted(d.files)])
    assert K.main(["--from-keras-npz", npz2, dst2, "--height", "64", "--width", "64", "--act-out", "sigmoid"]) == 0
    assert torch.equal(F.load_model(dst2, device="cpu").params, m.params)


def test_background_png_writes_flush(built_lib, tmp_path):
    """write_png_async / flush_writes: every queued file exists and round-trips after the flush; failures are re-raised."""
    from inconsistencymasks_amd import functions as F
    rng = np.random.default_rng(0)
    arrs = {f"a_{i}.png": rng.integers(0, 256, (17, 23, 3)).astype(np.uint8) for i in range(40)}
    arrs["g.png"] = rng.integers(0, 256, (9, 11)).astype(np.uint8)
    for n, a in arrs.items():
        F.write_png_async(str(tmp_path / n), a)
    F.flush_writes()
    assert not F._PENDING
    for n, a in arrs.items():
        back = F.read_png(str(tmp_path / n), 3 if a.ndim == 3 else 1)
        assert np.array_equal(back if a.ndim == 3 else back[..., 0], a)
    F.write_png_async(str(tmp_path / "no_such_dir" / "x.png"), arrs["g.png"])
    with pytest.raises(Exception):
        F.flush_writes()
    assert not F._PENDING


def test_dilate_mask_is_grey_dilation_in_the_oracle():
    """The identity the product relies on (functions.dilate_mask = 3x3 grey-level dilation of the class-id map): the
    reference's per-class loop with later classes overwriting earlier ones (functions.py:3075-3100) is a neighbourhood
    maximum -- checked here on the oracle's literal restatement of that loop."""
    from scipy import ndimage
    from oracle import im_oracle as O
    rng = np.random.default_rng(2)
    for k in (2, 9, 35):
        m = rng.integers(0, k, (37, 41)).astype(np.uint8)
        m[rng.random(m.shape) < 0.5] = 0
        assert np.array_equal(O.dilate_mask_per_class(m, 3), ndimage.grey_dilation(m, size=(3, 3), mode="constant", cval=0))


def test_position_contour_centres_known_answers():
    """get_pos_contours (functions.py:6181-6218) restated without OpenCV: polygon through the border pixels' centres,
    Green's-theorem moments, int(m10 / m00) + 1.  Known answers that follow from that definition: an axis-aligned
    rectangle's centre, a disc's centre, and NO position for blobs whose contour polygon has zero area (a single pixel, a
    one-pixel-wide line) -- cv2.moments gives m00 = 0 for those and the reference skips them."""
    from inconsistencymasks_amd import functions as F
    from oracle import hela_geometry as G
    m = np.zeros((40, 40), np.uint8)
    m[10:21, 5:16] = 255                                   # rows 10..20, columns 5..15: centre (x 10, y 15)
    assert F.get_pos_contours(m, erode_kernel=0) == [(11, 16)]
    assert F.get_pos_contours(m) == [(11, 16)]             # a 3x3 erosion shrinks the square around the same centre
    thin = np.zeros((20, 20), np.uint8)
    thin[5, 5] = 255
    thin[10, 3:9] = 255
    assert F.get_pos_contours(thin, erode_kernel=0) == []
    yy, xx = np.mgrid[0:64, 0:64]
    disc = (((yy - 30) ** 2 + (xx - 20) ** 2) <= 36).astype(np.uint8) * 255
    assert F.get_pos_contours(disc, erode_kernel=0) == [(21, 31)]
    # the polygon of a 3x3 ring of pixels is the 2x2 square through their centres
    ring = np.pad(np.array([[1, 1, 1], [1, 0, 1], [1, 1, 1]], bool), 1)
    assert G._polygon_moments(G._trace_outer_border(ring)) == (4.0, 8.0, 8.0)
    # RETR_TREE: a hole is a contour of its own (the blob pixels around it) and adds a position -- a square frame has two
    # concentric contours, both centred on the frame; a one-pixel hole's contour is the diamond of its 4 neighbours (area 2)
    frame = np.zeros((40, 40), np.uint8)
    frame[10:21, 5:16] = 255
    frame[13:18, 8:13] = 0
    assert F.get_pos_contours(frame, erode_kernel=0) == [(11, 16), (11, 16)]
    dot = np.zeros((40, 40), np.uint8)
    dot[10:21, 5:16] = 255
    dot[12, 7] = 0
    assert F.get_pos_contours(dot, erode_kernel=0) == [(11, 16), (8, 13)]
    two = np.zeros((40, 60), np.uint8)                      # holes belong to their own blob only
    two[5:16, 5:16] = 255
    two[8:13, 8:13] = 0
    two[20:31, 30:41] = 255
    assert sorted(F.get_pos_contours(two, erode_kernel=0)) == [(11, 11), (11, 11), (36, 26)]


def test_failed_background_png_write_fails_the_process(tmp_path):
    """write_png_async + interpreter exit: a write that fails in the background (here: the target directory does not exist) must not
    end in exit status 0 -- atexit handlers' exceptions are ignored by the interpreter, so the handler exits itself."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, numpy as np; sys.path.insert(0, %r)\n"
            "from inconsistencymasks_amd import functions as F\n"
            "F.write_png_async(%r, np.zeros((4, 4), np.uint8))\n") % (root, str(tmp_path / "missing_dir" / "x.png"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1, (r.returncode, r.stderr[-500:])


def test_package_import_raises_the_hardware_queue_limit():
    """More streams than the HIP runtime's hardware queues serialise "concurrent" work (profiles/README.md, round 3): importing the
    package before the first HIP call sets GPU_MAX_HW_QUEUES=8 unless the caller chose a value."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import os, sys; sys.path.insert(0, %r); import inconsistencymasks_amd; print(os.environ['GPU_MAX_HW_QUEUES'])" % root
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip() == "8"
    env["GPU_MAX_HW_QUEUES"] = "4"
    assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip() == "4"


def test_dp_bn_momentum_rule(monkeypatch):
    """Keras' 0.99 per step at any world size unless IMK_DP_BN_MOMENTUM=scaled opts into 0.99^N (the moving statistics keep
    their memory in samples; profiles/r03_dp_convergence.txt); one rank is always the reference's recipe."""
    from inconsistencymasks_amd import functions as F
    monkeypatch.delenv("IMK_DP_BN_MOMENTUM", raising=False)
    assert F.dp_bn_momentum_rule(1) == ("reference", 0.99)
    assert F.dp_bn_momentum_rule(8) == ("reference", 0.99)
    monkeypatch.setenv("IMK_DP_BN_MOMENTUM", "reference")
    assert F.dp_bn_momentum_rule(8) == ("reference", 0.99)
    monkeypatch.setenv("IMK_DP_BN_MOMENTUM", "scaled")
    name, m = F.dp_bn_momentum_rule(8)
    assert name == "scaled" and abs(m - 0.99 ** 8) < 1e-12
    assert F.dp_bn_momentum_rule(1) == ("reference", 0.99) and F.dp_bn_momentum_rule(2)[0] == "scaled"
    monkeypatch.setenv("IMK_DP_BN_MOMENTUM", "bogus")
    with pytest.raises(ValueError):
        F.dp_bn_momentum_rule(2)


def test_color_mask_table_equals_the_reference_loop(tmp_path):
    """convert_class_to_color_mask (functions.py:6127-6149) as one table gather: identical to the reference's per-class masked
    assignment for the SUIM and Cityscapes palettes, for ids without a colour (black) and for a class listed twice (last wins)."""
    from inconsistencymasks_amd import functions as F, im_driver
    rng = np.random.default_rng(0)
    cases = [(im_driver.color_mapping("SUIM", 9), 11), (im_driver.color_mapping("Cityscapes", 35), 40),
             ({(1, 2, 3): 1, (9, 9, 9): 1, (5, 5, 5): 2}, 4)]
    for i, (mapping, hi) in enumerate(cases):
        cm = rng.integers(0, hi, (37, 53)).astype(np.uint8)
        want = np.zeros(cm.shape + (3,), np.uint8)
        for col, cls in mapping.items():
            want[cm == cls] = col
        p = str(tmp_path / f"c{i}.png")
        F.convert_class_to_color_mask(cm, p, mapping)
        F.flush_writes()
        assert np.array_equal(F.read_png(p, 3), want)


BENCH_LINE_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                   "dtype", "data", "stage_ms", "config", "roofline", "detail"}


def check_bench_line(text, expect_cpu_baseline):
    """the driver's view of bench.py: the LAST stdout line, parsed as JSON, under 4 KB, with the contract's keys (a
    21 KB line was cut by the driver's 8 KB stdout tail and the round went unmeasured)"""
    import json
    last = text.strip().splitlines()[-1]
    assert len(last) < 4096, len(last)
    out = json.loads(last)
    assert BENCH_LINE_KEYS <= set(out), BENCH_LINE_KEYS - set(out)
    for k in ("workload", "shape", "alpha", "n_models", "unlabeled_images", "labeled_images", "infer_batch", "infer_batch_rule",
              "train_batch_per_gpu", "global_batch", "parallelism", "process_group", "epoch_steps", "kept", "bn_momentum"):
        assert k in out["config"], k
    assert "model" not in out["config"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "frac_rocprof", "live", "replayed_from", "by_stage"):
        assert k in out["roofline"], k
    assert out["roofline"]["bound"] in ("hbm", "mfma")
    if expect_cpu_baseline:
        for k in ("value", "unit", "cores", "kind", "sample", "host_cpus", "cpu_model", "t_infer_per_image_s", "t_train_step_s", "parity_sample"):
            assert k in out["cpu_baseline"], k
    return out


def test_bench_line_budget():
    """bench.compact_line on canned full records (the 21 KB line round 5 printed, and the same with every optional part at its
    largest): under the budget, contract keys present, the full record kept beside it."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench.json")))
    assert len(json.dumps(full)) > 20000
    line = bench.compact_line(full, "gpurun_out/bench_detail.json")
    out = check_bench_line(json.dumps(line), expect_cpu_baseline=True)
    assert out["value"] == full["value"] and out["ms_per_step"] == full["ms_per_step"]
    assert out["roofline"]["frac"] == full["roofline"]["frac"] and out["cpu_baseline"]["value"] == full["cpu_baseline"]["value"]
    assert set(out["other_configs"]) == {"fields", "suim", "cityscapes", "hela", "cityscapes_a2"}
    # worst case: long error strings in other_configs, a sharding check, a long workload name -- the guard sheds optional parts
    fat = json.loads(json.dumps(full))
    fat["other_configs"] = {f"cfg{i}": {"error": "RuntimeError: " + "x" * 4000} for i in range(8)}
    fat["sharding_check"] = {"sum_pred_size": 1e9, "sum_im_size": 1e8, "kept": 2307.0, "equals_sum_over_ranks": True}
    fat["cpu_baseline"]["sample"] = "s" * 3000
    line = bench.compact_line(fat, "/tmp/bench_detail.json")
    assert len(json.dumps(line)) < bench.LINE_BUDGET
    assert BENCH_LINE_KEYS <= set(line) and "cpu_baseline" in line and line["cpu_baseline"]["value"] == full["cpu_baseline"]["value"]


def test_bench_replay_provenance(tmp_path, monkeypatch):
    """A replayed value (traffic, frac_rocprof, by_stage kernels) is printed only when the committed profile was collected with the
    running bench.py and kernel sources; the per-configuration file tag never falls back to another configuration's files."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    assert bench.config_tag("isic", None) == "" and bench.config_tag("isic", 0.5) == ""
    assert bench.config_tag("suim", None) == "_suim" and bench.config_tag("cityscapes", 2.0) == "_cityscapes_a2"
    assert bench.config_tag("cityscapes", 1.25) == "_cityscapes_a125" and bench.config_tag("cityscapes", 1.0) == "_cityscapes"
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (tmp_path / "profiles").mkdir()
    ok, why, _ = bench.provenance("")
    assert not ok and "missing" in why
    rec = {"bench_py_sha16": bench.bench_py_sha16(), "lib_build_id": bench.lib_build_id()}
    (tmp_path / "profiles" / f"{bench.ROUND}_provenance.json").write_text(json.dumps(rec))
    assert bench.provenance("")[0]
    (tmp_path / "profiles" / f"{bench.ROUND}_provenance_suim.json").write_text(json.dumps({**rec, "lib_build_id": "0" * 16}))
    ok, why, _ = bench.provenance("_suim")
    assert not ok and "lib_build_id" in why


def test_infer_batch_rule():
    from inconsistencymasks_amd import functions as F
    assert F.infer_batch_size(0.5) == 584 and F.infer_batch_size(1.0) == 128 and F.infer_batch_size(2.0, 64) == 64
    assert F.infer_batches(2335, 584) == [(0, 584), (584, 1168), (1168, 1752), (1752, 2335)]
    assert F.infer_batches(292, 256) == [(0, 292)]                      # a 36-image tail is spread
    assert F.infer_batches(640, 256) == [(0, 256), (256, 512), (512, 640)]
    assert F.infer_batches(0, 256) == [] and F.infer_batches(5, 256) == [(0, 5)]
    for n in (1, 63, 64, 65, 257, 300, 1000, 2468):
        r = F.infer_batches(n, 128)
        assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))


def test_write_backlog_is_bounded_and_reader_pool_refuses_nested_use(built_lib, tmp_path, monkeypatch):
    """(a) a thread that queues more than IMK_MAX_PENDING_WRITES tasks waits for its oldest half first -- the backlog
    (and the arrays it pins) stays bounded, every file still arrives; (b) the shared reader pool asserts when one of its own tasks
    uses it (a task waiting for the pool it runs on can starve it); (c) the decode-cache signature sees a same-sized file replaced by
    an OLDER one."""
    from inconsistencymasks_amd import functions as F
    monkeypatch.setattr(F, "_MAX_PENDING_WRITES", 8)
    a = np.random.default_rng(1).integers(0, 256, (8, 8), dtype=np.uint8)
    peak = 0
    for i in range(100):
        F.write_png_async(str(tmp_path / f"w_{i}.png"), a)
        peak = max(peak, len(F._PENDING.get(__import__("threading").get_ident(), [])))
    assert peak <= 9
    F.flush_writes()
    assert len(os.listdir(tmp_path)) == 100
    with F._pool() as pool:
        with pytest.raises(AssertionError):
            pool.map(lambda i: pool.map(lambda j: j, range(2)), range(3))
    f1, f2 = tmp_path / "w_0.png", tmp_path / "w_1.png"
    sig = lambda: F._stat_sig([os.stat(f1), os.stat(f2)])
    before = sig()
    st = os.stat(f1)
    os.utime(f1, ns=(st.st_atime_ns, st.st_mtime_ns - 10**9))      # same size, OLDER mtime: sum-of-sizes + newest-mtime saw nothing
    assert sig() != before
