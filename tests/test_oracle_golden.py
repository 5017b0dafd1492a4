"""Pins the CPU oracle (numpy + C) to the golden vectors produced from the real reference
(tests/golden/make_golden.py).  CPU only."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import im_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


@pytest.fixture(scope="session")
def c_oracle():
    so = os.path.join(ROOT, "oracle", "libim_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    lib = ctypes.CDLL(so)
    return lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def c_binary(lib, preds, thr, ge, img=None, bi=0, bo=0):
    n, h, w, kb = preds.shape
    hw = h * w
    preds = np.ascontiguousarray(preds, np.float32)
    final = np.zeros((kb, h, w), np.uint8)
    im = np.zeros((h, w), np.uint8)
    ims = np.zeros(kb, np.int64)
    ps = np.zeros(kb, np.int64)
    c = 0 if img is None else img.shape[-1]
    img_out = None if img is None else np.zeros_like(img)
    lib.oracle_im_binary(_p(preds, ctypes.c_float), n, hw, kb, ctypes.c_float(thr), int(ge),
                         None if img is None else _p(np.ascontiguousarray(img), ctypes.c_uint8), c, bi, bo,
                         None if img is None else _p(img_out, ctypes.c_uint8),
                         _p(final, ctypes.c_uint8), _p(im, ctypes.c_uint8),
                         _p(ims, ctypes.c_int64), _p(ps, ctypes.c_int64))
    return final, im, ims, ps, img_out


def c_multi(lib, probs, img=None, bi=0, bo=0):
    n, h, w, k = probs.shape
    probs = np.ascontiguousarray(probs, np.float32)
    final = np.zeros((h, w), np.uint8)
    im = np.zeros((h, w), np.uint8)
    ims = np.zeros(1, np.int64)
    pres = np.zeros((n, k), np.uint8)
    c = 0 if img is None else img.shape[-1]
    img_out = None if img is None else np.zeros_like(img)
    lib.oracle_im_multiclass(_p(probs, ctypes.c_float), n, h * w, k,
                             None if img is None else _p(np.ascontiguousarray(img), ctypes.c_uint8), c, bi, bo,
                             None if img is None else _p(img_out, ctypes.c_uint8),
                             _p(final, ctypes.c_uint8), _p(im, ctypes.c_uint8), _p(ims, ctypes.c_int64),
                             _p(pres, ctypes.c_uint8))
    return final, im, ims[0], pres, img_out


def test_binary_golden(golden_dir, c_oracle):
    g = load(golden_dir, "im_binary.npz")
    assert len(g["cases"]) >= 30
    for k in g["cases"]:
        preds = g[k + "_preds"][:, 0]           # [N,H,W,1]
        r = O.im_binary(preds, 0.5, False)
        assert np.array_equal(r["final"][0], g[k + "_final"]), k
        assert np.array_equal(r["im"], g[k + "_im"]), k
        assert [int(r["im_size"]), int(r["pred_size"])] == g[k + "_sizes"].tolist(), k
        f, im, ims, ps, _ = c_binary(c_oracle, preds, 0.5, False)
        assert np.array_equal(f[0], g[k + "_final"]) and np.array_equal(im, g[k + "_im"]), k
        assert [int(ims[0]), int(ps[0])] == g[k + "_sizes"].tolist(), k


def test_hela_golden(golden_dir, c_oracle):
    g = load(golden_dir, "im_hela.npz")
    for k in g["cases"]:
        preds = g[k + "_preds"][:, 0]           # [N,H,W,3]
        r = O.im_binary(preds, 0.5, True)
        for c, nm in enumerate(("alive", "dead", "pos")):
            assert np.array_equal(r["final"][c], g[f"{k}_{nm}"]), (k, nm)
        assert np.array_equal(r["im"], g[k + "_im"]), k
        assert int(r["im_size"]) == int(g[k + "_sizes"][0]), k
        f, im, ims, ps, _ = c_binary(c_oracle, preds, 0.5, True)
        assert np.array_equal(f, r["final"]) and np.array_equal(im, r["im"]) and int(ims.sum()) == int(r["im_size"])


def test_multiclass_golden(golden_dir, c_oracle):
    g = load(golden_dir, "im_multiclass.npz")
    assert len(g["cases"]) >= 80
    for k in g["cases"]:
        probs = g[k + "_preds"][:, 0]           # [N,H,W,K]
        r = O.im_multiclass(probs, True)
        assert np.array_equal(r["final"], g[k + "_final"]), k
        assert np.array_equal(r["im"], g[k + "_im"]), k
        assert int(r["im_size"]) == int(g[k + "_sizes"][0]), k
        assert int(r["lists_equal"]) == int(g[k + "_lists_equal"][0]), k
        assert O.im_multiclass(probs, False)["lists_equal"] is True
        f, im, ims, pres, _ = c_multi(c_oracle, probs)
        assert np.array_equal(f, g[k + "_final"]) and np.array_equal(im, g[k + "_im"]) and int(ims) == int(r["im_size"])
        assert np.array_equal(pres, r["presence"]), k


def _writer_expect_isic(g, tag, bi, bo, filt):
    """Re-derive the reference writer's output files from the oracle primitives."""
    names = [str(n) for n in g["names"]]
    out, sizes = {}, []
    for i, name in enumerate(names):
        bgr = g[f"img_{i}"]
        preds = np.stack([g[f"pred_{i}_{n}"][0] for n in range(2)], 0)
        r = O.im_binary(preds, 0.5, False)
        sizes.append(r["im_size"])
        img, (mask,) = O.block(bgr, [r["final"][0]], r["im"], bi, bo)
        if O.keep_isic(r["pred_size"], r["im_size"], filt):
            out["images/" + name] = img
            out["masks/" + name] = mask
        out["im/" + name] = r["im"]
    return out, O.mean_im_size(sizes)


def test_writer_isic_golden(golden_dir):
    g = load(golden_dir, "writer_isic.npz")
    for tag in g["combos"]:
        tag = str(tag)
        bi, bo, filt = (tag[2] == "1"), (tag[6] == "1"), (tag[9] == "1")
        exp, mean = _writer_expect_isic(g, tag, bi, bo, filt)
        files = sorted(str(f) for f in g[tag + "_files"])
        assert files == sorted(exp), tag
        for f in files:
            assert np.array_equal(exp[f], g[tag + "/" + f]), (tag, f)
        assert mean == float(g[tag + "_mean"][0])
    # the filter must actually drop something in this fixture
    assert len(g["bi1_bo1_f1_files"]) < len(g["bi1_bo1_f0_files"])


def test_writer_multi_golden(golden_dir):
    g = load(golden_dir, "writer_multi.npz")
    names = [str(n) for n in g["names"]]
    for tag in g["combos"]:
        tag = str(tag)
        bi, bo, filt = (tag[2] == "1"), (tag[6] == "1"), (tag[9] == "1")
        exp, sizes = {}, []
        for i, name in enumerate(names):
            bgr = g[f"img_{i}"]
            probs = np.stack([g[f"pred_{i}_{n}"][0] for n in range(3)], 0)
            r = O.im_multiclass(probs, filt)
            sizes.append(r["im_size"])
            img, (mask,) = O.block(bgr, [r["final"]], r["im"], bi, bo)
            if (not filt) or r["lists_equal"]:
                exp["images/" + name] = img
                exp["masks/" + name] = mask
            exp["im/" + name] = r["im"]
        files = sorted(str(f) for f in g[tag + "_files"])
        assert files == sorted(exp), tag
        for f in files:
            assert np.array_equal(exp[f], g[tag + "/" + f]), (tag, f)
        assert O.mean_im_size(sizes) == float(g[tag + "_mean"][0])
    assert len(g["bi1_bo1_f1_files"]) < len(g["bi1_bo1_f0_files"])


def test_c_oracle_blocking(c_oracle):
    rng = np.random.default_rng(0)
    preds = rng.random((3, 9, 11, 1), dtype=np.float32)
    img = rng.integers(1, 255, (9, 11, 3)).astype(np.uint8)
    r = O.im_binary(preds)
    for bi in (0, 1):
        for bo in (0, 1):
            f, im, ims, ps, io = c_binary(c_oracle, preds, 0.5, False, img, bi, bo)
            ei, (em,) = O.block(img, [r["final"][0]], r["im"], bi, bo)
            assert np.array_equal(io, ei) and np.array_equal(f[0], em) and np.array_equal(im, r["im"])


def test_morphology_properties():
    rng = np.random.default_rng(1)
    m = (rng.random((20, 30)) > 0.6).astype(np.uint8) * 255
    for k in (3, 5):
        e, d = O.erode(m, k), O.dilate(m, k)
        assert np.all(e <= m) and np.all(d >= m)
        assert np.array_equal(O.erode(np.full_like(m, 255), k), np.full_like(m, 255))  # border never erodes a full mask
        assert np.array_equal(O.dilate(np.zeros_like(m), k), np.zeros_like(m))
    one = np.zeros((9, 9), np.uint8)
    one[4, 4] = 255
    assert O.dilate(one, 3).sum() == 9 * 255 and O.dilate(one, 5).sum() == 25 * 255
    assert np.array_equal(O.erode(O.dilate(one, 3), 3), one)


# ---- augmentation oracle (unpinned against cv2; structural checks) ---------------------------------------------
def test_aug_oracle_geometry_and_blur_properties():
    from oracle import aug_oracle as A
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (12, 12, 3), dtype=np.uint8)
    # flips / rotations compose like the cv2 calls they restate
    assert np.array_equal(A.geometric(img, 1, 0, 0), np.flipud(img))
    assert np.array_equal(A.geometric(img, 0, 1, 0), np.fliplr(img))
    assert np.array_equal(A.geometric(img, 0, 0, 1)[0], img[::-1, 0])          # 90 CW: first row = first column, bottom-up
    assert np.array_equal(A.geometric(A.geometric(img, 0, 0, 1), 0, 0, 3), img)
    assert np.array_equal(A.geometric(img, 0, 0, 2), img[::-1, ::-1])
    # blur: constant images are fixed points, kernels are normalised, result within the neighbourhood's range
    for k in (3, 5, 7):
        const = np.full((9, 9, 1), 77, np.uint8)
        assert np.array_equal(A.gaussian_blur(const, k), const)
        b = A.gaussian_blur(img, k)
        assert b.min() >= img.min() and b.max() <= img.max()
    # 3x3 known answer: centre impulse 255 -> [1 2 1]^T[1 2 1]/16 * 255, rounded half up
    imp = np.zeros((5, 5, 1), np.uint8); imp[2, 2, 0] = 255
    want = np.floor(np.outer([1, 2, 1], [1, 2, 1]) * 255 / 16 + 0.5).astype(np.uint8)
    assert np.array_equal(A.gaussian_blur(imp, 3)[1:4, 1:4, 0], want)
    # brightness: saturation and the absolute value
    a = np.array([[[0], [100], [255]]], np.uint8)
    assert A.convert_scale_abs(a, 1.5, -20).ravel().tolist() == [20, 130, 255]
    # noise range [-m, m)
    n = A.noise_field((64, 64, 3), 5, 1234)
    assert n.min() == -5 and n.max() == 4
    assert not np.array_equal(n, A.noise_field((64, 64, 3), 5, 1235))


def test_aug_oracle_noise_arithmetic_matches_reference_add_noise(golden_dir):
    """add_noise is the one numpy-only piece of the reference's augmentation chain: its outputs (with the noise field it
    drew) pin the oracle's add / clip / dtype arithmetic, which the GPU kernel is then compared with bit for bit"""
    from oracle import aug_oracle as A
    g = np.load(os.path.join(golden_dir, "augment.npz"))
    for k in g["cases"]:
        img, noise, want, m = g[k + "_img"], g[k + "_noise"], g[k + "_out"], int(g[k + "_m"][0])
        assert noise.min() >= -m and noise.max() <= m - 1            # the reference's range: [-m, m)
        got = A.apply_noise(img, noise.astype(np.int64))
        assert got.dtype == np.uint8 and np.array_equal(got, want)
