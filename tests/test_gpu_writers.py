"""The PRODUCT writers (`functions.create_pseudo_labels_im_ISIC_2018` / `_multiclass` / `_hela`) against the golden
outputs of the reference's own writers (tests/golden/writer_isic.npz, writer_multi.npz: the reference functions driven
over an in-memory directory, functions.py:2832-2891, 2988-3070): written file sets, pixel content, mean_im_size, all
eight BI / BO / filter combinations, bit-exact.  The models are prediction tables (`.predict` looks the image up), the
same fake models the fixtures were generated with, so what is under test is everything around the networks: PNG I/O,
RGB / BGR handling, the IM kernels, blocking, the keep rules, the mean.  The EK / DK > 0 branches (the reference's
DEFAULT arguments; unpinned because they need OpenCV) are checked against the oracle's restatement."""
import hashlib
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import im_oracle as O  # noqa: E402


class LookupModel:
    """prediction looked up by image content (the prepared RGB image, as the reference feeds it)"""

    def __init__(self, table):
        self.table = table

    def predict(self, x):
        x = x[0] if isinstance(x, (list, tuple)) else x
        return self.table[hashlib.sha1(np.ascontiguousarray(np.asarray(x)[0]).tobytes()).hexdigest()]


def _fixture(golden_dir, fname, n_models):
    from inconsistencymasks_amd import functions as F
    g = np.load(os.path.join(golden_dir, fname))
    names = [str(n) for n in g["names"]]
    tables = [dict() for _ in range(n_models)]
    images = {}
    for i, name in enumerate(names):
        rgb = np.ascontiguousarray(g[f"img_{i}"][..., ::-1])       # the fixture holds cv2's BGR array
        images[name] = rgb
        key = hashlib.sha1(rgb.tobytes()).hexdigest()
        for n in range(n_models):
            tables[n][key] = g[f"pred_{i}_{n}"]
    return F, g, names, images, [LookupModel(t) for t in tables]


def _write_inputs(F, images, src):
    os.makedirs(src, exist_ok=True)
    for name, rgb in images.items():
        F.write_png(os.path.join(src, name), rgb)                   # RGB on disk == what cv2.imread returns reversed


def _read_tree(F, dst):
    out = {}
    for sub in sorted(os.listdir(dst)):
        for name in sorted(os.listdir(os.path.join(dst, sub))):
            a = F.read_png(os.path.join(dst, sub, name), 3 if sub == "images" else 1)
            out[f"{sub}/{name}"] = a[..., ::-1] if sub == "images" else a[..., 0]      # back to cv2's BGR view
    return out


@pytest.mark.parametrize("which,fname,n_models", [("isic", "writer_isic.npz", 2), ("multi", "writer_multi.npz", 3)])
def test_product_writer_matches_reference_golden(golden_dir, tmp_path, which, fname, n_models):
    F, g, names, images, models = _fixture(golden_dir, fname, n_models)
    H, W, C = images[names[0]].shape
    src = str(tmp_path / "src")
    _write_inputs(F, images, src)
    fn = F.create_pseudo_labels_im_ISIC_2018 if which == "isic" else F.create_pseudo_labels_im_multiclass
    assert len(g["combos"]) == 8
    for tag in (str(t) for t in g["combos"]):
        bi, bo, filt = (tag[2] == "1"), (tag[6] == "1"), (tag[9] == "1")
        dst = str(tmp_path / f"dst_{which}_{tag}")
        mean = fn(models, H, W, C, src, dst, True, 0, 0, bi, bo, filt)      # positional, as the scripts call it
        got = _read_tree(F, dst)
        assert sorted(got) == sorted(str(f) for f in g[tag + "_files"]), tag
        for f, a in got.items():
            assert np.array_equal(a, g[tag + "/" + f]), (tag, f)
        assert float(mean) == float(g[tag + "_mean"][0]), tag
    # the fixture's filters do drop pairs, so the keep rules were exercised
    assert len(g["bi1_bo1_f1_files"]) < len(g["bi1_bo1_f0_files"])


def test_dilate_mask_matches_oracle():
    from inconsistencymasks_amd import functions as F
    rng = np.random.default_rng(5)
    for shape, k in (((40, 56), 5), ((33, 47), 9), ((64, 64), 35)):
        m = rng.integers(0, k, shape).astype(np.uint8)
        m[rng.random(shape) < 0.6] = 0
        assert np.array_equal(F.dilate_mask(m), O.dilate_mask_per_class(m, 3))
    b = (rng.random((3, 32, 48)) < 0.1).astype(np.uint8) * 255          # HeLa's alive / dead masks
    got = F.dilate_mask(torch.from_numpy(b).cuda()).cpu().numpy()
    for i in range(3):
        assert np.array_equal(got[i], O.dilate_mask_per_class(b[i], 3))


@pytest.mark.parametrize("ek,dk", [(5, 5), (3, 0), (0, 5)])
@pytest.mark.parametrize("which,fname,n_models", [("isic", "writer_isic.npz", 2), ("multi", "writer_multi.npz", 3)])
def test_product_writer_morphology_branch_matches_oracle(golden_dir, tmp_path, which, fname, n_models, ek, dk):
    """EK / DK > 0, incl. the DEFAULT arguments (5, 5) of the reference signature: erode -> [dilate_mask] -> dilate -> block
    (functions.py:2858-2874, 3041-3062); sizes, keep rule and mean from before the morphology."""
    F, g, names, images, models = _fixture(golden_dir, fname, n_models)
    H, W, C = images[names[0]].shape
    src, dst = str(tmp_path / "src"), str(tmp_path / "dst")
    _write_inputs(F, images, src)
    if which == "isic":
        mean = F.create_pseudo_labels_im_ISIC_2018(models, H, W, C, src, dst, True, ek, dk) if (ek, dk) != (5, 5) else \
            F.create_pseudo_labels_im_ISIC_2018(models, H, W, C, src, dst)                         # defaults must run
    else:
        mean = F.create_pseudo_labels_im_multiclass(models, H, W, C, src, dst, True, ek, dk, True, True, True)
    got = _read_tree(F, dst)
    exp, sizes = {}, []
    for i, name in enumerate(names):
        bgr = g[f"img_{i}"]
        preds = np.stack([g[f"pred_{i}_{n}"][0] for n in range(n_models)], 0)
        if which == "isic":
            r = O.im_binary(preds, 0.5, False)
            final, keep = r["final"][0], O.keep_isic(r["pred_size"], r["im_size"], True)
        else:
            r = O.im_multiclass(preds, True)
            final, keep = r["final"], r["lists_equal"]
        sizes.append(r["im_size"])
        im = r["im"]
        if ek > 0:
            im = O.erode(im, ek)
            if which == "multi":
                final = O.dilate_mask_per_class(final, 3)
        if dk > 0:
            im = O.dilate(im, dk)
        img, (mask,) = O.block(bgr, [final], im, True, True)
        if keep:
            exp["images/" + name], exp["masks/" + name] = img, mask
        exp["im/" + name] = im
    assert sorted(got) == sorted(exp)
    for f in exp:
        assert np.array_equal(got[f], exp[f]), f
    assert float(mean) == O.mean_im_size(sizes)


def test_hela_writer_morphology_branch(tmp_path):
    """create_pseudo_labels_im_hela with its DEFAULT erode / dilate kernels (functions.py:2895, 2940-2950): combined IM
    eroded then dilated, alive / dead dilated 3x3, blocking with the modified IM; table models, `>=` threshold."""
    from inconsistencymasks_amd import functions as F
    rng = np.random.default_rng(11)
    H, W, N = 32, 48, 2
    src, dst = str(tmp_path / "bf"), str(tmp_path / "out")
    os.makedirs(src)
    tables = [dict() for _ in range(N)]
    items = []
    for i in range(5):
        bf = rng.integers(1, 256, (H, W, 1)).astype(np.uint8)
        F.write_png(os.path.join(src, f"c_{i}.png"), bf)
        base = rng.random((1, H, W, 3), dtype=np.float32)
        preds = [np.clip(base + 0.25 * rng.random((1, H, W, 3), dtype=np.float32) - 0.1, 0, 1).astype(np.float32) for _ in range(N)]
        key = hashlib.sha1(np.ascontiguousarray(bf).tobytes()).hexdigest()
        for n in range(N):
            tables[n][key] = preds[n]
        items.append((f"c_{i}.png", bf, preds))
    mean = F.create_pseudo_labels_im_hela([LookupModel(t) for t in tables], H, W, 1, src, dst)
    sizes = []
    for name, bf, preds in items:
        r = O.im_binary(np.stack([p[0] for p in preds], 0), 0.5, True)        # Kb = 3, `>=`: the HeLa case
        sizes.append(r["im_size"])
        im = O.dilate(O.erode(r["im"], 5), 5)
        alive, dead = O.dilate_mask_per_class(r["final"][0], 3), O.dilate_mask_per_class(r["final"][1], 3)
        hit = im > 0
        e_bf = bf[..., 0].copy(); e_bf[hit] = 0
        alive[hit] = 0; dead[hit] = 0
        rd = lambda k: F.read_png(os.path.join(dst, k, name), 1)[..., 0]
        assert np.array_equal(rd("im"), im), name
        assert np.array_equal(rd("brightfield"), e_bf) and np.array_equal(rd("alive"), alive) and np.array_equal(rd("dead"), dead)
        assert not rd("mod_position")[hit].any()
    assert float(mean) == O.mean_im_size(sizes)
