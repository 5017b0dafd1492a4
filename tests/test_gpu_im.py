"""GPU parity of the fused IM kernels against the oracle and the golden vectors (through the C ABI)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import im_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def im():
    assert torch.cuda.is_available(), "gpu-marked test needs a GPU"
    from inconsistencymasks_amd import im as m
    return m


def _np(t):
    return t.cpu().numpy()


def test_binary_golden(im, golden_dir):
    g = np.load(os.path.join(golden_dir, "im_binary.npz"))
    for k in g["cases"]:
        preds = torch.from_numpy(g[k + "_preds"]).cuda()          # [N,1,H,W,1]
        r = im.im_binary(preds, 0.5, False, block_out=False)
        assert np.array_equal(_np(r["masks"])[0, 0], g[k + "_final"]), k
        assert np.array_equal(_np(r["im"])[0], g[k + "_im"]), k
        assert [int(r["im_size"][0, 0]), int(r["pred_size"][0, 0])] == g[k + "_sizes"].tolist(), k


def test_hela_golden(im, golden_dir):
    g = np.load(os.path.join(golden_dir, "im_hela.npz"))
    for k in g["cases"]:
        preds = torch.from_numpy(g[k + "_preds"]).cuda()          # [N,1,H,W,3]
        r = im.im_binary(preds, 0.5, True, block_out=False)
        m = _np(r["masks"])[0]
        for c, nm in enumerate(("alive", "dead", "pos")):
            assert np.array_equal(m[c], g[f"{k}_{nm}"]), (k, nm)
        assert np.array_equal(_np(r["im"])[0], g[k + "_im"]), k
        assert int(r["im_size"][0].sum()) == int(g[k + "_sizes"][0]), k


def test_multiclass_golden(im, golden_dir):
    g = np.load(os.path.join(golden_dir, "im_multiclass.npz"))
    for k in g["cases"]:
        probs = torch.from_numpy(g[k + "_preds"]).cuda()          # [N,1,H,W,K]
        r = im.im_multiclass(probs, block_out=False)
        assert np.array_equal(_np(r["final"])[0], g[k + "_final"]), k
        assert np.array_equal(_np(r["im"])[0], g[k + "_im"]), k
        assert int(r["im_size"][0]) == int(g[k + "_sizes"][0]), k
        pres = _np(r["presence"])[:, 0]
        assert int(np.all(pres == pres[0:1])) == int(g[k + "_lists_equal"][0]), k
        assert np.array_equal(pres, O.im_multiclass(g[k + "_preds"][:, 0])["presence"]), k


@pytest.mark.parametrize("shape", [(2, 3, 256, 256, 1, 3), (3, 2, 64, 48, 3, 1), (4, 2, 35, 21, 1, 3), (2, 5, 16, 16, 1, 1)])
@pytest.mark.parametrize("bi,bo", [(True, True), (False, True), (True, False)])
def test_binary_batch_vs_oracle(im, shape, bi, bo):
    n, b, h, w, kb, c = shape
    rng = np.random.default_rng(h * w + n)
    base = rng.random((1, b, h, w, kb), dtype=np.float32)
    preds = np.clip(base + (rng.random((n, b, h, w, kb), dtype=np.float32) - 0.5) * 0.3, 0, 1).astype(np.float32)
    preds[0, 0, 0, :4, 0] = [0.5, np.nan, 0.50000006, 0.49999997][:min(4, w)]
    img = rng.integers(1, 256, (b, h, w, c)).astype(np.uint8)
    ge = kb == 3
    r = im.im_binary(torch.from_numpy(preds).cuda(), 0.5, ge, torch.from_numpy(img).cuda(), bi, bo)
    for i in range(b):
        e = O.im_binary(preds[:, i], 0.5, ge)
        eimg, emasks = O.block(img[i], list(e["final"]), e["im"], bi, bo)
        assert np.array_equal(_np(r["im"])[i], e["im"])
        assert np.array_equal(_np(r["masks"])[i], np.stack(emasks))
        assert np.array_equal(_np(r["img_out"])[i], eimg)
        assert _np(r["im_size"])[i].tolist() == e["im_size_ch"].tolist()
        assert _np(r["pred_size"])[i].tolist() == e["pred_size_ch"].tolist()


@pytest.mark.parametrize("shape", [(3, 2, 256, 256, 9, 3), (2, 2, 208, 416, 35, 3), (2, 3, 13, 26, 35, 3), (3, 1, 17, 9, 4, 1)])
def test_multiclass_batch_vs_oracle(im, shape):
    n, b, h, w, k, c = shape
    rng = np.random.default_rng(k * 7 + h)
    base = rng.random((1, b, h, w, k), dtype=np.float32)
    probs = (base + 0.2 * rng.random((n, b, h, w, k), dtype=np.float32)).astype(np.float32)
    probs[:, :, : h // 3] = np.round(probs[:, :, : h // 3] * 3) / 3   # exact ties
    img = rng.integers(1, 256, (b, h, w, c)).astype(np.uint8)
    r = im.im_multiclass(torch.from_numpy(probs).cuda(), torch.from_numpy(img).cuda(), True, True)
    for i in range(b):
        e = O.im_multiclass(probs[:, i])
        eimg, (efinal,) = O.block(img[i], [e["final"]], e["im"], True, True)
        assert np.array_equal(_np(r["im"])[i], e["im"])
        assert np.array_equal(_np(r["final"])[i], efinal)
        assert np.array_equal(_np(r["img_out"])[i], eimg)
        assert int(r["im_size"][i]) == int(e["im_size"])
        assert np.array_equal(_np(r["presence"])[:, i], e["presence"])


def test_binary_properties_full_size(im):
    """Size-independent properties at the BASELINE shape (256x256x3, N=2): N=2 binary IM == XOR of votes,
    final & im disjoint, sizes add up, permutation invariance over models."""
    b = 64
    g = torch.Generator(device="cuda").manual_seed(0)
    preds = torch.rand((2, b, 256, 256, 1), device="cuda", generator=g)
    img = torch.randint(1, 256, (b, 256, 256, 3), device="cuda", dtype=torch.uint8, generator=g)
    r = im.im_binary(preds, 0.5, False, img, True, False)
    v = preds > 0.5
    xor = (v[0] ^ v[1])[..., 0]
    assert torch.equal(r["im"] > 0, xor)
    assert not torch.any((r["masks"][:, 0] > 0) & (r["im"] > 0))
    assert torch.equal(r["im_size"][:, 0], xor.sum(dim=(1, 2)))
    assert torch.equal(r["pred_size"][:, 0], (v[0] & v[1])[..., 0].sum(dim=(1, 2)))
    assert torch.equal(r["img_out"], img * (~xor)[..., None])
    r2 = im.im_binary(preds.flip(0), 0.5, False, img, True, False)
    for k in ("masks", "im", "im_size", "pred_size", "img_out"):
        assert torch.equal(r[k], r2[k])


def test_morph_and_block(im):
    rng = np.random.default_rng(3)
    m = (rng.random((3, 40, 56)) > 0.6).astype(np.uint8) * 255
    t = torch.from_numpy(m).cuda()
    for k in (3, 5):
        assert np.array_equal(_np(im.morph(t, k, "erode")), np.stack([O.erode(x, k) for x in m]))
        assert np.array_equal(_np(im.morph(t, k, "dilate")), np.stack([O.dilate(x, k) for x in m]))
    img = rng.integers(1, 256, (3, 40, 56, 3)).astype(np.uint8)
    masks = rng.integers(1, 256, (3, 2, 40, 56)).astype(np.uint8)
    ti, tm = torch.from_numpy(img).cuda(), torch.from_numpy(masks).cuda()
    im.block_apply(t, ti, tm)
    assert np.array_equal(_np(ti), img * (m == 0)[..., None])
    assert np.array_equal(_np(tm), masks * (m == 0)[:, None])


@pytest.mark.parametrize("k", [2, 3, 4, 5, 7])
def test_morph_against_scipy(im, k):
    """imk_morph against an implementation nobody here wrote (scipy.ndimage grey morphology, flat k x k footprint, constant border
    = the operator's identity; even footprints: anchor k // 2 like cv2.erode / cv2.dilate, functions.py:2858-2864).  The oracle is
    held to the same second opinion on the CPU (tests/test_cpu_second_opinion.py)."""
    from scipy import ndimage
    rng = np.random.default_rng(10 + k)
    m = (rng.random((4, 61, 83)) > rng.choice([0.3, 0.7])).astype(np.uint8) * 255
    t = torch.from_numpy(m).cuda()
    want_e = np.stack([ndimage.grey_erosion(x, size=(k, k), mode="constant", cval=255) for x in m])
    want_d = np.stack([ndimage.grey_dilation(x, size=(k, k), mode="constant", cval=0, origin=0 if k % 2 else -1) for x in m])
    assert np.array_equal(_np(im.morph(t, k, "erode")), want_e)
    assert np.array_equal(_np(im.morph(t, k, "dilate")), want_d)


@pytest.mark.parametrize("n,h,w,k", [(3, 256, 256, 9), (2, 208, 416, 35)])
def test_multiclass_properties_full_size(im, n, h, w, k):
    """Size-independent properties at the BASELINE multiclass shapes (SUIM N=3 K=9; Cityscapes N=2 K=35, 208x416):
    IM == not all argmaxes equal, final = model-0 label where consistent else 0, im_size = count, blocking."""
    b = 8
    g = torch.Generator(device="cuda").manual_seed(1)
    probs = torch.rand((n, b, h, w, k), device="cuda", generator=g)
    probs[1:, :, : h // 2] = probs[0:1, :, : h // 2]                  # top half: every model agrees
    img = torch.randint(1, 256, (b, h, w, 3), device="cuda", dtype=torch.uint8, generator=g)
    r = im.im_multiclass(probs, img, True, True)
    lab = probs.argmax(-1)
    agree = (lab == lab[0:1]).all(0)
    assert torch.equal(r["im"] > 0, ~agree)
    assert not torch.any(r["im"][:, : h // 2] > 0)
    assert torch.equal(r["final"].long(), torch.where(agree, lab[0], torch.zeros_like(lab[0])))
    assert torch.equal(r["im_size"], (~agree).sum(dim=(1, 2)))
    assert torch.equal(r["img_out"], img * agree[..., None])


def test_sharding_invariance(im):
    """a batch processed in shards (as the ranks of a multi-GPU run do) gives bit-identical outputs"""
    g = torch.Generator(device="cuda").manual_seed(2)
    preds = torch.rand((2, 37, 64, 64, 1), device="cuda", generator=g)
    img = torch.randint(0, 256, (37, 64, 64, 3), device="cuda", dtype=torch.uint8, generator=g)
    whole = im.im_binary(preds, 0.5, False, img, True, True)
    cuts = [0, 5, 19, 37]
    parts = [im.im_binary(preds[:, a:b].contiguous(), 0.5, False, img[a:b].contiguous(), True, True) for a, b in zip(cuts, cuts[1:])]
    for key in ("masks", "im", "im_size", "pred_size", "img_out"):
        assert torch.equal(whole[key], torch.cat([p[key] for p in parts], 0))


def test_randomized_shapes_vs_oracle(im):
    """60 random cases (seeded): models 2-6, batch 1-4, ragged sizes that exercise the vector / generic paths and partial
    workgroups, 1-5 sigmoid maps or 2-40 classes, thresholds incl. exact ties with the data, image channels 1 / 3 / none,
    every blocking combination; probabilities salted with the values SURVEY 8a' lists (0.5, neighbours of 0.5, 0, 1, -0.0,
    inf, denormals; NaN only for the binary chain, where the comparison defines its vote).  Bit-exact against the oracle."""
    rng = np.random.default_rng(2024)
    special = np.array([0.5, np.nextafter(np.float32(0.5), np.float32(1)), np.nextafter(np.float32(0.5), np.float32(0)),
                        0.0, 1.0, -0.0, np.inf, -np.inf, 1e-38, 0.25, 0.75], np.float32)
    for case in range(60):
        n, b = int(rng.integers(2, 7)), int(rng.integers(1, 5))
        h, w = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        c = int(rng.choice([0, 1, 3]))
        bi, bo = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        img = rng.integers(0, 256, (b, h, w, max(c, 1))).astype(np.uint8) if c else None
        imgd = torch.from_numpy(img).cuda() if c else None
        if case % 2 == 0:                                    # binary / HeLa chain
            kb = int(rng.integers(1, 6))
            ge = bool(rng.integers(0, 2))
            thr = float(rng.choice([0.5, 0.3, 0.75]))
            base = rng.random((1, b, h, w, kb), dtype=np.float32)
            p = np.clip(base + (rng.random((n, b, h, w, kb), dtype=np.float32) - 0.5) * 0.4, 0, 1).astype(np.float32)
            flat = p.reshape(-1)
            idx = rng.choice(flat.size, size=max(1, flat.size // 6), replace=False)
            flat[idx] = np.concatenate([special, [np.nan, thr]]).astype(np.float32)[rng.integers(0, len(special) + 2, idx.size)]
            r = im.im_binary(torch.from_numpy(p).cuda(), thr, ge, imgd, bi, bo)
            for i in range(b):
                e = O.im_binary(p[:, i], thr, ge)
                eimg, emasks = O.block(img[i] if c else np.zeros((h, w, 1), np.uint8), list(e["final"]), e["im"], bi, bo)
                assert np.array_equal(_np(r["im"])[i], e["im"]), case
                assert np.array_equal(_np(r["masks"])[i], np.stack(emasks)), case
                assert _np(r["im_size"])[i].tolist() == e["im_size_ch"].tolist(), case
                assert _np(r["pred_size"])[i].tolist() == e["pred_size_ch"].tolist(), case
                if c:
                    assert np.array_equal(_np(r["img_out"])[i], eimg), case
        else:                                                # multiclass chain
            k = int(rng.integers(2, 41))
            base = rng.random((1, b, h, w, k), dtype=np.float32)
            p = (base + 0.3 * rng.random((n, b, h, w, k), dtype=np.float32)).astype(np.float32)
            p[:, :, : h // 2] = np.round(p[:, :, : h // 2] * 4) / 4          # exact ties between classes
            flat = p.reshape(-1)
            idx = rng.choice(flat.size, size=max(1, flat.size // 10), replace=False)
            flat[idx] = special[rng.integers(0, len(special), idx.size)]
            r = im.im_multiclass(torch.from_numpy(p).cuda(), imgd, bi, bo)
            for i in range(b):
                e = O.im_multiclass(p[:, i])
                eimg, (efinal,) = O.block(img[i] if c else np.zeros((h, w, 1), np.uint8), [e["final"]], e["im"], bi, bo)
                assert np.array_equal(_np(r["im"])[i], e["im"]), case
                assert np.array_equal(_np(r["final"])[i], efinal), case
                assert int(r["im_size"][i]) == int(e["im_size"]), case
                assert np.array_equal(_np(r["presence"])[:, i], e["presence"]), case
                if c:
                    assert np.array_equal(_np(r["img_out"])[i], eimg), case
