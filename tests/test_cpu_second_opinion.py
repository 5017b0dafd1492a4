"""A second, independent implementation for every OpenCV piece this repository restates.

cv2 is not installed, so the morphology (functions.py:2858-2864, 3075-3100), the Gaussian blur of the augmentation
(functions.py:1495-1501) and the contour centres of the HeLa position masks (functions.py:6181-6252) are restatements in
oracle/ and inconsistencymasks_amd/functions.py, checked so far only against each other.  scipy.ndimage IS in the image and was
written by other people from other sources: where it offers the same operator, the restatement must agree with it bit for bit.
Not a pin on OpenCV (that needs cv2: README "Pins this image cannot produce") -- it removes "the oracle agrees with itself".
CPU only."""
import numpy as np
import pytest
from scipy import ndimage

from oracle import aug_oracle as A
from oracle import hela_geometry as G
from oracle import im_oracle as O


def _masks(rng, n, h, w):
    for _ in range(n):
        m = (rng.random((h, w)) < rng.choice([0.05, 0.3, 0.6, 0.9])).astype(np.uint8) * 255
        if rng.random() < 0.5:      # blobs instead of salt and pepper
            m = (ndimage.uniform_filter(m.astype(np.float32), 5) > 110).astype(np.uint8) * 255
        yield m


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 7])
def test_erode_dilate_against_scipy_grey_morphology(k):
    """k x k all-ones structuring element, anchor k // 2, pixels outside the image = the operator's identity (what cv2.erode /
    cv2.dilate do with their default border: functions.py:2858-2864).  scipy: grey_erosion / grey_dilation with a flat k x k
    footprint and a constant border of 255 / 0; an even footprint's centre sits at k // 2 in scipy too (origin 0) for erosion, and
    mirrored for dilation (scipy reflects the footprint there) -- hence origin -1 for even k."""
    rng = np.random.default_rng(k)
    for m in _masks(rng, 12, 37, 53):
        want_e = ndimage.grey_erosion(m, size=(k, k), mode="constant", cval=255)
        want_d = ndimage.grey_dilation(m, size=(k, k), mode="constant", cval=0, origin=0 if k % 2 else -1)
        assert np.array_equal(O.erode(m, k), want_e)
        assert np.array_equal(O.dilate(m, k), want_d)


def test_erode_then_dilate_is_an_opening_and_never_grows():
    rng = np.random.default_rng(0)
    for m in _masks(rng, 8, 40, 40):
        for k in (3, 5):
            o = O.dilate(O.erode(m, k), k)
            assert (o <= m).all()                                    # an opening is anti-extensive ...
            assert np.array_equal(O.dilate(O.erode(o, k), k), o)      # ... and idempotent


def test_dilate_mask_per_class_against_scipy_label_maps():
    """functions.py:3075-3100 (dilate_mask): every class dilated on its own, higher ids overwrite lower ones = the 3 x 3 grey-level
    dilation of the id map; scipy's binary_dilation per class, composed in ascending order, as the third opinion"""
    rng = np.random.default_rng(5)
    for _ in range(10):
        m = rng.integers(0, 6, (33, 41)).astype(np.uint8) * (rng.random((33, 41)) < 0.3)
        want = np.zeros_like(m)
        for u in np.unique(m):
            if u:
                want[ndimage.binary_dilation(m == u, structure=np.ones((3, 3), bool))] = u
        assert np.array_equal(O.dilate_mask_per_class(m, 3), want)


@pytest.mark.parametrize("k", [3, 5, 7])
def test_gaussian_blur_against_scipy_correlate(k):
    """cv2.GaussianBlur(img, (k, k), 0) on uint8 (functions.py:1495-1501) as restated in oracle/aug_oracle.py: OpenCV's fixed kernels
    for sigma 0, BORDER_REFLECT_101, exact integer accumulation, one rounding half up at the end.  scipy: two correlate1d passes
    with mode 'mirror' (= reflect without repeating the edge pixel = REFLECT_101) on int64, the same final rounding."""
    taps = {3: [1, 2, 1], 5: [1, 4, 6, 4, 1], 7: [2, 7, 14, 18, 14, 7, 2]}[k]
    scale = {3: 4, 5: 16, 7: 64}[k]
    rng = np.random.default_rng(k)
    for shape in ((31, 45, 3), (16, 16, 1), (9, 64, 3)):
        a = rng.integers(0, 256, shape).astype(np.uint8)
        acc = ndimage.correlate1d(ndimage.correlate1d(a.astype(np.int64), taps, axis=0, mode="mirror"), taps, axis=1, mode="mirror")
        want = ((2 * acc + scale * scale) // (2 * scale * scale)).astype(np.uint8)      # round half up of acc / scale^2
        assert np.array_equal(A.gaussian_blur(a, k), want)
    # the kernels are OpenCV's small_gaussian_tab, normalised: the oracle's x/64 table is the same numbers
    assert [int(v) * scale // 64 for v in A._GAUSS[k]] == taps or [int(v) for v in A._GAUSS[k]] == [t * 64 // scale for t in taps]


def test_convert_scale_abs_against_float64_definition():
    """cv2.convertScaleAbs: saturate_cast<uchar>(|alpha x + beta|), rounding to nearest even (cvRound) -- functions.py:2820-2824;
    the restatement works in float32, the definition in float64: equal wherever the float32 product is exact enough to land on
    the same side of .5 (asserted for the parameter ranges the augmentation draws: alpha in [0.5, 1.5], beta in [-50, 50])"""
    rng = np.random.default_rng(3)
    x = np.arange(256, dtype=np.uint8)
    for _ in range(200):
        alpha, beta = float(rng.uniform(0.5, 1.5)), float(rng.integers(-50, 51))
        want = np.minimum(np.rint(np.abs(x.astype(np.float64) * np.float64(np.float32(alpha)) + beta)), 255).astype(np.uint8)
        got = A.convert_scale_abs(x, alpha, beta)
        assert np.abs(got.astype(int) - want.astype(int)).max() <= 1
        assert (got != want).mean() <= 0.01        # float32 vs float64 can differ on exact .5 ties only


def _blobs(rng, n):
    """hole-free, centrally symmetric blobs (axis-aligned rectangles and ellipses), one per image"""
    for i in range(n):
        h, w = 48, 64
        yy, xx = np.mgrid[0:h, 0:w]
        cy, cx = int(rng.integers(14, h - 14)), int(rng.integers(14, w - 14))
        ry, rx = int(rng.integers(3, 12)), int(rng.integers(3, 12))
        m = (np.abs(yy - cy) <= ry) & (np.abs(xx - cx) <= rx) if i % 2 else (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0)
        yield m.astype(np.uint8) * 255, (cx, cy)


def test_contour_centres_against_scipy_centre_of_mass_and_picks_theorem():
    """functions.get_pos_contours (functions.py:6181-6218: cv2.findContours + cv2.moments -> int(m10 / m00) + 1, int(m01 / m00) + 1):
    for a centrally symmetric, hole-free blob the centroid of the contour polygon IS the blob's centre of mass (scipy:
    label + center_of_mass), and the polygon's area obeys Pick's theorem -- lattice polygon through the border pixels' centres:
    area = pixels - border pixels / 2 - 1 -- which checks the border tracing and the Green's-theorem moments without OpenCV."""
    from inconsistencymasks_amd import functions as F
    rng = np.random.default_rng(11)
    for m, (cx, cy) in _blobs(rng, 40):
        lab, n = ndimage.label(m > 10, structure=np.ones((3, 3)))
        assert n == 1
        com_y, com_x = ndimage.center_of_mass(m > 10, lab, 1)
        assert abs(com_x - cx) < 1e-9 and abs(com_y - cy) < 1e-9
        assert F.get_pos_contours(m, erode_kernel=0) == [(int(com_x) + 1, int(com_y) + 1)]
        comp = np.pad(lab == 1, 1)
        pts = G._trace_outer_border(comp)
        m00, m10, m01 = G._polygon_moments(pts)
        interior = ndimage.binary_erosion(comp, structure=ndimage.generate_binary_structure(2, 1))     # pixels whose 4 neighbours are all blob
        border = int(comp.sum() - interior.sum())
        assert len(set(pts)) == border                           # the trace visits every border pixel of a convex blob once
        assert abs(abs(m00) - (comp.sum() - border / 2 - 1)) < 1e-9
    # two blobs: scipy's labelled centres, in any order
    two = np.zeros((48, 64), np.uint8)
    two[5:12, 6:15] = 255
    two[30:41, 40:47] = 255
    lab, n = ndimage.label(two > 10, structure=np.ones((3, 3)))
    coms = sorted((int(x) + 1, int(y) + 1) for y, x in ndimage.center_of_mass(two > 10, lab, [1, 2]))
    assert sorted(F.get_pos_contours(two, erode_kernel=0)) == coms


def test_position_disc_rasteriser_against_the_distance_definition():
    """cv2.circle(img, centre, r, colour, -1) (functions.py:2960-2965) restated as OpenCV's midpoint rasteriser: a filled disc must
    contain every pixel within r - 0.5 of the centre, none beyond r + 0.5, be symmetric under the 8 reflections, and its area must
    be within the lattice-point bounds of a disc of that radius"""
    from inconsistencymasks_amd import functions as F
    draw = G._disc
    for r in range(1, 10):
        img = np.zeros((41, 41), np.uint8)
        draw(img, 20, 20, r, 255)
        yy, xx = np.mgrid[0:41, 0:41]
        d = np.hypot(yy - 20, xx - 20)
        assert (img[d <= r - 0.5] == 255).all() and (img[d >= r + 0.75] == 0).all()
        for t in (img.T, img[::-1], img[:, ::-1]):
            assert np.array_equal(t, img)
