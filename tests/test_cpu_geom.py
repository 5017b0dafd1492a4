"""CPU: the host C++ geometry of the HeLa position masks (csrc/imk_geom.cpp: imk_pos_contours, imk_mod_pos_size,
imk_cell_count -- the reference's functions.py:6181-6371 and the circle drawing of its pseudo-label writer, :2952-2966) against
the numpy / scipy restatement in oracle/hela_geometry.py, bit for bit, on the shapes the domain has: discs, touching and
overlapping discs, rings with islands inside their holes, salt-and-pepper noise (thousands of one-pixel blobs and holes), blobs
cut by the image edge, empty and full masks, one-pixel-wide images."""
import numpy as np
import pytest
from scipy import ndimage

from inconsistencymasks_amd import functions as F
from oracle import hela_geometry as G


def _masks(rng, n, size):
    yy, xx = np.mgrid[0:size, 0:size]
    for it in range(n):
        kind = it % 6
        m = np.zeros((size, size), np.uint8)
        if kind == 0:                                        # cells: discs, some touching, some cut by the edge
            for _ in range(rng.integers(0, 15)):
                cy, cx, rad = rng.integers(0, size), rng.integers(0, size), rng.integers(1, 9)
                m[(yy - cy) ** 2 + (xx - cx) ** 2 < rad * rad] = 255
        elif kind == 1:                                      # noise of a random density
            m = (rng.random((size, size)) < rng.random() * 0.6).astype(np.uint8) * 255
        elif kind == 2:                                      # rings, half of them with an island inside the hole
            for _ in range(rng.integers(1, 6)):
                cy, cx, rad = rng.integers(10, size - 10), rng.integers(10, size - 10), rng.integers(4, 20)
                d = (yy - cy) ** 2 + (xx - cx) ** 2
                m[(d < rad * rad) & (d >= (rad - 2) ** 2)] = 255
                if rng.random() < 0.5:
                    m[cy - 1:cy + 2, cx - 1:cx + 2] = 255
        elif kind == 3:                                      # one big blob full of holes
            m = ndimage.grey_dilation((rng.random((size, size)) < 0.5).astype(np.uint8) * 255, size=(3, 3))
        elif kind == 4:                                      # grey levels around the threshold (> 10 counts)
            m = rng.integers(0, 40, (size, size)).astype(np.uint8)
        else:
            m[:] = 255 if it % 12 == 5 else 0
        yield m


def _draw_like_the_writer(mask, max_r=8, min_r=3):
    """functions.py:2952-2966 with the restated pieces: a lone position gets distance 99, no blur"""
    out = np.zeros(mask.shape, np.uint8)
    positions = G.get_pos_contours(mask)
    for p in positions:
        md = G.get_min_dist(p, positions) if len(positions) > 1 else 99
        G._disc(out, p[0], p[1], max(min(int(md // 4), max_r), min_r), 255)
    return out


def test_native_positions_discs_and_counts_equal_the_restatement():
    rng = np.random.default_rng(5)
    for m in _masks(rng, 60, 96):
        for k in (0, 3, 5, 2):                               # 2: an even window goes through scipy's placement first
            assert F.get_pos_contours(m, erode_kernel=k) == G.get_pos_contours(m, erode_kernel=k)
        assert np.array_equal(F.mod_pos_size(m), G.mod_pos_size(m))
        assert np.array_equal(F.mod_pos_size(m, 5, 2), G.mod_pos_size(m, 5, 2))
        assert np.array_equal(F._redraw_positions(m, 8, 3, 99, 0), _draw_like_the_writer(m))
        pos = G.get_pos_contours(m)
        alive, dead = rng.integers(0, 30, m.shape).astype(np.uint8), rng.integers(0, 30, m.shape).astype(np.uint8)
        assert F.get_cell_count(pos, alive, dead) == G.get_cell_count(pos, alive, dead)
        assert F.get_cell_count(pos, alive, dead, 5) == G.get_cell_count(pos, alive, dead, 5)


def test_native_positions_on_degenerate_shapes():
    rng = np.random.default_rng(6)
    for shape in ((5, 3), (1, 9), (2, 2), (9, 1), (4, 40), (1, 1)):
        for _ in range(30):
            m = (rng.random(shape) < 0.6).astype(np.uint8) * 255
            for k in (0, 3, 5, 7):
                assert F.get_pos_contours(m, erode_kernel=k) == G.get_pos_contours(m, erode_kernel=k), (shape, k)
    # [h, w, 1] and [h, w, 3] inputs (functions.py:6186-6194) take the same route as the restatement
    m = np.zeros((32, 32, 3), np.uint8)
    m[8:20, 5:17] = (200, 30, 90)
    assert F.get_pos_contours(m) == G.get_pos_contours(m) != []
    assert F.get_pos_contours(m[..., :1]) == G.get_pos_contours(m[..., :1])
    with pytest.raises(AssertionError):
        F.get_pos_contours(np.zeros(7, np.uint8))


def test_more_positions_than_the_first_buffer_holds():
    """the wrapper asks again with a larger buffer when a mask has more than 256 positions"""
    m = np.zeros((128, 128), np.uint8)
    for y in range(0, 126, 4):
        for x in range(0, 126, 4):
            m[y:y + 2, x:x + 2] = 255                       # 32 x 32 = 1024 two-by-two blobs (polygon area 1 each)
    got = F.get_pos_contours(m, erode_kernel=0)
    assert len(got) == 1024 and got == G.get_pos_contours(m, erode_kernel=0)


def test_cell_count_rejects_what_the_reference_would_misread():
    """a window that cannot be moved inside the image (position far outside) is an error, not a wrapped slice"""
    z = np.zeros((16, 16), np.uint8)
    with pytest.raises(Exception):
        F.get_cell_count([(-9, 4)], z, z)
    assert F.get_cell_count([], z, z) == (0, 0, 0)


def test_cell_count_on_colour_and_single_channel_inputs():
    """a [h, w, 3] image is converted to grey first (functions.py:6321-6338), never read as an interleaved plane;
    [h, w, 1] is the plane; alive / dead masks of different sizes are refused"""
    rng = np.random.default_rng(9)
    alive = rng.integers(0, 40, (48, 64)).astype(np.uint8)
    dead = rng.integers(0, 40, (48, 64)).astype(np.uint8)
    pos = [(10, 10), (30, 20), (60, 40), (3, 45)]
    want = G.get_cell_count(pos, alive, dead)
    assert F.get_cell_count(pos, alive[..., None], dead[..., None]) == want
    grey3 = lambda a: np.repeat(a[..., None], 3, 2)                     # B = G = R: the grey conversion returns the plane (+- rounding)
    a3, d3 = grey3(alive), grey3(dead)
    assert F.get_cell_count(pos, a3, d3) == G.get_cell_count(pos, F._u8_plane(a3), F._u8_plane(d3))
    with pytest.raises(ValueError):
        F.get_cell_count(pos, alive, dead[:40])
