"""CPU restatement of the HeLa position-mask geometry.  TEST INFRASTRUCTURE ONLY.

The checker for the host C++ in inconsistencymasks_amd/csrc/imk_geom.cpp (imk_pos_contours, imk_mod_pos_size, imk_cell_count):
imported only by tests/ -- never by the product path, which calls libimk.so and fails loudly without it.  Until round 5 this
text WAS the product's host path (inconsistencymasks_amd/functions.py); it moved here unchanged when the native form took over.

Parity status: UNPINNED for the OpenCV calls (cv2.erode / findContours / moments / circle / blur are not importable in the build
container, so the reference cannot run them here): restated from the published algorithms, second opinions from scipy in
tests/test_cpu_second_opinion.py (centre of mass, Euclidean disc areas).  get_min_dist and get_cell_count are plain numpy in the
reference and are pinned by tests/golden (make_golden.py imports the reference).

Reference lines followed (relative to /root/reference): functions.py:6181-6218 get_pos_contours, 6221-6252 get_min_dist,
6255-6292 mod_pos_size, 6298-6371 get_cell_count."""
import numpy as np


def _trace_outer_border(comp):
    """Outer border of one 8-connected component (boolean array, padded by one background pixel on every side), followed
    clockwise from its top-left pixel with Moore-neighbour tracing -- the closed polygon through the border pixels' centres
    that cv2.findContours reports for an outer contour (CHAIN_APPROX_SIMPLE only drops collinear points, which changes
    neither the area nor the moments).  Returns the vertices as (x, y) in visiting order."""
    ys, xs = np.nonzero(comp)
    y0 = int(ys.min())
    x0 = int(xs[ys == y0].min())
    nbr = [(-1, 0), (-1, 1), (0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1)]     # (dy, dx), clockwise from north
    pts = [(x0, y0)]
    cy, cx, back = y0, x0, 6          # we "came from" the west: the pixel left of the start is background
    first_move = None
    for _ in range(4 * comp.size + 8):
        for k in range(1, 9):          # first foreground pixel clockwise after the backtrack direction
            d = (back + k) % 8
            ny, nx = cy + nbr[d][0], cx + nbr[d][1]
            if comp[ny, nx]:
                break
        else:
            return pts                 # isolated pixel
        if first_move is None:
            first_move = (cy, cx, d)
        elif (cy, cx, d) == first_move:
            return pts[:-1]            # back at the start, about to repeat the first move: the polygon is closed
        cy, cx = ny, nx
        back = (d + 4) % 8             # direction pointing back to where we came from
        pts.append((cx, cy))
    return pts


def _polygon_moments(pts):
    """m00, m10, m01 of a closed polygon (Green's theorem), as cv2.moments computes them for a contour"""
    a = m10 = m01 = 0.0
    n = len(pts)
    for i in range(n):
        x0, y0 = pts[i]
        x1, y1 = pts[(i + 1) % n]
        cr = x0 * y1 - x1 * y0
        a += cr
        m10 += (x0 + x1) * cr
        m01 += (y0 + y1) * cr
    return a / 2.0, m10 / 6.0, m01 / 6.0


def get_pos_contours(img, erode_kernel=3):
    """functions.py:6181-6218: centres (x, y) of the blobs of a position mask.  The reference erodes, thresholds at 10,
    takes cv2.findContours + cv2.moments of every contour and reports (int(m10 / m00) + 1, int(m01 / m00) + 1), skipping
    contours whose polygon area m00 is zero (single pixels, one-pixel-wide lines).  Restated here with connected components
    (scipy), Moore-neighbour border tracing and Green's-theorem polygon moments.  RETR_TREE also reports every HOLE of a blob
    as a contour of its own -- the blob's pixels that have a pixel of the hole in their 4-neighbourhood (Suzuki-Abe border
    following with 8-connected blobs and 4-connected holes) -- and the reference's loop adds a position for each: here a hole is a
    4-connected background component that does not reach the blob's bounding box, and its contour is traced as the outer border of
    (hole + that ring of blob pixels).  Unpinned: OpenCV is not available to the reference in this environment."""
    from scipy import ndimage
    a = np.asarray(img)
    assert a.ndim in (2, 3), "Invalid image dimensions."
    if a.ndim == 3:
        a = a[..., 0] if a.shape[2] == 1 else (0.114 * a[..., 0] + 0.587 * a[..., 1] + 0.299 * a[..., 2]).astype(np.uint8)
    if erode_kernel > 0:
        a = ndimage.grey_erosion(a.astype(np.uint8), size=(erode_kernel, erode_kernel), mode="constant", cval=255)
    lab, n = ndimage.label(a > 10, structure=np.ones((3, 3)))
    cross = ndimage.generate_binary_structure(2, 1)
    pos = []
    for sl, idx in zip(ndimage.find_objects(lab), range(1, n + 1)):
        comp = np.pad(lab[sl] == idx, 1)
        pts = _trace_outer_border(comp)
        m00, m10, m01 = _polygon_moments(pts)
        if m00 != 0:
            cx = m10 / m00 + sl[1].start - 1        # back to image coordinates (the component was cropped and padded)
            cy = m01 / m00 + sl[0].start - 1
            pos.append((int(cx) + 1, int(cy) + 1))
        bg, nb = ndimage.label(~comp, structure=cross)
        for h in range(1, nb + 1):
            if h == bg[0, 0]:                        # the outside (the padding ring belongs to it)
                continue
            hole = bg == h
            ring = ndimage.binary_dilation(hole, structure=cross) & comp
            m00, m10, m01 = _polygon_moments(_trace_outer_border(hole | ring))
            if m00 != 0:
                pos.append((int(m10 / m00 + sl[1].start - 1) + 1, int(m01 / m00 + sl[0].start - 1) + 1))
    return pos


def get_min_dist(xy, positions):
    """functions.py:6221-6252."""
    d = np.linalg.norm(np.array(positions) - np.array(xy), axis=1)
    d = d[d > 0]
    return 0 if d.size == 0 else float(np.min(d))


def _disc(img, cx, cy, r, value):
    """cv2.circle(img, (cx, cy), r, value, -1): the filled circle of OpenCV's integer midpoint rasteriser (restated from
    the published algorithm of imgproc's drawing code: horizontal spans cy +- dy: [cx - dx, cx + dx] and cy +- dx:
    [cx - dy, cx + dy] while dx >= dy, error update err += 2 dy + 1, step dx inwards when err > 0), clipped to the image.
    Not the Euclidean disc: r = 3 gives rows of 1, 5, 5, 7, 5, 5, 1 pixels.  Unpinned (needs OpenCV in the reference)."""
    h, w = img.shape[:2]

    def span(y, x0, x1):
        if 0 <= y < h:
            x0, x1 = max(x0, 0), min(x1, w - 1)
            if x0 <= x1:
                img[y, x0:x1 + 1] = value

    err, dx, dy, plus, minus = 0, int(r), 0, 1, 2 * int(r) - 1
    while dx >= dy:
        span(cy - dy, cx - dx, cx + dx); span(cy + dy, cx - dx, cx + dx)
        span(cy - dx, cx - dy, cx + dy); span(cy + dx, cx - dy, cx + dy)
        dy += 1
        err += plus
        plus += 2
        if err > 0:
            err -= minus
            dx -= 1
            minus -= 2


def mod_pos_size(gray_img, max_pos_circle_size=8, min_pos_circle_size=3):
    """functions.py:6255-6292: redraw every position blob as a filled circle of radius clamp(min_dist // 4, 3, 8), then
    `cv2.blur(out, (2, 2))` and `out[out < 254] = 0`: a pixel survives iff its whole 2x2 window (itself, left, upper,
    upper-left neighbour; BORDER_REFLECT_101 at the image edge) is set."""
    positions = get_pos_contours(gray_img)
    out = np.zeros(gray_img.shape, np.uint8)
    for p in positions:
        r = int(get_min_dist(p, positions) // 4)
        r = max(min(r, max_pos_circle_size), min_pos_circle_size)
        _disc(out, p[0], p[1], r, 255)
    on = np.pad(out > 0, ((1, 0), (1, 0)), mode="reflect")            # row / column -1 -> row / column 1
    keep = on[1:, 1:] & on[1:, :-1] & on[:-1, 1:] & on[:-1, :-1]
    return np.where(keep, 255, 0).astype(np.uint8)


def get_cell_count(positions, img_alive, img_dead, measuring_range=3):
    """functions.py:6298-6371."""
    a = (np.asarray(img_alive) > 10).astype(np.int64) * 255
    dd = (np.asarray(img_dead) > 10).astype(np.int64) * 255
    ih, iw = a.shape[:2]
    alive = dead = unclear = 0
    m = measuring_range
    for x, y in positions:
        if x - m <= 0:
            x += m
        if x + m > iw:
            x = iw - m
        if y - m < 0:
            y += m
        if y + m > ih:
            y = ih - m
        sa, sd = a[y - m:y + m, x - m:x + m].sum(), dd[y - m:y + m, x - m:x + m].sum()
        alive += sa > sd
        dead += sd > sa
        unclear += sa == sd
    return int(alive), int(dead), int(unclear)
