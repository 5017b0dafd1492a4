// Host-side geometry of the HeLa position masks: blob centres, disc re-drawing, cell counting.
// Restates what the reference does with OpenCV (functions.py:6181-6371: cv2.erode + cv2.findContours + cv2.moments,
// cv2.circle + cv2.blur) as plain C++: every pointer is a HOST pointer, no GPU call is made, no global state -- callers
// parallelise over images (ctypes drops the interpreter lock around the call).  A real-size HeLa driver spent 16 s of a
// 32 s run in the Python form of these three functions (profiles/r05_notes.md section 5).
#include "../../include/imk.h"

#include <cmath>
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

// no C++ exception leaves the C ABI (see imk_png.cpp): a failed allocation is IMK_EWORKSPACE
#define IMK_NOTHROW(...) try { __VA_ARGS__ } catch (const std::bad_alloc &) { return IMK_EWORKSPACE; } catch (...) { return IMK_EINVAL; }

namespace {

struct Pt { int x, y; };

// (dy, dx) of the 8 neighbours, clockwise from north
const int NB_DY[8] = {-1, -1, 0, 1, 1, 1, 0, -1};
const int NB_DX[8] = {0, 1, 1, 1, 0, -1, -1, -1};

// Moore-neighbour tracing of the outer border of the set {m[y * w + x] != 0} inside an h x w array that has a one-pixel
// background margin; starts at the top-left pixel of the set (first in raster order) "coming from" the west and follows the
// border clockwise until the first move would repeat.  Vertices in visiting order (the closed polygon through the border
// pixels' centres that cv2.findContours reports for an outer contour, up to collinear points), shifted by (ox, oy).
void trace_outer_border(const uint8_t *m, int h, int w, std::vector<Pt> &pts, int ox = 0, int oy = 0) {
    pts.clear();
    int64_t n = (int64_t)h * w, s = 0;
    while (s < n && !m[s]) ++s;
    if (s == n) return;
    int y0 = (int)(s / w), x0 = (int)(s % w);
    pts.push_back({x0 + ox, y0 + oy});
    int cy = y0, cx = x0, back = 6;
    bool have_first = false;
    int fy = 0, fx = 0, fd = 0;
    for (int64_t it = 0; it < 4 * n + 8; ++it) {
        int d = -1, ny = 0, nx = 0;
        for (int k = 1; k <= 8; ++k) {
            int dd = (back + k) & 7;
            ny = cy + NB_DY[dd]; nx = cx + NB_DX[dd];
            if (m[(int64_t)ny * w + nx]) { d = dd; break; }
        }
        if (d < 0) return;                                  // isolated pixel
        if (!have_first) { have_first = true; fy = cy; fx = cx; fd = d; }
        else if (cy == fy && cx == fx && d == fd) { pts.pop_back(); return; }
        cy = ny; cx = nx;
        back = (d + 4) & 7;
        pts.push_back({cx + ox, cy + oy});
    }
}

// m00, m10, m01 of a closed polygon by Green's theorem, accumulated in the order cv2.moments uses for a contour
void polygon_moments(const std::vector<Pt> &pts, double &m00, double &m10, double &m01) {
    double a = 0.0, sx = 0.0, sy = 0.0;
    size_t n = pts.size();
    for (size_t i = 0; i < n; ++i) {
        const Pt &p = pts[i], &q = pts[(i + 1) % n];
        int64_t cr = (int64_t)p.x * q.y - (int64_t)q.x * p.y;
        a += (double)cr;
        sx += (double)((int64_t)(p.x + q.x) * cr);
        sy += (double)((int64_t)(p.y + q.y) * cr);
    }
    m00 = a / 2.0; m10 = sx / 6.0; m01 = sy / 6.0;
}

struct Positions {
    std::vector<Pt> v;
    void add(double m00, double m10, double m01, int x_start, int y_start) {
        if (m00 == 0.0) return;
        double cx = m10 / m00 + (double)x_start - 1.0;      // back to image coordinates (cropped + padded component)
        double cy = m01 / m00 + (double)y_start - 1.0;
        v.push_back({(int)cx + 1, (int)cy + 1});
    }
};

// k x k minimum filter, window centred on the pixel, pixels outside the image do not take part (the reference's cv2.erode
// with its default border; scipy.ndimage.grey_erosion(mode="constant", cval=255) on uint8)
void erode_min(const uint8_t *in, int h, int w, int k, std::vector<uint8_t> &out) {
    int r = k / 2;
    std::vector<uint8_t> rows((size_t)h * w);
    for (int y = 0; y < h; ++y) {                           // along the rows: the edge pixels clipped, the interior as r + r
        const uint8_t *src = in + (size_t)y * w;            // shifted elementwise minima (the compiler vectorises those)
        uint8_t *dst = rows.data() + (size_t)y * w;
        for (int x = 0; x < w; ++x) {
            if (x == r && w - r > r) {
                memcpy(dst + r, src + r, (size_t)(w - 2 * r));
                for (int o = 1; o <= r; ++o)
                    for (int i = r; i < w - r; ++i) {
                        uint8_t lo = src[i - o], hi = src[i + o], m = lo < hi ? lo : hi;
                        dst[i] = m < dst[i] ? m : dst[i];
                    }
                x = w - r - 1;
                continue;
            }
            int lo = x - r < 0 ? 0 : x - r, hi = x + r >= w ? w - 1 : x + r;
            uint8_t m = 255;
            for (int i = lo; i <= hi; ++i) m = src[i] < m ? src[i] : m;
            dst[x] = m;
        }
    }
    out.resize((size_t)h * w);
    for (int y = 0; y < h; ++y) {
        int lo = y - r < 0 ? 0 : y - r, hi = y + r >= h ? h - 1 : y + r;
        uint8_t *dst = out.data() + (size_t)y * w;
        memcpy(dst, rows.data() + (size_t)lo * w, (size_t)w);
        for (int i = lo + 1; i <= hi; ++i) {
            const uint8_t *src = rows.data() + (size_t)i * w;
            for (int x = 0; x < w; ++x) dst[x] = src[x] < dst[x] ? src[x] : dst[x];
        }
    }
}

// centres of the blobs (and of their holes) of `img > 10` after the erosion, in the order the host API reports them:
// blobs in raster order of their first pixel, every blob followed by its holes in raster order of the holes' first pixel
void pos_contours(const uint8_t *img, int h, int w, int erode_kernel, Positions &pos) {
    std::vector<uint8_t> er;
    const uint8_t *a = img;
    if (erode_kernel > 1) { erode_min(img, h, w, erode_kernel, er); a = er.data(); }
    std::vector<int32_t> lab((size_t)h * w, 0);
    std::vector<int32_t> stack;
    std::vector<uint8_t> comp, bgl, shape;
    std::vector<Pt> pts;
    int32_t n_lab = 0;
    for (int sy = 0; sy < h; ++sy)
        for (int sx = 0; sx < w; ++sx) {
            size_t si = (size_t)sy * w + sx;
            if (a[si] <= 10 || lab[si]) continue;
            // one 8-connected blob: flood fill from its first pixel in raster order, bounding box on the way
            int32_t id = ++n_lab;
            int y0 = sy, y1 = sy, x0 = sx, x1 = sx;
            lab[si] = id;
            stack.clear(); stack.push_back((int32_t)si);
            while (!stack.empty()) {
                int32_t p = stack.back(); stack.pop_back();
                int py = p / w, px = p % w;
                y0 = py < y0 ? py : y0; y1 = py > y1 ? py : y1; x0 = px < x0 ? px : x0; x1 = px > x1 ? px : x1;
                for (int d = 0; d < 8; ++d) {
                    int qy = py + NB_DY[d], qx = px + NB_DX[d];
                    if (qy < 0 || qy >= h || qx < 0 || qx >= w) continue;
                    size_t qi = (size_t)qy * w + qx;
                    if (a[qi] > 10 && !lab[qi]) { lab[qi] = id; stack.push_back((int32_t)qi); }
                }
            }
            // the blob cropped to its bounding box with a one-pixel background ring
            int ch = y1 - y0 + 3, cw = x1 - x0 + 3;
            comp.assign((size_t)ch * cw, 0);
            for (int y = y0; y <= y1; ++y)
                for (int x = x0; x <= x1; ++x)
                    comp[(size_t)(y - y0 + 1) * cw + (x - x0 + 1)] = lab[(size_t)y * w + x] == id;
            trace_outer_border(comp.data(), ch, cw, pts);
            double m00, m10, m01;
            polygon_moments(pts, m00, m10, m01);
            pos.add(m00, m10, m01, x0, y0);
            // holes: 4-connected background components of the crop that do not reach the ring.  Label the outside first
            // (flood from the corner), then every remaining background pixel in raster order starts a hole.
            bgl.assign((size_t)ch * cw, 0);
            int fy0, fy1, fx0, fx1;                                         // bounding box of the last flood
            auto flood4 = [&](int fy, int fx, uint8_t mark) {
                stack.clear(); stack.push_back(fy * cw + fx); bgl[(size_t)fy * cw + fx] = mark;
                fy0 = fy1 = fy; fx0 = fx1 = fx;
                while (!stack.empty()) {
                    int32_t p = stack.back(); stack.pop_back();
                    int py = p / cw, px = p % cw;
                    fy0 = py < fy0 ? py : fy0; fy1 = py > fy1 ? py : fy1; fx0 = px < fx0 ? px : fx0; fx1 = px > fx1 ? px : fx1;
                    const int dy4[4] = {-1, 1, 0, 0}, dx4[4] = {0, 0, -1, 1};
                    for (int d = 0; d < 4; ++d) {
                        int qy = py + dy4[d], qx = px + dx4[d];
                        if (qy < 0 || qy >= ch || qx < 0 || qx >= cw) continue;
                        size_t qi = (size_t)qy * cw + qx;
                        if (!comp[qi] && !bgl[qi]) { bgl[qi] = mark; stack.push_back((int32_t)qi); }
                    }
                }
            };
            flood4(0, 0, 1);
            for (int y = 1; y < ch - 1; ++y)
                for (int x = 1; x < cw - 1; ++x) {
                    size_t i = (size_t)y * cw + x;
                    if (comp[i] || bgl[i]) continue;
                    flood4(y, x, 2);                                        // this hole: mark 2 while it is worked on
                    // hole + its ring of 4-adjacent blob pixels live inside the hole's box grown by one; one more for the
                    // tracer's background margin.  Vertices stay in the crop's coordinates (the moments are not translation
                    // invariant in floating point).
                    int oy = fy0 - 2, ox = fx0 - 2, sh = fy1 - fy0 + 5, sw = fx1 - fx0 + 5;
                    shape.assign((size_t)sh * sw, 0);
                    for (int yy = fy0 - 1; yy <= fy1 + 1; ++yy)
                        for (int xx = fx0 - 1; xx <= fx1 + 1; ++xx) {
                            size_t j = (size_t)yy * cw + xx;
                            bool in = bgl[j] == 2 ||
                                      (comp[j] && (bgl[j - 1] == 2 || bgl[j + 1] == 2 || bgl[j - cw] == 2 || bgl[j + cw] == 2));
                            shape[(size_t)(yy - oy) * sw + (xx - ox)] = in;
                        }
                    for (int yy = fy0; yy <= fy1; ++yy)
                        for (int xx = fx0; xx <= fx1; ++xx)
                            if (bgl[(size_t)yy * cw + xx] == 2) bgl[(size_t)yy * cw + xx] = 3;     // done
                    trace_outer_border(shape.data(), sh, sw, pts, ox, oy);
                    polygon_moments(pts, m00, m10, m01);
                    pos.add(m00, m10, m01, x0, y0);
                }
        }
}

// cv2.circle(img, (cx, cy), r, value, -1): OpenCV's integer midpoint rasteriser, clipped to the image
void disc(uint8_t *img, int h, int w, int cx, int cy, int r, uint8_t value) {
    auto span = [&](int y, int x0, int x1) {
        if (y < 0 || y >= h) return;
        x0 = x0 < 0 ? 0 : x0; x1 = x1 > w - 1 ? w - 1 : x1;
        if (x0 <= x1) memset(img + (size_t)y * w + x0, value, (size_t)(x1 - x0 + 1));
    };
    int err = 0, dx = r, dy = 0, plus = 1, minus = 2 * r - 1;
    while (dx >= dy) {
        span(cy - dy, cx - dx, cx + dx); span(cy + dy, cx - dx, cx + dx);
        span(cy - dx, cx - dy, cx + dy); span(cy + dx, cx - dy, cx + dy);
        ++dy; err += plus; plus += 2;
        if (err > 0) { err -= minus; --dx; minus -= 2; }
    }
}

}  // namespace

// functions.py:6181-6218 get_pos_contours on one h x w uint8 mask.  Writes at most `cap` (x, y) pairs to xy, returns the
// number of positions found (the caller retries with a larger buffer when it exceeds cap).
extern "C" IMK_API int imk_pos_contours(const uint8_t *img, int h, int w, int erode_kernel, int32_t *xy, int cap) {
    if (!img || h <= 0 || w <= 0 || (int64_t)h * w > INT32_MAX || cap < 0 || (cap > 0 && !xy)) return IMK_EINVAL;
    if (erode_kernel > 1 && !(erode_kernel & 1)) return IMK_EUNSUPPORTED;    // even windows: the caller erodes
    IMK_NOTHROW(
    Positions pos;
    pos_contours(img, h, w, erode_kernel, pos);
    int n = (int)pos.v.size();
    for (int i = 0; i < n && i < cap; ++i) { xy[2 * i] = pos.v[i].x; xy[2 * i + 1] = pos.v[i].y; }
    return n;
    )
}

// functions.py:6255-6292 mod_pos_size: every blob re-drawn as a filled circle of radius clamp(min_dist // 4, min_r, max_r),
// then (blur2 != 0) cv2.blur(out, (2, 2)) and out[out < 254] = 0: a pixel survives iff its 2 x 2 window -- itself, left,
// upper, upper-left neighbour, BORDER_REFLECT_101 at the edge -- is fully set.  min_dist = the smallest non-zero distance to
// another position, 0 when there is none; `lone_dist` replaces it when the mask holds exactly one position (the pseudo-label
// writer, functions.py:2952-2966, draws a lone cell with distance 99 and does not blur).  out: h x w uint8 in {0, 255}.
extern "C" IMK_API int imk_mod_pos_size(const uint8_t *img, int h, int w, int max_r, int min_r, int lone_dist, int blur2,
                                        uint8_t *out) {
    if (!img || !out || h <= 1 || w <= 1 || (int64_t)h * w > INT32_MAX) return IMK_EINVAL;
    IMK_NOTHROW(
    Positions pos;
    pos_contours(img, h, w, 3, pos);
    std::vector<uint8_t> scratch;
    uint8_t *drawn = out;
    if (blur2) { scratch.assign((size_t)h * w, 0); drawn = scratch.data(); }
    else memset(out, 0, (size_t)h * w);
    size_t n = pos.v.size();
    for (size_t i = 0; i < n; ++i) {
        double best = 0.0;
        for (size_t j = 0; j < n; ++j) {
            int64_t dx = pos.v[j].x - pos.v[i].x, dy = pos.v[j].y - pos.v[i].y;
            if (dx == 0 && dy == 0) continue;
            double d = std::sqrt((double)(dx * dx + dy * dy));
            if (best == 0.0 || d < best) best = d;
        }
        if (n == 1) best = (double)lone_dist;
        int r = (int)std::floor(best / 4.0);
        r = r < max_r ? r : max_r;
        r = r > min_r ? r : min_r;
        disc(drawn, h, w, pos.v[i].x, pos.v[i].y, r, 255);
    }
    if (!blur2) return IMK_OK;
    for (int y = 0; y < h; ++y) {
        int yu = y == 0 ? 1 : y - 1;
        for (int x = 0; x < w; ++x) {
            int xl = x == 0 ? 1 : x - 1;
            bool keep = drawn[(size_t)y * w + x] && drawn[(size_t)y * w + xl] && drawn[(size_t)yu * w + x] && drawn[(size_t)yu * w + xl];
            out[(size_t)y * w + x] = keep ? 255 : 0;
        }
    }
    return IMK_OK;
    )
}

// functions.py:6298-6371 get_cell_count: at every position, the class with more pixels > 10 in the 2m x 2m window around it
// (moved inside the image the way the reference moves it).  counts = {alive, dead, unclear}.
extern "C" IMK_API int imk_cell_count(const int32_t *xy, int n, const uint8_t *alive, const uint8_t *dead, int h, int w,
                                      int measuring_range, int32_t counts[3]) {
    const int64_t m = measuring_range;          // 64-bit window arithmetic: positions and the range are the caller's, any int32
    if (n < 0 || (n > 0 && !xy) || !alive || !dead || !counts || m <= 0 || h < 2 * m || w < 2 * m) return IMK_EINVAL;
    counts[0] = counts[1] = counts[2] = 0;
    for (int i = 0; i < n; ++i) {
        int64_t x = xy[2 * i], y = xy[2 * i + 1];
        if (x - m <= 0) x += m;
        if (x + m > w) x = w - m;
        if (y - m < 0) y += m;
        if (y + m > h) y = h - m;
        if (x - m < 0 || y - m < 0 || x + m > w || y + m > h) return IMK_EINVAL;     // a position outside the image
        int sa = 0, sd = 0;
        for (int64_t yy = y - m; yy < y + m; ++yy)
            for (int64_t xx = x - m; xx < x + m; ++xx) {
                sa += alive[(size_t)yy * w + xx] > 10;
                sd += dead[(size_t)yy * w + xx] > 10;
            }
        counts[sa > sd ? 0 : (sd > sa ? 1 : 2)]++;
    }
    return IMK_OK;
}
