// Noisy-Student augmentation of (image, mask) pairs on the GPU (IM+ / AIM+ drivers).
// One fused streaming kernel per batch: flips / 90-degree rotations (image and mask), convertScaleAbs brightness,
// Gaussian blur 3/5/7, uniform integer noise, clip -- the chain of augment_image_and_mask (functions.py:2779-2826),
// add_noise_and_blur (:1481-1506) and add_noise (:1463-1478) of the reference, which runs it per image in OpenCV.
// HBM-bound: the image is read once (blur taps hit L1/L2) and written once.
//   * blur: OpenCV's fixed small Gaussian kernels for sigma = 0 ([1 2 1]/4, [1 4 6 4 1]/16, [2 7 14 18 14 7 2]/64),
//     BORDER_REFLECT_101, exact integer accumulation, round half up (what its fixed-point u8 path produces);
//   * brightness: saturate_cast<uchar>(|a*x + b|) with round-to-nearest-even;
//   * noise: uniform integer in [-m, m) per element from a counter-based hash of (seed, element index) -- the
//     reference draws from numpy's unseeded global stream, which nothing can reproduce.
#include "imk_common.h"

namespace {

__device__ __forceinline__ uint32_t hash32(uint32_t x) {   // lowbias32
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

__global__ __launch_bounds__(256) void augment_kernel(const uint8_t *__restrict__ img, const uint8_t *__restrict__ mask,
                                                      int H, int W, int C, int Cm, const imk_aug_params *__restrict__ prm,
                                                      uint8_t *__restrict__ img_out, uint8_t *__restrict__ mask_out) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) return;
    const imk_aug_params q = prm[b];
    const bool swap = (q.rot == 1 || q.rot == 3);          // 90-degree turns: only square images reach here (host checks)
    const int Ho = swap ? W : H, Wo = swap ? H : W;
    const int yo = p / Wo, xo = p - yo * Wo;
    // output pixel -> coordinates in the flipped image (inverse rotation)
    int yf, xf;
    switch (q.rot) {
        case 1: yf = H - 1 - xo; xf = yo; break;            // ROTATE_90_CLOCKWISE
        case 2: yf = H - 1 - yo; xf = W - 1 - xo; break;    // ROTATE_180
        case 3: yf = xo; xf = W - 1 - yo; break;            // ROTATE_90_COUNTERCLOCKWISE
        default: yf = yo; xf = xo;
    }
    (void)Ho;
    // The blur acts on the transformed image: its taps are neighbours in OUTPUT space.  Map every tap back.
    const int k = q.blur_k;                                  // 0/1 = none, 3, 5, 7
    const int r = k > 1 ? k / 2 : 0;
    int wgt[7] = {64, 0, 0, 0, 0, 0, 0};
    if (k == 3) { wgt[0] = 16; wgt[1] = 32; wgt[2] = 16; }
    else if (k == 5) { wgt[0] = 4; wgt[1] = 16; wgt[2] = 24; wgt[3] = 16; wgt[4] = 4; }
    else if (k == 7) { wgt[0] = 2; wgt[1] = 7; wgt[2] = 14; wgt[3] = 18; wgt[4] = 14; wgt[5] = 7; wgt[6] = 2; }
    auto src_index = [&](int y_out, int x_out) -> size_t {
        int ys, xs;
        switch (q.rot) {
            case 1: ys = H - 1 - x_out; xs = y_out; break;
            case 2: ys = H - 1 - y_out; xs = W - 1 - x_out; break;
            case 3: ys = x_out; xs = W - 1 - y_out; break;
            default: ys = y_out; xs = x_out;
        }
        if (q.flip_h) xs = W - 1 - xs;
        if (q.flip_v) ys = H - 1 - ys;
        return ((size_t)b * H + ys) * W + xs;
    };
    (void)yf; (void)xf;
    auto bright = [&](int v) -> int {
        if (!q.bright_on) return v;
        const int o = __float2int_rn(fabsf((float)v * q.alpha + q.beta));
        return o > 255 ? 255 : o;
    };
    for (int c = 0; c < C; ++c) {
        int val;
        if (r == 0) {
            val = bright(img[src_index(yo, xo) * C + c]);
        } else {
            int acc = 0;
            for (int dy = -r; dy <= r; ++dy) {
                const int yy = reflect101(yo + dy, Ho);
                int row = 0;
                for (int dx = -r; dx <= r; ++dx) {
                    const int xx = reflect101(xo + dx, Wo);
                    row += wgt[dx + r] * bright(img[src_index(yy, xx) * C + c]);
                }
                acc += wgt[dy + r] * row;
            }
            val = (acc + 2048) >> 12;                        // weights sum to 64 per axis
        }
        if (q.noise_max > 0) {
            const uint32_t e = (uint32_t)(((size_t)p * C + c));
            const uint32_t h = hash32(hash32(q.seed ^ 0x9e3779b9u) + e);
            const int noise = (int)(((uint64_t)h * (uint32_t)(2 * q.noise_max)) >> 32) - q.noise_max;
            val = min(255, max(0, val + noise));
        }
        img_out[((size_t)b * Ho * Wo + p) * C + c] = (uint8_t)val;
    }
    if (mask)
        for (int c = 0; c < Cm; ++c) mask_out[((size_t)b * Ho * Wo + p) * Cm + c] = mask[src_index(yo, xo) * Cm + c];
}

// ---- epoch assembly of a device-resident training set ------------------------------------------------------------------
// dst row r = src row idx[r]: the shuffle + batch of the reference's tf.data pipeline (functions.py:207-209) over a set that
// already lives in HBM, with the mask normalisation of its parsers folded in (parse_image_ISIC_2018: mask / 255,
// functions.py:975; parse_image_hela: / 255, position x Position_weight, functions.py:1001-1011).
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint8_t *__restrict__ src, const int64_t *__restrict__ idx,
                                                          uint8_t *__restrict__ dst, int64_t row_bytes, int vec) {
    const int64_t r = blockIdx.y;
    const uint8_t *s = src + idx[r] * row_bytes;
    uint8_t *d = dst + r * row_bytes;
    if (vec) {
        const int64_t n16 = row_bytes / 16;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256)
            reinterpret_cast<uint4 *>(d)[i] = reinterpret_cast<const uint4 *>(s)[i];
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < row_bytes; i += (int64_t)gridDim.x * 256) d[i] = s[i];
    }
}
// planar [n_src][planes][hw] -> interleaved [n][hw][planes], v -> (div255 ? v / 255 : v) * mul[plane]
__global__ __launch_bounds__(256) void gather_planes_kernel(const uint8_t *__restrict__ src, const int64_t *__restrict__ idx,
                                                            uint8_t *__restrict__ dst, int planes, int64_t hw, int div255,
                                                            const uint8_t *__restrict__ mul) {
    const int64_t r = blockIdx.y;
    const uint8_t *s = src + idx[r] * planes * hw;
    uint8_t *d = dst + r * planes * hw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)gridDim.x * 256)
        for (int k = 0; k < planes; ++k) {
            unsigned v = s[(int64_t)k * hw + i];
            if (div255) v /= 255u;
            if (mul) v *= mul[k];
            d[i * planes + k] = (uint8_t)v;
        }
}

}  // namespace

extern "C" int imk_gather_pairs(const uint8_t *img, int64_t row_img, const uint8_t *mask, int planes, int64_t hw, int div255,
                                const uint8_t *mul, const int64_t *idx, int64_t n, uint8_t *img_out, uint8_t *mask_out,
                                void *stream_) {
    IMK_CHECK_ARG(idx && n > 0 && (img || mask));
    IMK_CHECK_ARG(!img || (img_out && row_img > 0 && img != img_out));
    IMK_CHECK_ARG(!mask || (mask_out && planes > 0 && hw > 0 && mask != mask_out));
    if (n > 65535) return IMK_EUNSUPPORTED;      // rows ride on gridDim.y
    hipStream_t stream = (hipStream_t)stream_;
    auto a16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (img) {
        const int vec = row_img % 16 == 0 && a16(img) && a16(img_out);
        const int64_t work = vec ? row_img / 16 : row_img;
        const int bx = (int)(work / 256 < 1 ? 1 : (work / 256 > 64 ? 64 : work / 256));
        imk_klaunch(gather_rows_kernel, dim3(dim3(bx, (unsigned)n)), dim3(256), 0, stream, img, idx, img_out, row_img, vec);
        IMK_LAUNCH_CHECK();
    }
    if (mask) {
        if (planes == 1 && !div255 && !mul) {
            const int64_t row = hw;
            const int vec = row % 16 == 0 && a16(mask) && a16(mask_out);
            const int64_t work = vec ? row / 16 : row;
            const int bx = (int)(work / 256 < 1 ? 1 : (work / 256 > 64 ? 64 : work / 256));
            imk_klaunch(gather_rows_kernel, dim3(dim3(bx, (unsigned)n)), dim3(256), 0, stream, mask, idx, mask_out, row, vec);
        } else {
            const int bx = (int)(hw / 256 < 1 ? 1 : (hw / 256 > 64 ? 64 : hw / 256));
            imk_klaunch(gather_planes_kernel, dim3(dim3(bx, (unsigned)n)), dim3(256), 0, stream, mask, idx, mask_out, planes, hw, div255, mul);
        }
        IMK_LAUNCH_CHECK();
    }
    return IMK_OK;
}

extern "C" int imk_augment(const uint8_t *img, const uint8_t *mask, int batch, int h, int w, int c, int cm,
                           const imk_aug_params *params, uint8_t *img_out, uint8_t *mask_out, int any_quarter_turn,
                           void *stream_) {
    IMK_CHECK_ARG(img && img_out && params && batch > 0 && h > 0 && w > 0 && c > 0);
    IMK_CHECK_ARG(!mask || (mask_out && cm > 0));
    IMK_CHECK_ARG(img != img_out && (!mask || mask != mask_out));
    if (any_quarter_turn && h != w) return IMK_EUNSUPPORTED;   // the output would change shape
    dim3 grid(imk_cdiv((int64_t)h * w, 256), batch);
    imk_klaunch(augment_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream_, img, mask, h, w, c, cm, params, img_out, mask_out);
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}
