// Implicit-GEMM convolution kernels for the tiny U-Net on gfx950 (MI355X): forward / dgrad (one kernel,
// different packed weights) and wgrad.  NHWC fp16 activations, fp32 accumulation on the matrix cores.
//
// Forward/dgrad:  D[co][pixel] = sum_k Wp[co][k] * X[k][pixel],  k = (tap, input channel).
//   * one workgroup (4 waves) = one TH x 16 output tile of one image; the input tile (+1 halo for 3x3) is
//     staged ONCE into LDS as fp16 [pixel][channel chunk of 8] with the producer's BatchNorm / max-pool /
//     upsample+add / u8->float applied on the way in, so those layers never make their own HBM pass;
//   * the pixel operand (MFMA B) is one ds_read_b128 per lane per k-step straight from that tile: the 8
//     consecutive k of a lane are the 8 channels of one chunk of one (shifted) pixel, so im2col is free;
//     pixel stride is an odd number of 16-byte chunks => conflict-free across the 16 pixels of a group;
//   * the weight operand (MFMA A) is pre-packed in fragment order in HBM (a few KB, L2 resident) and read
//     with one coalesced 16-byte load per lane per k-step, shared by the wave's P pixel groups;
//   * output channels sit on the accumulator rows, so a lane owns 4 consecutive channels of one pixel and
//     stores them as one 8-byte NHWC write; bias+ReLU and the BatchNorm statistics (sum, sum of squares
//     of the fp16-rounded outputs) are fused in the epilogue (deterministic per-tile partials, no atomics).
// Wgrad: dW[tap][ci][co] = sum_pixels X[pixel+tap][ci] * dA[pixel][co]: both MFMA operands need "8 pixels
//   of one channel" per lane, read from the [pixel][channel] LDS tiles with ds_read_b64_tr_b16.
//
// Reference layers replaced: Conv2D / BatchNormalization / MaxPooling2D / UpSampling2D+add / Lambda of
// unet.py:4-43 and their gradients inside model.fit (functions.py:218).
#include "imk_kernels.h"

typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 h4;
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T *)(p))

namespace {

constexpr int TW = 16;  // tile width = one MFMA pixel group per tile row

struct TileCoord { int b, ty0, tx0; };

__device__ __forceinline__ TileCoord tile_coord(int tile, int tiles_x, int tiles_y, int th) {
    TileCoord c;
    const int per_img = tiles_x * tiles_y;
    c.b = tile / per_img;
    const int r = tile - c.b * per_img;
    const int ty = r / tiles_x;
    c.ty0 = ty * th;
    c.tx0 = (r - ty * tiles_x) * TW;
    return c;
}

__device__ __forceinline__ f16x8 affine8(f16x8 z, const float *sc, const float *sh) {
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (f16)((float)z[j] * sc[j] + sh[j]);
    return o;
}

// One 8-channel chunk of the conv's input at conv-resolution pixel (y, x) (must be inside the image).
// s_aff: LDS table [sc | sh | sc2 | sh2], each cs_in floats.
__device__ __forceinline__ f16x8 load_chunk(const ImkInput &in, int b, int y, int x, int H, int W, int c8,
                                            const float *s_aff) {
    const int cs = in.cs_in;
    switch (in.lmode) {
        case LM_RAW: {
            const f16 *p = (const f16 *)in.in + ((size_t)(b * H + y) * W + x) * cs + c8 * 8;
            return *(const f16x8 *)p;
        }
        case LM_AFFINE: {
            const f16 *p = (const f16 *)in.in + ((size_t)(b * H + y) * W + x) * cs + c8 * 8;
            return affine8(*(const f16x8 *)p, s_aff + c8 * 8, s_aff + cs + c8 * 8);
        }
        case LM_POOL: {
            const int H2 = 2 * H, W2 = 2 * W;
            const f16 *p = (const f16 *)in.in + ((size_t)(b * H2 + 2 * y) * W2 + 2 * x) * cs + c8 * 8;
            const f16x8 z00 = *(const f16x8 *)p, z01 = *(const f16x8 *)(p + cs);
            const f16x8 z10 = *(const f16x8 *)(p + (size_t)W2 * cs), z11 = *(const f16x8 *)(p + (size_t)W2 * cs + cs);
            const float *sc = s_aff + c8 * 8, *sh = s_aff + cs + c8 * 8;
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a = (float)z00[j] * sc[j] + sh[j], bq = (float)z01[j] * sc[j] + sh[j];
                const float c = (float)z10[j] * sc[j] + sh[j], d = (float)z11[j] * sc[j] + sh[j];
                o[j] = (f16)fmaxf(fmaxf(a, bq), fmaxf(c, d));
            }
            return o;
        }
        case LM_UPADD: {
            const int Hl = H / 2, Wl = W / 2;
            const f16 *pl = (const f16 *)in.in + ((size_t)(b * Hl + (y >> 1)) * Wl + (x >> 1)) * cs + c8 * 8;
            const f16 *ps = (const f16 *)in.in2 + ((size_t)(b * H + y) * W + x) * cs + c8 * 8;
            const f16x8 lo = affine8(*(const f16x8 *)pl, s_aff + c8 * 8, s_aff + cs + c8 * 8);
            const f16x8 sk = affine8(*(const f16x8 *)ps, s_aff + 2 * cs + c8 * 8, s_aff + 3 * cs + c8 * 8);
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (f16)((float)lo[j] + (float)sk[j]);
            return o;
        }
        default: {  // LM_U8: cin <= 8 bytes per pixel, single chunk
            const uint8_t *p = (const uint8_t *)in.in + ((size_t)(b * H + y) * W + x) * in.cin;
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (f16)(j < in.cin ? (float)p[j] / 255.0f : 0.0f);
            return o;
        }
    }
}

__device__ __forceinline__ void stage_affine_table(const ImkInput &in, float *s_aff) {
    const int cs = in.cs_in;
    if (in.lmode == LM_AFFINE || in.lmode == LM_POOL || in.lmode == LM_UPADD)
        for (int i = threadIdx.x; i < cs; i += 256) { s_aff[i] = in.sc[i]; s_aff[cs + i] = in.sh[i]; }
    if (in.lmode == LM_UPADD)
        for (int i = threadIdx.x; i < cs; i += 256) { s_aff[2 * cs + i] = in.sc2[i]; s_aff[3 * cs + i] = in.sh2[i]; }
}

// =====================================================================================================
// forward / dgrad
// =====================================================================================================
template <int TH, int MT>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ImkConvArgs a, int tiles_x, int tiles_y, int mt_total,
                                                        int nc8, int ps, int nq, int ns) {
    constexpr int P = TH / 4;  // pixel groups (tile rows) per wave
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int ks3 = (a.ksize == 3);
    const int halo = ks3 ? 1 : 0;
    const int HT = TH + 2 * halo, WT = TW + 2 * halo;
    uint8_t *s_tile = smem;
    float *s_aff = reinterpret_cast<float *>(smem + (size_t)HT * WT * ps * 16);
    const int t = threadIdx.x;
    const TileCoord tc = tile_coord(blockIdx.x, tiles_x, tiles_y, TH);
    const int H = a.H, W = a.W;

    stage_affine_table(a.x, s_aff);
    if (a.x.lmode != LM_RAW && a.x.lmode != LM_U8) __syncthreads();

    // ---- stage the input tile -------------------------------------------------------------------
    const int n_items = HT * WT * nc8;
    for (int i = t; i < n_items; i += 256) {
        const int pix = i / nc8;
        const int c8 = i - pix * nc8;
        const int py = pix / WT, px = pix - py * WT;
        const int y = tc.ty0 + py - halo, x = tc.tx0 + px - halo;
        f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (y >= 0 && y < H && x >= 0 && x < W) v = load_chunk(a.x, tc.b, y, x, H, W, c8, s_aff);
        *reinterpret_cast<f16x8 *>(s_tile + ((size_t)pix * ps + c8) * 16) = v;
    }
    __syncthreads();

    // ---- MFMA main loop -------------------------------------------------------------------------
    const int lane = t & 63, wave = t >> 6, n = lane & 15, g = lane >> 4;
    const int ct0 = blockIdx.y * MT;
    int base[P];
#pragma unroll
    for (int p = 0; p < P; ++p) base[p] = ((wave * P + p) * WT + n) * ps * 16;
    f32x4 acc[MT][P];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int p = 0; p < P; ++p) acc[m][p] = f32x4{0, 0, 0, 0};

    int q = g;
    int tap = q / nc8;
    int c8 = q - tap * nc8;
    int ty = ks3 ? tap / 3 : 0, tx = ks3 ? tap - 3 * (tap / 3) : 0;
    const f16 *wp = a.wpk + (size_t)lane * 8;
    for (int s = 0; s < ns; ++s) {
        const bool vq = q < nq;
        const int off = vq ? ((ty * WT + tx) * ps + c8) * 16 : 0;
        f16x8 af[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int ct = ct0 + m;
            af[m] = (ct < mt_total) ? *reinterpret_cast<const f16x8 *>(wp + ((size_t)ct * ns + s) * 512)
                                    : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            f16x8 bf = *reinterpret_cast<const f16x8 *>(s_tile + base[p] + off);
            if (!vq) bf = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[m], bf, acc[m][p], 0, 0, 0);
        }
        q += 4;
        c8 += 4;
        while (c8 >= nc8) {
            c8 -= nc8;
            if (++tx == 3) { tx = 0; ++ty; }
        }
    }

    // ---- epilogue ---------------------------------------------------------------------------------
    const int x = tc.tx0 + n;
    float s1[MT][4], s2[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[m][r] = s2[m][r] = 0.f;
    const bool want_stats = (a.epi == EP_RELU) && a.stats_partial;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int co0 = (ct0 + m) * 16 + 4 * g;
        if (co0 >= a.cs_out) continue;
        float bias[4] = {0, 0, 0, 0};
        if (a.epi == EP_RELU && a.bias)
#pragma unroll
            for (int r = 0; r < 4; ++r) bias[r] = (co0 + r < a.cout) ? a.bias[co0 + r] : 0.f;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int y = tc.ty0 + wave * P + p;
            if (y >= H || x >= W) continue;
            const size_t o = ((size_t)(tc.b * H + y) * W + x) * a.cs_out + co0;
            f16x4 v;
            if (a.epi == EP_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (f16)fmaxf(acc[m][p][r] + bias[r], 0.f);
            } else if (a.epi == EP_MASK) {
                const f16x4 mk = *reinterpret_cast<const f16x4 *>(a.mask + o);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = ((float)mk[r] > 0.f) ? (f16)acc[m][p][r] : (f16)0.f;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (f16)acc[m][p][r];
            }
            *reinterpret_cast<f16x4 *>(a.out + o) = v;
            if (want_stats)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float f = (float)v[r]; s1[m][r] += f; s2[m][r] += f * f; }
        }
    }
    if (want_stats) {  // workgroup-uniform branch
        __syncthreads();  // everyone is done reading the tile; reuse its LDS
        float *s_red = reinterpret_cast<float *>(smem);  // [4 waves][2][16*MT]
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v1 = wave_sum<16>(s1[m][r]), v2 = wave_sum<16>(s2[m][r]);
                if (n == 0) {
                    s_red[(wave * 2 + 0) * 16 * MT + m * 16 + 4 * g + r] = v1;
                    s_red[(wave * 2 + 1) * 16 * MT + m * 16 + 4 * g + r] = v2;
                }
            }
        __syncthreads();
        if (t < 2 * 16 * MT) {
            const int which = t / (16 * MT), c = t - which * 16 * MT;
            const int co = ct0 * 16 + c;
            if (co < a.cs_out) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) v += s_red[(w * 2 + which) * 16 * MT + c];
                a.stats_partial[(size_t)blockIdx.x * 2 * a.cs_out + which * a.cs_out + co] = v;
            }
        }
    }
}

// =====================================================================================================
// wgrad
// =====================================================================================================
constexpr int WG_STRIDE_H = 24;  // halfs per pixel in the wgrad LDS slices (16 channels + 8 pad = 48 B)

__global__ __launch_bounds__(256) void wgrad_mfma_kernel(ImkWgradArgs a, int tiles_x, int tiles_y, int n_tiles,
                                                         int cit_n, int cot_n, int nc8_in, int nc8_out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int ks3 = (a.ksize == 3);
    const int halo = ks3 ? 1 : 0;
    const int HT = 16 + 2 * halo, WT = TW + 2 * halo;
    const int T = ks3 ? 9 : 1;
    f16 *s_x = reinterpret_cast<f16 *>(smem);
    f16 *s_d = s_x + HT * WT * WG_STRIDE_H;
    float *s_aff = reinterpret_cast<float *>(s_d + 256 * WG_STRIDE_H);
    const int t = threadIdx.x;
    const int pair = blockIdx.y;
    const int cit = pair / cot_n, cot = pair - cit * cot_n;
    const int H = a.H, W = a.W;
    const int lane = t & 63, wave = t >> 6, g = lane >> 4, i16 = lane & 15, qq = i16 >> 2, pp = i16 & 3;

    stage_affine_table(a.x, s_aff);

    f32x4 acc[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = f32x4{0, 0, 0, 0};
    f16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (f16)(i16 == 0 ? 1.0f : 0.0f);

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const TileCoord tc = tile_coord(tile, tiles_x, tiles_y, 16);
        __syncthreads();  // previous tile's reads are done (also covers the affine table on the first pass)
        // x slice: 16 channels (2 chunks) of every halo-tile pixel
        for (int i = t; i < HT * WT * 2; i += 256) {
            const int pix = i >> 1, c8l = i & 1;
            const int py = pix / WT, px = pix - py * WT;
            const int y = tc.ty0 + py - halo, x = tc.tx0 + px - halo;
            const int c8 = 2 * cit + c8l;
            f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (c8 < nc8_in && y >= 0 && y < H && x >= 0 && x < W) v = load_chunk(a.x, tc.b, y, x, H, W, c8, s_aff);
            *reinterpret_cast<f16x8 *>(s_x + pix * WG_STRIDE_H + c8l * 8) = v;
        }
        // dA slice
        for (int i = t; i < 256 * 2; i += 256) {
            const int pix = i >> 1, c8l = i & 1;
            const int py = pix >> 4, px = pix & 15;
            const int y = tc.ty0 + py, x = tc.tx0 + px;
            const int c8 = 2 * cot + c8l;
            f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (c8 < nc8_out && y < H && x < W)
                v = *reinterpret_cast<const f16x8 *>(a.dA + ((size_t)(tc.b * H + y) * W + x) * a.cs_out + c8 * 8);
            *reinterpret_cast<f16x8 *>(s_d + pix * WG_STRIDE_H + c8l * 8) = v;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int r0 = 2 * (wave + 4 * kk);          // tile rows r0, r0+1 form this k-step's 32 pixels
            const int row = r0 + (g >> 1);
            const int xx = 8 * (g & 1) + qq;             // +4h
            // B operand: dA[pixel][co]
            const f16 *pb = s_d + (row * 16 + xx) * WG_STRIDE_H + 4 * pp;
            const h4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(LDS_PTR(h4, pb));
            const h4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(LDS_PTR(h4, pb + 4 * WG_STRIDE_H));
            f16x8 bf;
#pragma unroll
            for (int e = 0; e < 4; ++e) { bf[e] = (f16)b0[e]; bf[4 + e] = (f16)b1[e]; }
            const f16 *pa = s_x + (row * WT + xx) * WG_STRIDE_H + 4 * pp;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if (tap < T) {
                    const int ty = ks3 ? tap / 3 : 0, tx = ks3 ? tap % 3 : 0;
                    const f16 *p = pa + (ty * WT + tx) * WG_STRIDE_H;
                    const h4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(LDS_PTR(h4, p));
                    const h4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(LDS_PTR(h4, p + 4 * WG_STRIDE_H));
                    f16x8 af;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { af[e] = (f16)a0[e]; af[4 + e] = (f16)a1[e]; }
                    acc[tap] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, acc[tap], 0, 0, 0);
                }
            }
            acc[9] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, bf, acc[9], 0, 0, 0);  // column sums -> bias grad
        }
    }
    // ---- reduce the 4 waves' accumulators through LDS, write this workgroup's partial ----------------
    __syncthreads();
    float *s_acc = reinterpret_cast<float *>(smem);  // [4][10][256]
#pragma unroll
    for (int i = 0; i < 10; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s_acc[(wave * 10 + i) * 256 + r * 64 + lane] = acc[i][r];
    __syncthreads();
    float *dst = a.partial + ((size_t)blockIdx.x * gridDim.y + pair) * (T + 1) * 256;
    for (int i = 0; i <= T; ++i) {
        const int src = (i == T) ? 9 : i;
        dst[i * 256 + t] = s_acc[(0 * 10 + src) * 256 + t] + s_acc[(1 * 10 + src) * 256 + t] +
                           s_acc[(2 * 10 + src) * 256 + t] + s_acc[(3 * 10 + src) * 256 + t];
    }
}

// stage 1 of the deterministic split reduction: [n_split][n_tiles][256] -> [n_chunks][n_tiles][256],
// 16 splits per chunk; coalesced (thread = element of a 16x16 tile).
constexpr int WG_RED_CHUNK = 16;
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ partial, int n_split, int n_tiles,
                                                           float *__restrict__ red) {
    const int tile = blockIdx.x, chunk = blockIdx.y, t = threadIdx.x;
    const int s0 = chunk * WG_RED_CHUNK, s1 = min(n_split, s0 + WG_RED_CHUNK);
    const size_t stride = (size_t)n_tiles * 256;
    const float *p = partial + (size_t)tile * 256 + t;
    float v[WG_RED_CHUNK];
#pragma unroll
    for (int i = 0; i < WG_RED_CHUNK; ++i) v[i] = (s0 + i < s1) ? p[(size_t)(s0 + i) * stride] : 0.f;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < WG_RED_CHUNK; ++i) acc += v[i];
    red[((size_t)chunk * n_tiles + tile) * 256 + t] = acc;
}

__global__ __launch_bounds__(256) void wgrad_finalize_kernel(const float *__restrict__ partial, int n_split, int T,
                                                             int cin, int cout, int cit_n, int cot_n,
                                                             const float *__restrict__ inv_scale_ptr,
                                                             float *__restrict__ dw, float *__restrict__ db,
                                                             float *__restrict__ found_inf) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n_w = T * cin * cout;
    if (i >= n_w + cout) return;
    int tap, ci, co;
    if (i < n_w) {
        co = i % cout;
        const int r = i / cout;
        ci = r % cin;
        tap = r / cin;
    } else {
        co = i - n_w; ci = 0; tap = T;
    }
    const int cit = ci >> 4, cot = co >> 4, m = ci & 15, nn = co & 15;
    const int e = (m & 3) * 64 + (m >> 2) * 16 + nn;
    const int n_pairs = cit_n * cot_n;
    const size_t stride = (size_t)n_pairs * (T + 1) * 256;
    const float *p = partial + ((size_t)(cit * cot_n + cot) * (T + 1) + tap) * 256 + e;
    float s = 0.f;
    for (int k = 0; k < n_split; ++k) s += p[(size_t)k * stride];
    s *= *inv_scale_ptr;
    if (!isfinite(s)) *found_inf = 1.0f;
    if (i < n_w) dw[i] = s; else db[co] = s;
}

// fp32 HWIO -> fp16 MFMA-A fragment order.  element index = ((ct*ns + s)*64 + lane)*8 + j
__global__ __launch_bounds__(256) void pack_conv_kernel(const float *__restrict__ w, int T, int cin, int cout,
                                                        int transposed, int m_dim, int k_dim, int nc8, int ns,
                                                        int total, f16 *__restrict__ dst) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int j = i & 7, lane = (i >> 3) & 63;
    const int cs = i >> 9;
    const int s = cs % ns, ct = cs / ns;
    const int m = lane & 15, g = lane >> 4;
    const int q = 4 * s + g;
    const int tap = q / nc8, c8 = q - tap * nc8;
    const int mi = ct * 16 + m, ki = c8 * 8 + j;
    float v = 0.f;
    if (tap < T && mi < m_dim && ki < k_dim) {
        if (!transposed) v = w[((size_t)tap * cin + ki) * cout + mi];                 // m = co, k = ci
        else v = w[((size_t)(T - 1 - tap) * cin + mi) * cout + ki];                    // m = ci, k = co, flipped taps
    }
    dst[i] = (f16)v;
}

// all conv layers of a model in one launch: blockIdx.y = job
__global__ __launch_bounds__(256) void pack_conv_batched_kernel(ImkPackJobs jobs) {
    const ImkPackJob &jb = jobs.j[blockIdx.y];
    const int T = jb.ksize == 3 ? 9 : 1;
    const int m_dim = jb.transposed ? jb.cin : jb.cout, k_dim = jb.transposed ? jb.cout : jb.cin;
    const int nc8 = ((k_dim + 7) & ~7) / 8;
    const int ns = (T * nc8 + 3) / 4;
    const int total = ((m_dim + 15) / 16) * ns * 512;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int j = i & 7, lane = (i >> 3) & 63;
        const int cs = i >> 9;
        const int s = cs % ns, ct = cs / ns;
        const int m = lane & 15, g = lane >> 4;
        const int q = 4 * s + g;
        const int tap = q / nc8, c8 = q - tap * nc8;
        const int mi = ct * 16 + m, ki = c8 * 8 + j;
        float v = 0.f;
        if (tap < T && mi < m_dim && ki < k_dim) {
            if (!jb.transposed) v = jb.w[((size_t)tap * jb.cin + ki) * jb.cout + mi];
            else v = jb.w[((size_t)(T - 1 - tap) * jb.cin + mi) * jb.cout + ki];
        }
        jb.dst[i] = (f16)v;
    }
}

}  // namespace

int imk_launch_pack_jobs(const ImkPackJobs &jobs, hipStream_t stream) {
    if (jobs.n <= 0) return IMK_OK;
    pack_conv_batched_kernel<<<dim3(16, jobs.n), 256, 0, stream>>>(jobs);
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}

// -----------------------------------------------------------------------------------------------------
static inline int odd_ps(int nc8) { return nc8 | 1; }

// Tile height: 16 rows unless the LDS tile would exceed 64 KB (wide layers), then 8.
static inline int conv_tile_h(int cs_in, int ksize) {
    const int halo = ksize == 3 ? 1 : 0;
    const size_t b16 = (size_t)(16 + 2 * halo) * (TW + 2 * halo) * odd_ps(cs_in / 8) * 16;
    return b16 > 64 * 1024 ? 8 : 16;
}

int imk_conv_num_tiles(int B, int H, int W, int cs_in, int ksize) {
    return B * imk_cdiv(H, conv_tile_h(cs_in, ksize)) * imk_cdiv(W, TW);
}

// ---- optional per-launch event timing (bench.py roofline) ---------------------------------------------
#include <vector>
namespace {
struct ProfRec { hipEvent_t e0, e1; int variant; double bytes; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof;       // recorded launches since the last collect
std::vector<hipEvent_t> g_ev_pool; // recycled events

hipEvent_t prof_event() {
    if (!g_ev_pool.empty()) { hipEvent_t e = g_ev_pool.back(); g_ev_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

double conv_algorithmic_bytes(const ImkConvArgs &a) {
    const double px = (double)a.B * a.H * a.W;
    double in_b;
    switch (a.x.lmode) {
        case LM_POOL: in_b = 4.0 * px * a.x.cs_in * 2; break;                     // reads the 2H x 2W tensor
        case LM_UPADD: in_b = px * a.x.cs_in * 2 + 0.25 * px * a.x.cs_in * 2; break;  // skip + low-res tensor
        case LM_U8: in_b = px * a.x.cin; break;
        default: in_b = px * a.x.cs_in * 2;
    }
    double out_b = px * a.cs_out * 2;
    if (a.epi == EP_MASK) out_b += px * a.cs_out * 2;
    return in_b + out_b;
}
}  // namespace

extern "C" int imk_prof_enable(int on) { g_prof_on = on != 0; return IMK_OK; }

extern "C" int imk_prof_collect(int64_t *count, double *ms, double *bytes) {
    IMK_CHECK_ARG(count && ms && bytes);
    for (int v = 0; v < IMK_PROF_VARIANTS; ++v) { count[v] = 0; ms[v] = 0; bytes[v] = 0; }
    for (auto &r : g_prof) {
        IMK_HIP(hipEventSynchronize(r.e1));
        float t = 0.f;
        IMK_HIP(hipEventElapsedTime(&t, r.e0, r.e1));
        count[r.variant] += 1; ms[r.variant] += t; bytes[r.variant] += r.bytes;
        g_ev_pool.push_back(r.e0); g_ev_pool.push_back(r.e1);
    }
    g_prof.clear();
    return IMK_OK;
}

template <int TH>
static int launch_conv_th(const ImkConvArgs &a, hipStream_t stream) {
    const int halo = a.ksize == 3 ? 1 : 0;
    const int nc8 = a.x.cs_in / 8, ps = odd_ps(nc8);
    const int T = a.ksize == 3 ? 9 : 1;
    const int nq = T * nc8, ns = (nq + 3) / 4;
    const int mt_total = (a.cout + 15) / 16;
    const int tiles_x = imk_cdiv(a.W, TW), tiles_y = imk_cdiv(a.H, TH);
    const size_t tile_bytes = (size_t)(TH + 2 * halo) * (TW + 2 * halo) * ps * 16;
    const size_t stats_bytes = 4 * 2 * 16 * 4 * sizeof(float);
    size_t lds = tile_bytes + 4 * (size_t)a.x.cs_in * sizeof(float);
    if (lds < stats_bytes) lds = stats_bytes;
    const int mt = mt_total >= 4 ? 4 : (mt_total >= 2 ? 2 : 1);
    dim3 grid(a.B * tiles_x * tiles_y, imk_cdiv(mt_total, mt));
    if (lds > 160 * 1024) return IMK_EUNSUPPORTED;  // wider than ~384 input channels on a 3x3: needs K passes
    auto launch = [&](auto kern) -> int {
        if (lds > 64 * 1024)  // above the default dynamic-LDS limit: opt in (idempotent, no sync)
            IMK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        kern<<<grid, 256, lds, stream>>>(a, tiles_x, tiles_y, mt_total, nc8, ps, nq, ns);
        return IMK_OK;
    };
    ProfRec pr{};
    if (g_prof_on) {
        pr.e0 = prof_event(); pr.e1 = prof_event();
        pr.variant = (TH == 8 ? 3 : 0) + (mt == 4 ? 2 : (mt == 2 ? 1 : 0));
        pr.bytes = conv_algorithmic_bytes(a);
        IMK_HIP(hipEventRecord(pr.e0, stream));
    }
    int rc;
    if (mt == 4) rc = launch(conv_mfma_kernel<TH, 4>);
    else if (mt == 2) rc = launch(conv_mfma_kernel<TH, 2>);
    else rc = launch(conv_mfma_kernel<TH, 1>);
    if (rc) return rc;
    if (g_prof_on) {
        IMK_HIP(hipEventRecord(pr.e1, stream));
        g_prof.push_back(pr);
    }
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}

int imk_launch_conv(const ImkConvArgs &a, hipStream_t stream) {
    IMK_CHECK_ARG(a.x.in && a.wpk && a.out && a.B > 0 && a.H > 0 && a.W > 0);
    IMK_CHECK_ARG(a.ksize == 1 || a.ksize == 3);
    IMK_CHECK_ARG(a.x.cs_in % 8 == 0 && a.cs_out % 8 == 0 && a.x.cs_in >= a.x.cin && a.cs_out >= a.cout);
    IMK_CHECK_ARG(a.x.lmode != LM_U8 || (a.x.cs_in == 8 && a.x.cin <= 8));
    if (a.x.cs_in > 512) return IMK_EUNSUPPORTED;
    if (conv_tile_h(a.x.cs_in, a.ksize) == 16) return launch_conv_th<16>(a, stream);
    return launch_conv_th<8>(a, stream);
}

int imk_wgrad_splits(int B, int H, int W, int cin, int cout) {
    const int n_tiles = B * imk_cdiv(H, 16) * imk_cdiv(W, TW);
    const int n_pairs = ((imk_pad8(cin) + 15) / 16) * ((imk_pad8(cout) + 15) / 16);
    int s = 1024 / n_pairs;
    if (s < 1) s = 1;
    if (s > n_tiles) s = n_tiles;
    return s;
}

size_t imk_wgrad_partial_floats(int B, int H, int W, int ksize, int cin, int cout) {
    const int n_pairs = ((imk_pad8(cin) + 15) / 16) * ((imk_pad8(cout) + 15) / 16);
    const int T = ksize == 3 ? 9 : 1;
    const size_t ns = (size_t)imk_wgrad_splits(B, H, W, cin, cout);
    return (ns + (ns + WG_RED_CHUNK - 1) / WG_RED_CHUNK) * n_pairs * (T + 1) * 256;  // partials + stage-1 scratch
}

int imk_launch_wgrad(const ImkWgradArgs &a, hipStream_t stream) {
    IMK_CHECK_ARG(a.x.in && a.dA && a.partial && a.B > 0 && a.H > 0 && a.W > 0 && a.n_split > 0);
    IMK_CHECK_ARG(a.ksize == 1 || a.ksize == 3);
    IMK_CHECK_ARG(a.x.cs_in % 8 == 0 && a.cs_out % 8 == 0);
    const int halo = a.ksize == 3 ? 1 : 0;
    const int cit_n = (a.x.cs_in + 15) / 16, cot_n = (a.cs_out + 15) / 16;
    const int tiles_x = imk_cdiv(a.W, TW), tiles_y = imk_cdiv(a.H, 16);
    const int n_tiles = a.B * tiles_x * tiles_y;
    size_t lds = ((size_t)(16 + 2 * halo) * (TW + 2 * halo) + 256) * WG_STRIDE_H * sizeof(f16) + 4 * (size_t)a.x.cs_in * sizeof(float);
    const size_t red = 4 * 10 * 256 * sizeof(float);
    if (lds < red) lds = red;
    dim3 grid(a.n_split, cit_n * cot_n);
    wgrad_mfma_kernel<<<grid, 256, lds, stream>>>(a, tiles_x, tiles_y, n_tiles, cit_n, cot_n, a.x.cs_in / 8, a.cs_out / 8);
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}

int imk_launch_wgrad_finalize(const float *partial, int n_split, int ksize, int cin, int cout,
                              const float *inv_scale_ptr, float *dw, float *db, float *found_inf, hipStream_t stream) {
    const int T = ksize == 3 ? 9 : 1;
    const int cit_n = (imk_pad8(cin) + 15) / 16, cot_n = (imk_pad8(cout) + 15) / 16;
    const int total = T * cin * cout + cout;
    if (n_split > WG_RED_CHUNK) {  // two-stage: the second stage then walks at most 64 values per element
        const int n_tiles = cit_n * cot_n * (T + 1);
        const int n_chunks = imk_cdiv(n_split, WG_RED_CHUNK);
        float *red = const_cast<float *>(partial) + (size_t)n_split * n_tiles * 256;  // scratch behind the partials
        wgrad_reduce_kernel<<<dim3(n_tiles, n_chunks), 256, 0, stream>>>(partial, n_split, n_tiles, red);
        IMK_LAUNCH_CHECK();
        partial = red;
        n_split = n_chunks;
    }
    wgrad_finalize_kernel<<<imk_cdiv(total, 256), 256, 0, stream>>>(partial, n_split, T, cin, cout, cit_n, cot_n,
                                                                   inv_scale_ptr, dw, db, found_inf);
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}

size_t imk_packed_conv_halfs(int ksize, int cin, int cout, int transposed) {
    const int T = ksize == 3 ? 9 : 1;
    const int m_dim = transposed ? cin : cout, k_dim = transposed ? cout : cin;
    const int nc8 = imk_pad8(k_dim) / 8;
    const int ns = (T * nc8 + 3) / 4;
    return (size_t)((m_dim + 15) / 16) * ns * 512;
}

int imk_launch_pack_conv(const float *w, int ksize, int cin, int cout, int transposed, f16 *dst, hipStream_t stream) {
    const int T = ksize == 3 ? 9 : 1;
    const int m_dim = transposed ? cin : cout, k_dim = transposed ? cout : cin;
    const int nc8 = imk_pad8(k_dim) / 8;
    const int ns = (T * nc8 + 3) / 4;
    const int total = (int)imk_packed_conv_halfs(ksize, cin, cout, transposed);
    pack_conv_kernel<<<imk_cdiv(total, 256), 256, 0, stream>>>(w, T, cin, cout, transposed, m_dim, k_dim, nc8, ns, total, dst);
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}
