// GEMM-class convolution kernels for the WIDE layers of the U-Net / EvalNet on gfx950 (more than 32 channels on a side:
// levels 2-4 at alpha = 1, everything below full resolution at alpha = 2 -- BASELINE configs[2..4] and the IM+ width
// schedule, Cityscapes/11_Cityscapes_IM+.py:48).  Same arithmetic, same packed weights and the same k order as the per-tile
// kernel of imk_conv.hip (conv_mfma_kernel), so stored activations are bit-identical to it; what changes is how much of
// each operand a workgroup pulls per MFMA:
//
//   forward / dgrad (conv_gemm_kernel):  D[co][pixel] = sum_k Wp[co][k] * X[k][pixel],  k = (pass, tap, channel)
//     * workgroup tile = 128 pixels (8 rows x 16) x up to 128 output channels, 4 waves as 2 (pixels) x 2 (channels), a wave
//       owns 64 pixels x 64 (or 32) channels = 16 (8) accumulator fragments: every staged input chunk feeds 8 (4)
//       channel tiles instead of 1-4, every weight fragment 4 pixel groups;
//     * the input tile (+ halo) is staged per channel STAGE (32 channels of a 3x3, up to 64 of a 1x1) into one of TWO LDS
//       buffers: the global loads of stage s + 2 are in flight (registers) and stage s + 1 is transformed (BatchNorm /
//       pool / upsample + add / BatchNorm backward on load, imk_stage.h) and written while stage s feeds the matrix cores;
//       one barrier per stage;
//     * weight fragments stream global -> registers (fragment-ordered pack: one coalesced KB per wave and k-step, L2
//       resident), one k-step ahead of the MFMAs, across stage boundaries -- they never touch LDS, so two workgroups share a
//       compute unit and one's staging / epilogue overlaps the other's MFMAs;
//     * epilogue through LDS: the accumulators leave as [pixel][channel] fp16, and the elementwise part (ReLU mask of a
//       dgrad, BatchNorm statistics / BatchNorm-gradient statistics) runs on 16-byte chunks with coalesced loads and
//       stores; statistics are per-workgroup partial rows summed in a fixed order (bit-reproducible, no atomics);
//     * workgroups that share an input tile (the output-channel groups) are neighbours on one XCD (ids are dealt
//       round-robin over the 8 XCDs), so the tile's second read is an L2 hit.
//
//   weight gradient (wgrad_gemm_kernel):  dW[tap][ci][co] = sum_pixels X[pixel + tap][ci] * dA[pixel][co]
//     * workgroup = up to 64 input x 64 output channels x all taps, a wave = one 16-channel input tile x 4 output tiles x 9
//       taps (36 accumulator fragments): x and dA are read once per 64 x 64 channel block instead of once per 16 x 16;
//     * no cross-wave reduction: every wave owns its (input tile, output tile) pairs and writes its partial rows in the
//       layout the split reduction of imk_conv.hip (wgf_stage1 / 2) already reads.
//
// Reference layers replaced: Conv2D (+ the BatchNormalization / MaxPooling2D / UpSampling2D + add around it) of
// unet.py:11-43 and evalnet.py:8-21 and their gradients inside model.fit (functions.py:218).
#include <cstdlib>
#include <type_traits>
#include "imk_stage.h"

namespace {

constexpr int G_PM = 4, G_WM = 2, G_WN = 2;   // pixel groups per wave; waves along pixels / output channels
constexpr int G_TH = G_PM * G_WM;             // tile rows (of 16 pixels)
constexpr int G_BM = 16 * G_TH;               // pixels per workgroup tile

struct GemmGeom {
    int tiles_x, tiles_y, n_sp, gy, mt_total;
    int nc8, nc8p, n_pass, nsp, ns_total;     // packed-weight geometry: chunks, chunks per pass, passes, k-steps per pass / in all
    int spp, cps, n_stage, ps;                // passes per stage, chunks per stage, stages, LDS pixel pitch in chunks (odd)
};

template <int LM, bool KS3, int PN>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(ImkConvArgs a, GemmGeom gm) {
    constexpr int NT = 256, PM = G_PM, WN = G_WN, TH = G_TH;
    constexpr int BN = 16 * PN * WN;
    constexpr int halo = KS3 ? 1 : 0, T = KS3 ? 9 : 1;
    constexpr int HT = TH + 2 * halo, WT = TW + 2 * halo;
    constexpr int NX = KS3 ? 3 : 4;           // staging items per thread: 10 x 18 pixels x 4 chunks / 128 pixels x 8 chunks
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    // workgroup id -> (spatial tile, output-channel group): the groups of one tile are consecutive on one XCD
    const int id = blockIdx.x;
    const int xcd = id & 7, jj = id >> 3;
    const int by = jj % gm.gy;
    const int sp = (jj / gm.gy) * 8 + xcd;
    if (sp >= gm.n_sp) return;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, n = lane & 15, g = lane >> 4;
    const int wm = wave & (G_WM - 1), wn = wave / G_WM;
    const int H = a.H, W = a.W;
    const int nc8 = gm.nc8, nc8p = gm.nc8p, nsp = gm.nsp, ns_total = gm.ns_total, spp = gm.spp, cps = gm.cps, ps = gm.ps;
    PTile tc;
    {
        const int per_img = gm.tiles_x * gm.tiles_y;
        const int b = sp / per_img, r = sp - b * per_img, ty = r / gm.tiles_x;
        tc.b = __builtin_amdgcn_readfirstlane(b); tc.r = __builtin_amdgcn_readfirstlane(r);
        tc.ty0 = __builtin_amdgcn_readfirstlane(ty * TH);
        tc.tx0 = __builtin_amdgcn_readfirstlane((r - ty * gm.tiles_x) * TW);
    }
    const PSrc<LM> src = psrc_of<LM, 0>(a.x, tc, halo, HT, WT, H, W);
    const size_t tile_bytes = (size_t)HT * WT * ps * 16;
    uint8_t *s_tile0 = smem, *s_tile1 = smem + tile_bytes;
    float *s_aff = reinterpret_cast<float *>(smem + 2 * tile_bytes);
    stage_affine_table(a.x, s_aff);

    // ---- staging items of this thread (the same every stage; the stage only moves the channel base) -------------------
    const int n_items = HT * WT * cps;
    int x_dst[NX], x_py[NX], x_px[NX], x_c8[NX];
#pragma unroll
    for (int u = 0; u < NX; ++u) {
        const int i = t + u * NT;
        const int ii = i < n_items ? i : 0;            // idle slots repeat item 0 (a cache hit)
        const int pix = ii / cps;
        x_c8[u] = ii - pix * cps;
        x_py[u] = pix / WT;
        x_px[u] = pix - x_py[u] * WT;
        x_dst[u] = i < n_items ? (pix * ps + x_c8[u]) * 16 : -1;
    }
    RawChunk<LM> xr[NX];
    unsigned xok = 0;
    auto issue = [&](int st) {                          // unconditional, clamped loads (see conv_mfma_body)
        xok = 0;
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int c8 = st * cps + x_c8[u];
            const bool live = c8 < nc8;
            const bool inside = psrc_load<LM, 0>(src, x_py[u], x_px[u], live ? c8 : nc8 - 1, W, xr[u]);
            xok |= ((live && inside) ? 1u : 0u) << u;
        }
    };
    auto commit = [&](int st, uint8_t *buf) {
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            if (x_dst[u] >= 0) {
                const int c8 = min(st * cps + x_c8[u], nc8 - 1);
                f16x8 v = raw_transform<LM>(xr[u], s_aff, a.x.cs_in, c8, a.x.cin, a.x.u8_div);
                if (!(xok & (1u << u))) v = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                *reinterpret_cast<f16x8 *>(buf + x_dst[u]) = v;
            }
        }
    };

    // ---- operands ------------------------------------------------------------------------------------------------------
    const int ct0 = by * (BN / 16);
    const f16 *wbase[PN];
#pragma unroll
    for (int m = 0; m < PN; ++m) {
        int ct = ct0 + wn * PN + m;
        if (ct >= gm.mt_total) ct = gm.mt_total - 1;     // padding tiles of the last group: computed, never stored
        wbase[m] = a.wpk + (size_t)ct * ns_total * 512 + lane * 8;
    }
    auto loadA = [&](f16x8 (&af)[PN], int F) {
        const int fc = min(F, ns_total - 1);
#pragma unroll
        for (int m = 0; m < PN; ++m) af[m] = *reinterpret_cast<const f16x8 *>(wbase[m] + (size_t)fc * 512);
    };
    int base[PM];
#pragma unroll
    for (int p = 0; p < PM; ++p) base[p] = ((wm * PM + p) * WT + n) * ps * 16;
    f32x4 acc[PN][PM];
#pragma unroll
    for (int m = 0; m < PN; ++m)
#pragma unroll
        for (int p = 0; p < PM; ++p) acc[m][p] = f32x4{0, 0, 0, 0};
    auto mma = [&](const f16x8 (&af)[PN], const f16x8 (&bf)[PM]) {
#pragma unroll
        for (int p = 0; p < PM; ++p)
#pragma unroll
            for (int m = 0; m < PN; ++m) acc[m][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[m], bf[p], acc[m][p], 0, 0, 0);
    };

    const int n_stage = gm.n_stage;
    issue(0);
    f16x8 a0[PN], a1[PN];
    loadA(a0, 0);
    __syncthreads();                                     // affine table visible
    commit(0, s_tile0);
    issue(min(1, n_stage - 1));
    __syncthreads();
    const int mg_q = (65536 + nc8p - 1) / nc8p;          // q / nc8p as multiply-shift (q < 40)
    int F = 0;                                           // flat k-step index of the stage's first step
    for (int st = 0; st < n_stage; ++st) {
        const uint8_t *cur = (st & 1) ? s_tile1 : s_tile0;
        uint8_t *nxt = (st & 1) ? s_tile0 : s_tile1;
        if (st + 1 < n_stage) commit(st + 1, nxt);       // its loads were issued a whole stage ago
        issue(min(st + 2, n_stage - 1));                 // in flight during this stage's MFMAs
        const int np = min(spp, gm.n_pass - st * spp);   // passes of this stage
        const int nk = np * nsp;                         // k-steps
        int pi = 0, s = 0;                               // (pass in stage, step in pass) of the next pixel-operand read
        auto loadB = [&](f16x8 (&bf)[PM]) {
            const int q = 4 * s + g;
            const int tap = (q * mg_q) >> 16, c8 = q - tap * nc8p;
            const int ty = KS3 ? (tap * 11) >> 5 : 0, tx = KS3 ? tap - 3 * ty : 0;
            const bool vq = (q < T * nc8p) && ((st * spp + pi) * nc8p + c8 < nc8) && (pi < np);
            const int off = vq ? ((ty * WT + tx) * ps + pi * nc8p + c8) * 16 : 0;   // invalid k-slots: zero weights, any finite data
#pragma unroll
            for (int p = 0; p < PM; ++p) bf[p] = *reinterpret_cast<const f16x8 *>(cur + base[p] + off);
            if (++s == nsp) { s = 0; ++pi; }
        };
        f16x8 b0[PM], b1[PM];
        loadB(b0);
        for (int f = 0; f < nk; f += 2) {
            loadA(a1, F + f + 1);
            loadB(b1);
            mma(a0, b0);
            loadA(a0, F + f + 2);
            loadB(b0);
            if (f + 1 < nk) mma(a1, b1);
        }
        if (nk & 1) {                                    // the look-ahead fragment IS the next stage's first
#pragma unroll
            for (int m = 0; m < PN; ++m) a0[m] = a1[m];
        }
        F += nk;
        __syncthreads();
    }

    // ---- epilogue ------------------------------------------------------------------------------------------------------
    constexpr int OP = BN + 8;                           // halfs per pixel of the output tile in LDS (17 / 9 chunks: odd)
    f16 *s_out = reinterpret_cast<f16 *>(smem);
    const int cs_o = a.cs_out;
    {
        float bias[PN][4];
#pragma unroll
        for (int m = 0; m < PN; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = (ct0 + wn * PN + m) * 16 + 4 * g + r;
                bias[m][r] = (a.epi == EP_RELU && a.bias && co < a.cout) ? a.bias[co] : 0.f;
            }
        const bool relu = a.epi == EP_RELU;
#pragma unroll
        for (int m = 0; m < PN; ++m)
#pragma unroll
            for (int p = 0; p < PM; ++p) {
                f16x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = relu ? (f16)fmaxf(acc[m][p][r] + bias[m][r], 0.f) : (f16)acc[m][p][r];
                *reinterpret_cast<f16x4 *>(s_out + ((wm * PM + p) * 16 + n) * OP + (wn * PN + m) * 16 + 4 * g) = v;
            }
    }
    __syncthreads();
    constexpr int CPP = BN / 8, RPI = NT / CPP, IT = G_BM / RPI;   // chunks per pixel, pixels per sweep, sweeps
    const int cj = t % CPP, prow = t / CPP;
    const int ch0 = ct0 * 16 + cj * 8;
    const bool ch_live = ch0 < cs_o;
    const int my = H - 1 - tc.ty0, mx = W - 1 - tc.tx0;            // last row / column of the image, relative to the tile
    const unsigned pitch = (unsigned)cs_o * 2u;
    const unsigned cho = ch_live ? (unsigned)ch0 * 2u : 0u;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
    auto sweep = [&](auto EPI_T, auto STAT_T) {
        constexpr int EPI = decltype(EPI_T)::value;      // EP_RELU / EP_PLAIN / EP_MASK
        constexpr int STAT = decltype(STAT_T)::value;    // 0 none, 1 sum / sum of squares, 2 sum dy / sum dy * z
        char *ob = const_cast<char *>(pix_base(a.out, tc.b, H, W, tc.ty0, tc.tx0, pitch)) + cho;
        const char *mb = EPI == EP_MASK ? pix_base(a.mask, tc.b, H, W, tc.ty0, tc.tx0, pitch) + cho : nullptr;
        const char *zb = STAT == 2 ? pix_base(a.dystat_z, tc.b, H, W, tc.ty0, tc.tx0, pitch) + cho : nullptr;
        f16x8 mk[IT], zz[IT];
        unsigned off[IT];
#pragma unroll
        for (int k = 0; k < IT; ++k) {                   // all mask / z chunks requested up front (clamped: dead lanes read valid data)
            const int pixel = prow + k * RPI, py = pixel >> 4, px = pixel & 15;
            off[k] = __umul24(__umul24(min(py, my), W) + min(px, mx), pitch);
            if (EPI == EP_MASK) mk[k] = *reinterpret_cast<const f16x8 *>(mb + off[k]);
            if (STAT == 2) zz[k] = *reinterpret_cast<const f16x8 *>(zb + off[k]);
        }
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int pixel = prow + k * RPI, py = pixel >> 4, px = pixel & 15;
            f16x8 v = *reinterpret_cast<const f16x8 *>(s_out + pixel * OP + cj * 8);
            if (EPI == EP_MASK) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = ((float)mk[k][e] > 0.f) ? v[e] : (f16)0.f;
            }
            const bool live = ch_live && py <= my && px <= mx;
            if (live) *reinterpret_cast<f16x8 *>(ob + off[k]) = v;
            if (STAT) {
                const float lf = live ? 1.f : 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = lf * (float)v[e];
                    s1[e] += f;
                    s2[e] += f * (STAT == 2 ? (float)zz[k][e] : f);
                }
            }
        }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    const bool dystat = (a.epi != EP_RELU) && a.dystat_z && a.stats_partial;
    const bool want_stats = ((a.epi == EP_RELU) && a.stats_partial) || dystat;
    if (a.epi == EP_RELU) { if (want_stats) sweep(I0{}, I1{}); else sweep(I0{}, I0{}); }
    else if (a.epi == EP_MASK) { if (dystat) sweep(I2{}, I2{}); else sweep(I2{}, I0{}); }
    else { if (dystat) sweep(I1{}, I2{}); else sweep(I1{}, I0{}); }
    if (want_stats) {                                    // workgroup-uniform
        __syncthreads();                                 // everyone is done reading the output tile
        float *s_red = reinterpret_cast<float *>(smem);  // [2][RPI][BN]
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s_red[(0 * RPI + prow) * BN + cj * 8 + e] = s1[e];
            s_red[(1 * RPI + prow) * BN + cj * 8 + e] = s2[e];
        }
        __syncthreads();
        for (int i = t; i < 2 * BN; i += NT) {
            const int which = i / BN, c = i - which * BN;
            const int co = ct0 * 16 + c;
            if (co < cs_o) {
                float v = 0.f;
                for (int r = 0; r < RPI; ++r) v += s_red[(which * RPI + r) * BN + c];
                a.stats_partial[(size_t)sp * 2 * cs_o + which * cs_o + co] = v;
            }
        }
    }
}

// =====================================================================================================
// weight gradient of the wide layers
// =====================================================================================================
// Workgroup = (split of the pixel tiles) x (group of <= NFI input-channel tiles x <= 4 output-channel tiles, 16 channels each).
// Waves: NFI along the input-channel tiles x (4 / NFI) along the k-steps (32 pixels = two rows) of a pixel tile; a wave owns
// ONE input-channel tile x 4 output-channel tiles x all taps = 36 accumulator fragments (+ 4 for the bias gradient), and writes
// them as its own partial rows: split index = blockIdx.x * (4 / NFI) + k-part, the layout wgf_stage1 / 2 (imk_conv.hip) read.
// Pixel tile: 4 rows x 16 (two k-steps; NFI = 1: 8 rows, four k-steps), single-buffered in LDS as per-channel-tile slices
// [pixel][16 channels] (the layout of wgrad_mfma_kernel: transposed reads, 8 consecutive pixels of a 32-lane half = one
// bank row), the next tile's global loads in flight in registers during the MFMAs.
struct WgGemmGeom { int tiles_x, tiles_y, n_tiles, cit_n, cot_n, nci, nco, gi_n, go_n, fi_per, fo_per; };

template <int LM, bool BNB, bool KS3, int NFI>
__global__ __launch_bounds__(256, 2) void wgrad_gemm_kernel(ImkWgradArgs a, WgGemmGeom gm) {
    constexpr int KP = 4 / NFI;                      // waves along the k-steps
    constexpr int TR = NFI == 1 ? 8 : 4;             // tile rows (8 rows for the 1x1 forms too: measured slower, 5.77 vs 5.65 ms)
    constexpr int KS = TR / 2;                       // k-steps per tile
    constexpr int halo = KS3 ? 1 : 0, T = KS3 ? 9 : 1;
    constexpr int HT = TR + 2 * halo, WT = TW + 2 * halo;
    constexpr int NPX = HT * WT, NPD = TR * 16;      // pixels of an x slice / a dA slice
    constexpr int NX = (NPX * 2 * NFI + 255) / 256, ND = (NPD * 8 + 255) / 256;
    constexpr int H16 = WG_STRIDE_H;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    f16 *s_x = reinterpret_cast<f16 *>(smem);                       // [NFI][NPX][16]
    f16 *s_d = s_x + NFI * NPX * H16;                                // [4][NPD][16]
    float *s_aff = reinterpret_cast<float *>(s_d + 4 * NPD * H16);
    float *s_coef = s_aff + 4 * a.x.cs_in;                           // [A | B | C] of the dA-side BatchNorm backward
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 4, i16 = lane & 15, qq = i16 >> 2, pp = i16 & 3;
    const int fi = wave % NFI, kp = wave / NFI;
    const int gi = blockIdx.y / gm.go_n, go = blockIdx.y - gi * gm.go_n;
    const int cit0 = gi * gm.fi_per, cot0 = go * gm.fo_per;
    const int nfi = min(gm.fi_per, gm.cit_n - cit0), nfo = min(gm.fo_per, gm.cot_n - cot0);
    const int H = a.H, W = a.W;
    const int n_tiles = gm.n_tiles;

    stage_affine_table(a.x, s_aff);
    if (BNB)
        for (int i = t; i < 3 * a.cs_out; i += 256) s_coef[i] = a.dA_coef[i];

    f32x4 acc[4][T], accb = f32x4{0, 0, 0, 0};       // accb: row o = column sums of dA over output tile o (the bias gradient)
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int tp = 0; tp < T; ++tp) acc[o][tp] = f32x4{0, 0, 0, 0};

    // Staging items of this thread: x item i = (pixel, chunk j of the group's 2 * NFI), dA item i = (pixel, chunk j of 8);
    // item -> coordinates is recomputed per tile from compile-time divisors (a few VALU operations) instead of being held in
    // registers over the MFMA loop (the 3x3 forms sit at the 256-register limit of two waves per SIMD)
    struct Item { int py, px, c8, lds; bool live; };
    auto x_item = [&](int k) {
        const int i = t + 256 * k;
        const bool used = i < NPX * 2 * NFI;
        const int ii = used ? i : 0;                 // idle slots repeat item 0 (a cache hit), never written
        const int pix = ii / (2 * NFI), j = ii - pix * (2 * NFI);
        const int c8 = 2 * cit0 + j;
        Item it;
        it.py = pix / WT; it.px = pix - it.py * WT;
        it.live = used && (j < 2 * nfi) && (c8 < gm.nci);
        it.c8 = it.live ? c8 : 0;
        it.lds = used ? ((j >> 1) * NPX + pix) * H16 + (j & 1) * 8 : -1;
        return it;
    };
    auto d_item = [&](int k) {
        const int i = t + 256 * k;                   // NPD * 8 is a multiple of 256: no idle slots
        const int pix = i >> 3, j = i & 7;
        const int c8 = 2 * cot0 + j;
        Item it;
        it.py = pix >> 4; it.px = pix & 15;
        it.live = (j < 2 * nfo) && (c8 < gm.nco);
        it.c8 = it.live ? c8 : 0;
        it.lds = ((j >> 1) * NPD + pix) * H16 + (j & 1) * 8;
        return it;
    };
    RawChunk<LM> xr[NX];
    f16x8 dr[ND], dz[ND];
    unsigned vx = 0, vd = 0;
    auto issue = [&](int tile) {                     // unconditional, clamped loads
        const TileCoord tc = tile_coord(tile, gm.tiles_x, gm.tiles_y, TR);
        vx = vd = 0;
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const Item it = x_item(k);
            const int y = tc.ty0 + it.py - halo, x = tc.tx0 + it.px - halo;
            const bool ok = it.live && y >= 0 && y < H && x >= 0 && x < W;
            raw_load<LM>(a.x, tc.b, min(max(y, 0), H - 1), min(max(x, 0), W - 1), H, W, it.c8, xr[k]);
            vx |= (ok ? 1u : 0u) << k;
        }
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            const Item it = d_item(k);
            const int y = tc.ty0 + it.py, x = tc.tx0 + it.px;
            const bool ok = it.live && y < H && x < W;
            const size_t o = ((size_t)(tc.b * H + min(y, H - 1)) * W + min(x, W - 1)) * a.cs_out + it.c8 * 8;
            dr[k] = *reinterpret_cast<const f16x8 *>(a.dA + o);
            if (BNB) dz[k] = *reinterpret_cast<const f16x8 *>(a.dA_z + o);
            vd |= (ok ? 1u : 0u) << k;
        }
    };

    const bool wave_live = fi < nfi;
    const bool do_bias = wave_live && (cit0 + fi == 0);
    int tile = blockIdx.x;
    const int nbx = gridDim.x;
    issue(tile < n_tiles ? tile : n_tiles - 1);
    __syncthreads();                                 // affine / coefficient tables visible
    while (tile < n_tiles) {
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const Item it = x_item(k);
            if (it.lds >= 0) {
                f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (vx & (1u << k)) v = raw_transform<LM>(xr[k], s_aff, a.x.cs_in, it.c8, a.x.cin, a.x.u8_div);
                *reinterpret_cast<f16x8 *>(s_x + it.lds) = v;
            }
        }
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            const Item it = d_item(k);
            f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (vd & (1u << k)) {
                v = dr[k];
                if (BNB) {
                    const float *A = s_coef + it.c8 * 8, *Bc = A + a.cs_out, *Cc = Bc + a.cs_out;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float zf = (float)dz[k][j];
                        v[j] = zf > 0.f ? (f16)(A[j] * (float)dr[k][j] + Bc[j] * zf + Cc[j]) : (f16)0.f;
                    }
                }
            }
            *reinterpret_cast<f16x8 *>(s_d + it.lds) = v;
        }
        __syncthreads();
        const int next = tile + nbx;
        issue(next < n_tiles ? next : tile);         // in flight during the MFMAs below (the last one re-reads this tile)
        if (wave_live) {
#pragma unroll
            for (int it = 0; it < KS / KP; ++it) {
                const int kk = kp + it * KP;
                // k-slot <-> pixel map of wgrad_mfma_kernel: lane group g's elements 0-3 are pixels 4 (g & 1) + 0..3 of row
                // 2 kk + (g >> 1), elements 4-7 the pixels 8 further right
                const int row = 2 * kk + (g >> 1);
                const int xx = 4 * (g & 1) + qq;
                f16x8 bf[4];
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const f16 *pb = s_d + ((o * NPD + row * 16 + xx) * H16 + 4 * pp);
                    const h4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(LDS_PTR(h4, pb));
                    const h4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(LDS_PTR(h4, pb + 8 * H16));
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bf[o][e] = (f16)b0[e]; bf[o][4 + e] = (f16)b1[e]; }
                }
                const f16 *pa = s_x + ((fi * NPX + row * WT + xx) * H16 + 4 * pp);
#pragma unroll
                for (int tap = 0; tap < T; ++tap) {
                    const int ty = KS3 ? tap / 3 : 0, tx = KS3 ? tap % 3 : 0;
                    const f16 *p = pa + (ty * WT + tx) * H16;
                    const h4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(LDS_PTR(h4, p));
                    const h4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(LDS_PTR(h4, p + 8 * H16));
                    f16x8 af;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { af[e] = (f16)a0[e]; af[4 + e] = (f16)a1[e]; }
#pragma unroll
                    for (int o = 0; o < 4; ++o)
                        if (o < nfo) acc[o][tap] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[o], acc[o][tap], 0, 0, 0);   // uniform
                }
                if (do_bias) {                       // column sums of dA -> the bias gradient: A = ones in row o
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        f16x8 e;
#pragma unroll
                        for (int j = 0; j < 8; ++j) e[j] = (f16)(i16 == o ? 1.0f : 0.0f);
                        accb = __builtin_amdgcn_mfma_f32_16x16x32_f16(e, bf[o], accb, 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();                             // tile reads done before the next tile overwrites LDS
        tile = next;
    }
    if (!wave_live) return;
    const int n_pairs = gm.cit_n * gm.cot_n;
    const size_t split = (size_t)blockIdx.x * KP + kp;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        if (o < nfo) {
            float *dst = a.partial + ((split * n_pairs + (size_t)(cit0 + fi) * gm.cot_n + cot0 + o) * (T + 1)) * 256 + lane;
#pragma unroll
            for (int tap = 0; tap < T; ++tap)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[tap * 256 + r * 64] = acc[o][tap][r];
            if (cit0 + fi == 0) {                    // row 0 of the block = lanes 0-15 of register 0 (wgf_stage2_kernel)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[T * 256 + r * 64] = (r == 0 && lane < 16) ? accb[o] : 0.f;
            }
        }
    }
}

inline int odd_up(int v) { return v | 1; }

bool gemm_env_on() {
    static const bool on = []() { const char *e = getenv("IMK_CONV_GEMM"); return !(e && e[0] == '0'); }();
    return on;
}

int plan_conv_gemm(const ImkConvArgs &a, GemmGeom &gm, int &pn, size_t &lds, int &grid) {
    const int nc8 = a.x.cs_in / 8;
    const int nc8p = imk_pass_chunks(nc8), n_pass = imk_cdiv_d(nc8, nc8p);
    const bool ks3 = a.ksize == 3;
    const int T = ks3 ? 9 : 1;
    const int nsp = (T * nc8p + 3) / 4;
    const int spp = ks3 ? 1 : (8 / nc8p < n_pass ? 8 / nc8p : n_pass);     // 1x1: up to 8 chunks per stage
    const int cps = spp * nc8p;
    const int mt_total = (a.cout + 15) / 16;
    // output channels per workgroup: 128 or 64, whichever pads less (ties: the larger tile, one read of the input fewer)
    const int g128 = imk_cdiv_d(mt_total, 8), g64 = imk_cdiv_d(mt_total, 4);
    pn = (g128 * 8 <= g64 * 4) ? 4 : 2;
    const int bn = 16 * pn * G_WN;
    const int halo = ks3 ? 1 : 0;
    gm.tiles_x = imk_cdiv(a.W, TW); gm.tiles_y = imk_cdiv(a.H, G_TH);
    gm.n_sp = a.B * gm.tiles_x * gm.tiles_y;
    gm.gy = imk_cdiv_d(mt_total, bn / 16);
    gm.mt_total = mt_total;
    gm.nc8 = nc8; gm.nc8p = nc8p; gm.n_pass = n_pass; gm.nsp = nsp; gm.ns_total = n_pass * nsp;
    gm.spp = spp; gm.cps = cps; gm.n_stage = imk_cdiv_d(n_pass, spp); gm.ps = odd_up(cps);
    const size_t tile_bytes = (size_t)(G_TH + 2 * halo) * (TW + 2 * halo) * gm.ps * 16;
    lds = 2 * tile_bytes + 4 * (size_t)a.x.cs_in * sizeof(float);
    const size_t out_bytes = (size_t)G_BM * (bn + 8) * sizeof(f16);
    const size_t red_bytes = 2 * (size_t)256 * 8 * sizeof(float);
    if (lds < out_bytes) lds = out_bytes;
    if (lds < red_bytes) lds = red_bytes;
    if (lds > 64 * 1024) return IMK_EUNSUPPORTED;
    grid = imk_cdiv_d(gm.n_sp, 8) * 8 * gm.gy;
    return IMK_OK;
}

template <int LM, bool KS3>
int launch_conv_gemm_k(const ImkConvArgs &a, const GemmGeom &gm, int pn, size_t lds, int grid, hipStream_t stream) {
    if (pn == 4) conv_gemm_kernel<LM, KS3, 4><<<grid, 256, lds, stream>>>(a, gm);
    else conv_gemm_kernel<LM, KS3, 2><<<grid, 256, lds, stream>>>(a, gm);
    return IMK_OK;
}

}  // namespace

// Which launches the GEMM-class kernel takes: more than 32 channels on a side (the per-tile kernel's layers), plain
// (unchained) convs whose input is a fp16 tensor.
bool imk_conv_gemm_ok(const ImkConvArgs &a) {
    if (!gemm_env_on()) return false;
    if (a.wpk2 || a.pre_wpk || a.wg_partial) return false;
    if (a.x.cs_in <= 32 && a.cout <= 32) return false;
    if (a.x.cs_in > 512) return false;
    switch (a.x.lmode) {
        case LM_RAW: case LM_AFFINE: case LM_POOL: case LM_UPADD: case LM_BNBWD: break;
        default: return false;
    }
    if (a.epi == EP_RELU) { if (a.x.lmode == LM_BNBWD) return false; }
    else if (a.x.lmode != LM_RAW && a.x.lmode != LM_BNBWD) return false;
    return (long long)a.H * a.W < imk_conv_max_pixels() && a.W < (1 << 16);
}

int imk_conv_gemm_num_tiles(int B, int H, int W) { return B * imk_cdiv(H, G_TH) * imk_cdiv(W, TW); }

int imk_launch_conv_gemm(const ImkConvArgs &a, hipStream_t stream) {
    GemmGeom gm{};
    int pn = 4, grid = 0;
    size_t lds = 0;
    int rc = plan_conv_gemm(a, gm, pn, lds, grid);
    if (rc) return rc;
    ImkProfScope prof(PF_CONV_GEMM, imk_conv_algorithmic_bytes(a), stream, imk_conv_flops(a));
    const bool ks3 = a.ksize == 3;
#define IMK_GEMM_LM(LM) (ks3 ? launch_conv_gemm_k<LM, true>(a, gm, pn, lds, grid, stream) : launch_conv_gemm_k<LM, false>(a, gm, pn, lds, grid, stream))
    switch (a.x.lmode) {
        case LM_RAW: rc = IMK_GEMM_LM(LM_RAW); break;
        case LM_AFFINE: rc = IMK_GEMM_LM(LM_AFFINE); break;
        case LM_POOL: rc = IMK_GEMM_LM(LM_POOL); break;
        case LM_UPADD: rc = IMK_GEMM_LM(LM_UPADD); break;
        case LM_BNBWD: rc = IMK_GEMM_LM(LM_BNBWD); break;
        default: return IMK_EUNSUPPORTED;
    }
#undef IMK_GEMM_LM
    if (rc) return rc;
    IMK_LAUNCH_CHECK();
    if (a.stats_rows) *a.stats_rows = gm.n_sp;
    return IMK_OK;
}

// ---- weight gradient: launch geometry ----------------------------------------------------------------------------------
namespace {
struct WgGemmPlan { WgGemmGeom gm; int nfi_t, kp, n_split; size_t lds; };

bool wgemm_env_on() {
    static const bool on = []() { const char *e = getenv("IMK_WGRAD_GEMM"); return !(e && e[0] == '0'); }();
    return on && gemm_env_on();
}

void plan_wgrad_gemm(int lmode, int B, int H, int W, int ksize, int cs_in, int cs_out, WgGemmPlan &P) {
    WgGemmGeom &gm = P.gm;
    gm.cit_n = (cs_in + 15) / 16; gm.cot_n = (cs_out + 15) / 16;
    gm.nci = cs_in / 8; gm.nco = cs_out / 8;
    // pooling on load holds 4 raw chunks per staged item: with 4 input-channel tiles per workgroup the 3x3 form spills inside
    // its MFMA loop (72 scratch operations), with 2 it does not
    const int fi_max = (lmode == LM_POOL && ksize == 3) ? 2 : 4;
    gm.gi_n = imk_cdiv_d(gm.cit_n, fi_max); gm.fi_per = imk_cdiv_d(gm.cit_n, gm.gi_n);
    gm.go_n = imk_cdiv_d(gm.cot_n, 4); gm.fo_per = imk_cdiv_d(gm.cot_n, gm.go_n);
    P.nfi_t = gm.fi_per == 1 ? 1 : (gm.fi_per == 2 ? 2 : 4);
    P.kp = 4 / P.nfi_t;
    const int tr = P.nfi_t == 1 ? 8 : 4;
    const int halo = ksize == 3 ? 1 : 0;
    gm.tiles_x = imk_cdiv(W, TW); gm.tiles_y = imk_cdiv(H, tr);
    gm.n_tiles = B * gm.tiles_x * gm.tiles_y;
    static const int target = []() { const char *e = getenv("IMK_WGRAD_GEMM_WGS"); return e ? atoi(e) : 512; }();
    // Every workgroup ends by writing its accumulators (148 KB at 64 x 64 channels x 9 taps): with few tiles per workgroup those
    // partials -- and the split reduction that reads them back -- outweigh the operands, so a workgroup gets at least
    // IMK_WGRAD_GEMM_TILES pixel tiles (the deep levels then run on fewer workgroups than the chip has slots: they are short)
    static const int min_tiles = []() { const char *e = getenv("IMK_WGRAD_GEMM_TILES"); return e ? atoi(e) : 8; }();
    int ns = target / (gm.gi_n * gm.go_n);
    const int mt = min_tiles * 64 / (tr * 16) > 1 ? min_tiles * 64 / (tr * 16) : 1;       // counted in 64-pixel tiles
    if (ns > gm.n_tiles / mt) ns = gm.n_tiles / mt;
    if (ns < 1) ns = 1;
    P.n_split = ns;
    const size_t npx = (size_t)(tr + 2 * halo) * (TW + 2 * halo), npd = (size_t)tr * 16;
    P.lds = (P.nfi_t * npx + 4 * npd) * WG_STRIDE_H * sizeof(f16) + (4 * (size_t)cs_in + 3 * (size_t)cs_out) * sizeof(float);
}

// the (load mode, BatchNorm backward on the gradient operand, kernel size) combinations the networks produce
bool wgrad_gemm_combo(int lmode, bool bnb, int ksize) {
    if (ksize == 3) return !bnb && (lmode == LM_POOL || lmode == LM_AFFINE || lmode == LM_RAW);
    return (bnb && (lmode == LM_RAW || lmode == LM_UPADD)) || (!bnb && lmode == LM_AFFINE);
}

template <int LM, bool BNB, bool KS3>
int launch_wgrad_gemm_k(const ImkWgradArgs &a, const WgGemmPlan &P, hipStream_t stream) {
    const dim3 grid(P.n_split, P.gm.gi_n * P.gm.go_n);
    if (P.nfi_t == 4) wgrad_gemm_kernel<LM, BNB, KS3, 4><<<grid, 256, P.lds, stream>>>(a, P.gm);
    else if (P.nfi_t == 2) wgrad_gemm_kernel<LM, BNB, KS3, 2><<<grid, 256, P.lds, stream>>>(a, P.gm);
    else wgrad_gemm_kernel<LM, BNB, KS3, 1><<<grid, 256, P.lds, stream>>>(a, P.gm);
    return IMK_OK;
}
}  // namespace

// wide layers: more than 32 channels on a side (the forward's rule), and a combination that is instantiated
// Which layers: more than 32 channels on a side; exactly 32 only with >= 2 M pixels (full resolution at alpha = 2: Cityscapes
// step 6.20 -> 5.98 ms; at half resolution -- alpha = 1 -- the 16 x 16-channel kernel of imk_conv.hip is faster: 2.64 vs 2.69 ms).
// IMK_WGRAD_GEMM_MIN overrides the channel threshold for every size.
bool imk_wgrad_gemm_wide(int cs_in, int cs_out, long long pixels) {
    static const int v = []() { const char *e = getenv("IMK_WGRAD_GEMM_MIN"); return e ? atoi(e) : 0; }();
    const int c = cs_in > cs_out ? cs_in : cs_out;
    if (v > 0) return c >= v;
    return c > 32 || (c == 32 && pixels >= (2ll << 20));
}
bool imk_wgrad_gemm_ok(int lmode, bool bnb, int ksize, int cs_in, int cs_out, long long pixels) {
    if (!wgemm_env_on()) return false;
    if (!imk_wgrad_gemm_wide(cs_in, cs_out, pixels)) return false;
    // Few 16 x 16 channel pairs AND few pixels (the deep levels at alpha = 0.5): the per-pair kernel of imk_conv.hip already
    // spreads such a layer over 768 workgroups with little re-reading, and this one would run on 16-64 (ISIC step 1.014 vs 1.050 ms);
    // with many pairs (alpha >= 1: 64-512 pairs at the same pixel counts) it re-reads both operands per pair and this kernel wins
    static const int min_pairs = []() { const char *e = getenv("IMK_WGRAD_GEMM_PAIRS"); return e ? atoi(e) : 17; }();
    static const long long min_pix = []() { const char *e = getenv("IMK_WGRAD_GEMM_PIX"); return e ? atoll(e) : 100000; }();
    if (((cs_in + 15) / 16) * ((cs_out + 15) / 16) < min_pairs && pixels < min_pix) return false;
    return wgrad_gemm_combo(lmode, bnb, ksize);
}

// split count the finalize job must be told (partial rows per (input tile, output tile, tap)), and the partial buffer's size
int imk_wgrad_gemm_splits(int lmode, int B, int H, int W, int ksize, int cs_in, int cs_out) {
    WgGemmPlan P{};
    plan_wgrad_gemm(lmode, B, H, W, ksize, cs_in, cs_out, P);
    return P.n_split * P.kp;
}

int imk_launch_wgrad_gemm(const ImkWgradArgs &a, hipStream_t stream) {
    WgGemmPlan P{};
    plan_wgrad_gemm(a.x.lmode, a.B, a.H, a.W, a.ksize, a.x.cs_in, a.cs_out, P);
    if (P.n_split * P.kp != a.n_split) return IMK_EINVAL;       // the caller sized the partial buffer / the finalize job with it
    if (P.lds > 64 * 1024) return IMK_EUNSUPPORTED;
    const bool bnb = a.dA_z != nullptr;
    const double px = (double)a.B * a.H * a.W;
    double in_b;
    switch (a.x.lmode) {
        case LM_POOL: in_b = 4.0 * px * a.x.cs_in * 2; break;
        case LM_UPADD: in_b = 1.25 * px * a.x.cs_in * 2; break;
        default: in_b = px * a.x.cs_in * 2;
    }
    const int T = a.ksize == 3 ? 9 : 1;
    const double bytes = in_b + px * a.cs_out * 2 * (bnb ? 2 : 1) + (double)a.n_split * P.gm.cit_n * P.gm.cot_n * (T + 1) * 1024;
    ImkProfScope prof(PF_WGRAD_GEMM, bytes, stream, imk_wgrad_flops(a));
    int rc = IMK_EUNSUPPORTED;
    if (a.ksize == 3 && !bnb) {
        if (a.x.lmode == LM_POOL) rc = launch_wgrad_gemm_k<LM_POOL, false, true>(a, P, stream);
        else if (a.x.lmode == LM_AFFINE) rc = launch_wgrad_gemm_k<LM_AFFINE, false, true>(a, P, stream);
        else if (a.x.lmode == LM_RAW) rc = launch_wgrad_gemm_k<LM_RAW, false, true>(a, P, stream);
    } else if (a.ksize == 1) {
        if (bnb && a.x.lmode == LM_RAW) rc = launch_wgrad_gemm_k<LM_RAW, true, false>(a, P, stream);
        else if (bnb && a.x.lmode == LM_UPADD) rc = launch_wgrad_gemm_k<LM_UPADD, true, false>(a, P, stream);
        else if (!bnb && a.x.lmode == LM_AFFINE) rc = launch_wgrad_gemm_k<LM_AFFINE, false, false>(a, P, stream);
    }
    if (rc) return rc;
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}
