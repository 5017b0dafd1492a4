// GEMM-class weight-gradient kernel for the WIDE layers (the forward / dgrad counterpart and the design notes are in
// imk_gemm.hip; a translation unit of its own so that the two compile side by side).
//
//   dW[tap][ci][co] = sum_pixels X[pixel + tap][ci] * dA[pixel][co]
//     * workgroup = up to 64 input x 64 output channels x all taps, a wave = one 16-channel input tile x 4 output tiles x 9
//       taps (36 accumulator fragments): x and dA are read once per 64 x 64 channel block instead of once per 16 x 16;
//     * no cross-wave reduction: every wave owns its (input tile, output tile) pairs and writes its partial rows in the
//       layout the split reduction of imk_conv.hip (wgf_stage1 / 2) already reads.
//
// Reference: the Conv2D kernel / bias gradients inside model.fit (functions.py:218) of unet.py:11-43 and evalnet.py:8-21.
#include <cstdlib>
#include "imk_stage.h"

IMK_STAMP_TABLE(wgemm)

namespace {

bool gemm_env_on() {
    static const bool on = []() { const char *e = getenv("IMK_CONV_GEMM"); return !(e && e[0] == '0'); }();
    return on;
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

// NFO: output-channel tiles a wave accumulates (4; 2 for layers with at most 32 output channels -- the full-resolution 3x3 layers
// of alpha in (1, 2] and EvalNet's towers: half the accumulators, 3 workgroups per CU instead of 2)
template <int LM, bool BNB, bool KS3, int NFI, int NFO = 4>
__global__ __launch_bounds__(256, NFO == 2 ? 3 : 2) void wgrad_gemm_kernel(ImkWgradArgs a, WgGemmGeom gm) {
    IMK_STAMP_BEGIN(wgemm, 80000 + LM * 10 + (KS3 ? 1 : 0));
    constexpr int KP = 4 / NFI;                      // waves along the k-steps
    constexpr int TR = NFI == 1 ? 8 : 4;             // tile rows (8 rows for the 1x1 forms too: measured slower, 5.77 vs 5.65 ms)
    constexpr int KS = TR / 2;                       // k-steps per tile
    constexpr int halo = KS3 ? 1 : 0, T = KS3 ? 9 : 1;
    constexpr int HT = TR + 2 * halo, WT = TW + 2 * halo;
    constexpr int NPX = HT * WT, NPD = TR * 16;      // pixels of an x slice / a dA slice
    constexpr int DCH = NFO == 2 ? 4 : 8, DSH = NFO == 2 ? 2 : 3;   // dA chunks per pixel: the two-output-tile form has at most 4
    constexpr int NX = (NPX * 2 * NFI + 255) / 256, ND = (NPD * DCH + 255) / 256;
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

    f32x4 acc[NFO][T], accb = f32x4{0, 0, 0, 0};     // accb: row o = column sums of dA over output tile o (the bias gradient)
#pragma unroll
    for (int o = 0; o < NFO; ++o)
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
        const int i = t + 256 * k;                   // NPD * DCH is a multiple of 256: no idle slots
        const int pix = i >> DSH, j = i & (DCH - 1);
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
                f16x8 bf[NFO];
#pragma unroll
                for (int o = 0; o < NFO; ++o) {
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
                    for (int o = 0; o < NFO; ++o)
                        if (o < nfo) acc[o][tap] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[o], acc[o][tap], 0, 0, 0);   // uniform
                }
                if (do_bias) {                       // column sums of dA -> the bias gradient: A = ones in row o
#pragma unroll
                    for (int o = 0; o < NFO; ++o) {
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
    for (int o = 0; o < NFO; ++o) {
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
    IMK_STAMP_END(1);
}

}  // namespace

// ---- weight gradient: launch geometry ----------------------------------------------------------------------------------
namespace {
struct WgGemmPlan { WgGemmGeom gm; int nfi_t, kp, n_split; size_t lds; };

bool wgemm_env_on() {
    static const bool on = []() { const char *e = getenv("IMK_WGRAD_GEMM"); return !(e && e[0] == '0'); }();
    return on && gemm_env_on();
}

bool wgrad_nfo2_on() {
    static const bool on = []() { const char *e = getenv("IMK_WGRAD_NFO2"); return !(e && e[0] == '0'); }();
    return on;
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
    // the two-output-tile form of the 3x3 (wgrad_gemm_kernel<.., 2, 2>) runs 3 workgroups per compute unit
    const bool nfo2 = wgrad_nfo2_on() && lmode == LM_AFFINE && ksize == 3 && P.nfi_t == 2 && gm.cot_n <= 2;
    int ns = (nfo2 ? target * 3 / 2 : target) / (gm.gi_n * gm.go_n);
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
    if constexpr (LM == LM_AFFINE && KS3 && !BNB) {
        if (wgrad_nfo2_on() && P.nfi_t == 2 && P.gm.cot_n <= 2) {
            imk_klaunch(wgrad_gemm_kernel<LM, BNB, KS3, 2, 2>, dim3(grid), dim3(256), P.lds, stream, a, P.gm);
            return IMK_OK;
        }
    }
    if (P.nfi_t == 4) imk_klaunch(wgrad_gemm_kernel<LM, BNB, KS3, 4>, dim3(grid), dim3(256), P.lds, stream, a, P.gm);
    else if (P.nfi_t == 2) imk_klaunch(wgrad_gemm_kernel<LM, BNB, KS3, 2>, dim3(grid), dim3(256), P.lds, stream, a, P.gm);
    else imk_klaunch(wgrad_gemm_kernel<LM, BNB, KS3, 1>, dim3(grid), dim3(256), P.lds, stream, a, P.gm);
    return IMK_OK;
}
}  // namespace

// wide layers: more than 32 channels on a side (the forward's rule), and a combination that is instantiated
// Which layers: more than 32 channels on a side; 24-32 only with >= 2 M pixels (full resolution at alpha = 2: Cityscapes
// step 6.20 -> 5.98 ms; at half resolution -- alpha = 1 -- the 16 x 16-channel kernel of imk_conv.hip is faster: 2.64 vs 2.69 ms).
// IMK_WGRAD_GEMM_MIN overrides the channel threshold for every size.
bool imk_wgrad_gemm_wide(int cs_in, int cs_out, long long pixels) {
    static const int v = []() { const char *e = getenv("IMK_WGRAD_GEMM_MIN"); return e ? atoi(e) : 0; }();
    const int c = cs_in > cs_out ? cs_in : cs_out;
    if (v > 0) return c >= v;
    return c > 32 || (c >= 24 && pixels >= (2ll << 20));     // (24 ... 31: alpha 1.25 / 1.5 at full resolution: step 4.11 -> 4.05 / 4.29 -> 4.20 ms)
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
