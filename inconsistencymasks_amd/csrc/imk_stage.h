// Input staging shared by the conv kernels (imk_conv.hip: per-tile / persistent kernels; imk_gemm.hip: the GEMM-class
// kernels for wide layers): how a tile of the conv's input is materialised from its source tensor(s) -- the producer's
// BatchNorm, max-pool, upsample + add, x/255 or the BatchNorm backward applied on load (ImkLoadMode) -- and the
// (scalar tile base) + (32-bit lane offset) address arithmetic.
#pragma once
#include <cstdlib>
#include "imk_kernels.h"
#include "imk_elem.h"

typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 h4;
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T *)(p))

namespace {

constexpr int TW = 16;  // tile width = one MFMA pixel group per tile row
constexpr int WG_STRIDE_H = 16;  // halfs per pixel in the wgrad LDS slices: 16 channels = 32 B, unpadded on purpose -- a 32-lane half
                                 // of a transposed read covers 8 CONSECUTIVE pixels = one contiguous 256-B bank row (conflict-free)
constexpr int WG_RED_CHUNK = 16; // splits summed per stage-1 chunk of the weight-gradient reduction (wgf_stage1_kernel)

// Pixel pitch (in 16-byte chunks) of an LDS tile [pixel][nc8 chunks] that is read as the MFMA B operand with ds_read_b128, lane
// (pixel n = lane & 15, group g = lane >> 4) taking chunk f(g) of pixel n (+ a uniform tap shift).  The hardware serves a
// ds_read_b128 in four groups of 16 lanes that are NOT lanes 0-15, 16-31, ... but {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and
// the same + 32 (MI355X_MICROARCH.md, LDS): a group holds pixels 0-3 and 12-15 with one chunk and pixels 4-11 with the NEXT
// chunk.  With the odd pitches of rounds 1-3 ("conflict-free across the 16 pixels of a group" -- true for contiguous groups)
// every such read of a tile with 2-4 chunks per pixel took 8 LDS cycles instead of 4 (three slots of every group two-way;
// SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS 1.0-2.6 on conv_gemm / conv_wide / the 16-channel conv_pipe variants,
// profiles/r03_sq_counters_*.csv), and with two workgroups per CU the LDS array, not the matrix cores, set the pace of the wide
// layers.  Enumerating the groups (tests/gpu_probe/lds_pitch_model.py): a pitch of 2 mod 4 chunks is conflict-free for 2 and 4
// chunks per pixel (2 -> 2, 4 -> 6, 8 -> 10) and the best choice for 3 (6); one chunk per pixel keeps pitch 1.
__host__ __device__ constexpr int imk_lds_pitch(int nc8) { return nc8 <= 1 ? 1 : ((nc8 + 1) & ~3) + 2; }

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

// Chunks (of 8 channels) per K pass.  The per-tile kernel walks the input channels in passes of at most IMK_PASS_CAP
// chunks, each pass staging its channel slice of the tile and of the packed weights into LDS (k order = pass, tap,
// chunk in pass).  Short passes keep the LDS footprint near 60 KB whatever the layer width, so 2-3 workgroups share a
// CU and one's staging overlaps another's MFMAs: measured per inference call of 128 images (alpha = 1 nets) with caps
// 32 / 8 / 4 / 2: SUIM 1.97 / 1.74 / 1.64 / 1.82 ms, Cityscapes 3.68 / - / 2.93 / 3.17 ms; ISIC (alpha 0.5) 0.71 / 0.71 /
// 0.70 / 0.73 ms.
__host__ __device__ inline int imk_cdiv_d(int a, int b) { return (a + b - 1) / b; }
#ifndef IMK_PASS_CAP
#define IMK_PASS_CAP 4
#endif
__host__ __device__ inline int imk_pass_chunks(int nc8) {
    if (nc8 <= IMK_PASS_CAP) return nc8;
    const int n_pass = imk_cdiv_d(nc8, IMK_PASS_CAP);
    return imk_cdiv_d(nc8, n_pass);
}

// n / d for n < 2^23 via a float reciprocal (exact after one correction step): ~6 VALU ops instead of ~35
__device__ __forceinline__ unsigned fast_div(unsigned n, unsigned d, float inv_d) {
    unsigned q = (unsigned)((float)n * inv_d);
    const int r = (int)n - (int)(q * d);
    if (r < 0) --q; else if (r >= (int)d) ++q;
    return q;
}

__device__ __forceinline__ TileCoord tile_coord_fast(int tile, int tiles_x, int per_img, float inv_tx, float inv_pi) {
    TileCoord c;
    c.b = (int)fast_div((unsigned)tile, (unsigned)per_img, inv_pi);
    const int r = tile - c.b * per_img;
    const int ty = (int)fast_div((unsigned)r, (unsigned)tiles_x, inv_tx);
    c.ty0 = ty * 16;
    c.tx0 = (r - ty * tiles_x) * TW;
    return c;
}

__device__ __forceinline__ f16x8 affine8(f16x8 z, const float *sc, const float *sh) { return imk_affine8(z, sc, sh); }   // imk_common.h

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
            const int H2 = in.src_h ? in.src_h : 2 * H, W2 = in.src_w ? in.src_w : 2 * W;
            const f16 *p = (const f16 *)in.in + ((size_t)(b * H2 + 2 * y) * W2 + 2 * x) * cs + c8 * 8;
            const f16x8 z00 = *(const f16x8 *)p, z01 = *(const f16x8 *)(p + cs);
            const f16x8 z10 = *(const f16x8 *)(p + (size_t)W2 * cs), z11 = *(const f16x8 *)(p + (size_t)W2 * cs + cs);
            return imk_affine_pool8(z00, z01, z10, z11, s_aff + c8 * 8, s_aff + cs + c8 * 8);
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
        case LM_BNBWD: {
            const size_t o = ((size_t)(b * H + y) * W + x) * cs + c8 * 8;
            const f16x8 dy = *(const f16x8 *)((const f16 *)in.in + o), z = *(const f16x8 *)((const f16 *)in.in2 + o);
            const float *A = s_aff + c8 * 8, *Bc = s_aff + cs + c8 * 8, *Cc = s_aff + 2 * cs + c8 * 8;
            f16x8 o8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float zf = (float)z[j];
                o8[j] = zf > 0.f ? (f16)(A[j] * (float)dy[j] + Bc[j] * zf + Cc[j]) : (f16)0.f;
            }
            return o8;
        }
        default: {  // LM_U8: cin <= 8 bytes per pixel, single chunk
            const uint8_t *p = (const uint8_t *)in.in + ((size_t)(b * H + y) * W + x) * in.cin;
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (f16)(j < in.cin ? (float)p[j] / in.u8_div : 0.0f);
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
    if (in.lmode == LM_BNBWD)
        for (int i = threadIdx.x; i < 3 * cs; i += 256) s_aff[i] = in.sc[i];
    if (in.lmode == LM_STEM) {   // [sc | sh | W[0] | W[1] | W[2] | W[3] | bias], cs floats each; weights as the fp16 values the stem kernel uses
        for (int i = threadIdx.x; i < cs; i += 256) {
            const bool real = i < in.cin;
            s_aff[i] = in.sc[i]; s_aff[cs + i] = in.sh[i];
            for (int c = 0; c < 4; ++c) s_aff[(2 + c) * cs + i] = (real && c < in.u8_c) ? (float)(f16)in.sc2[c * in.cin + i] : 0.f;
            s_aff[6 * cs + i] = real ? in.sh2[i] : 0.f;
        }
    }
}

// ---- two-phase input staging: raw global loads into registers, transform later ------------------------------------
template <int LM> struct RawChunk { f16x8 v[LM == LM_POOL ? 4 : ((LM == LM_UPADD || LM == LM_BNBWD) ? 2 : 1)]; };
template <> struct RawChunk<LM_U8> { uint32_t b[4]; };
template <> struct RawChunk<LM_STEM> { uint32_t b[4]; };

template <int LM>
__device__ __forceinline__ void raw_load(const ImkInput &in, int b, int y, int x, int H, int W, int c8, RawChunk<LM> &r) {
    const int cs = in.cs_in;
    if constexpr (LM == LM_RAW || LM == LM_AFFINE) {
        r.v[0] = *(const f16x8 *)((const f16 *)in.in + ((size_t)(b * H + y) * W + x) * cs + c8 * 8);
    } else if constexpr (LM == LM_POOL) {
        const int H2 = in.src_h ? in.src_h : 2 * H, W2 = in.src_w ? in.src_w : 2 * W;
        const f16 *p = (const f16 *)in.in + ((size_t)(b * H2 + 2 * y) * W2 + 2 * x) * cs + c8 * 8;
        r.v[0] = *(const f16x8 *)p;
        r.v[1] = *(const f16x8 *)(p + cs);
        r.v[2] = *(const f16x8 *)(p + (size_t)W2 * cs);
        r.v[3] = *(const f16x8 *)(p + (size_t)W2 * cs + cs);
    } else if constexpr (LM == LM_UPADD) {
        r.v[0] = *(const f16x8 *)((const f16 *)in.in + ((size_t)(b * (H / 2) + (y >> 1)) * (W / 2) + (x >> 1)) * cs + c8 * 8);
        r.v[1] = *(const f16x8 *)((const f16 *)in.in2 + ((size_t)(b * H + y) * W + x) * cs + c8 * 8);
    } else if constexpr (LM == LM_BNBWD) {
        const size_t o = ((size_t)(b * H + y) * W + x) * cs + c8 * 8;
        r.v[0] = *(const f16x8 *)((const f16 *)in.in + o);
        r.v[1] = *(const f16x8 *)((const f16 *)in.in2 + o);
    } else {
        const int nb = LM == LM_STEM ? in.u8_c : in.cin;     // bytes per pixel
        const uint8_t *p = (const uint8_t *)in.in + ((size_t)(b * H + y) * W + x) * nb;
        // One unaligned dword instead of nb byte loads (RGB images: the byte loads made the stem's weight gradient
        // issue-bound), branch-free like every load of the prefetches: the last pixels of an image take their dword a few
        // bytes early, so nothing past the image is read (H * W * nb >= 4).
        typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
        const int rem = (H * W - (y * W + x)) * nb;          // bytes from this pixel to the end of its image
        const int back = rem >= 4 ? 0 : 4 - rem;
        const uint32_t v = *reinterpret_cast<const u32_unaligned *>(p - back) >> (8 * back);
#pragma unroll
        for (int j = 0; j < 4; ++j) r.b[j] = (j < nb) ? ((v >> (8 * j)) & 0xffu) : 0u;
    }
}

template <int LM>
__device__ __forceinline__ f16x8 raw_transform(const RawChunk<LM> &r, const float *s_aff, int cs, int c8, int cin,
                                               float u8_div, int u8_c = 4) {
    if constexpr (LM == LM_RAW) {
        return r.v[0];
    } else if constexpr (LM == LM_AFFINE) {
        return affine8(r.v[0], s_aff + c8 * 8, s_aff + cs + c8 * 8);
    } else if constexpr (LM == LM_POOL) {
        return imk_affine_pool8(r.v[0], r.v[1], r.v[2], r.v[3], s_aff + c8 * 8, s_aff + cs + c8 * 8);
    } else if constexpr (LM == LM_UPADD) {
        const f16x8 lo = affine8(r.v[0], s_aff + c8 * 8, s_aff + cs + c8 * 8);
        const f16x8 sk = affine8(r.v[1], s_aff + 2 * cs + c8 * 8, s_aff + 3 * cs + c8 * 8);
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)((float)lo[j] + (float)sk[j]);
        return o;
    } else if constexpr (LM == LM_BNBWD) {
        const float *A = s_aff + c8 * 8, *Bc = s_aff + cs + c8 * 8, *Cc = s_aff + 2 * cs + c8 * 8;
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float zf = (float)r.v[1][j];
            o[j] = zf > 0.f ? (f16)(A[j] * (float)r.v[0][j] + Bc[j] * zf + Cc[j]) : (f16)0.f;
        }
        return o;
    } else if constexpr (LM == LM_STEM) {
        // x * (1/255) rounds to the same fp16 as x / 255 for all 256 byte values (checked exhaustively)
        // (this transform is what bounds the inference stem: its VALU is ~85 % busy.  RGB and grey images skip the products of
        //  the absent 4th / 2nd-4th channel -- their weights are zero -- behind a uniform branch.)
        float xin[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) xin[c] = (float)(f16)((float)r.b[c] * (1.0f / 255.0f));
        const float *sc = s_aff + c8 * 8, *sh = s_aff + cs + c8 * 8, *bias = s_aff + 6 * cs + c8 * 8;
        f16x8 o;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = s_aff[2 * cs + c8 * 8 + j] * xin[0];
        if (u8_c > 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(s_aff[4 * cs + c8 * 8 + j], xin[2], fmaf(s_aff[3 * cs + c8 * 8 + j], xin[1], acc[j]));
        }
        if (u8_c > 3) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(s_aff[5 * cs + c8 * 8 + j], xin[3], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const f16x2 z2 = imk_bias_relu2(f32x2{acc[j], acc[j + 1]}, f32x2{bias[j], bias[j + 1]});               // the stem's stored output
            const f16x2 w2 = imk_affine2(z2, f32x2{sc[j], sc[j + 1]}, f32x2{sh[j], sh[j + 1]});
            o[j] = w2[0]; o[j + 1] = w2[1];
        }
        return o;
    } else {
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)((j < 4 && j < cin) ? (float)r.b[j] / u8_div : 0.0f);
        return o;
    }
}

// =====================================================================================================
// Addresses of the conv kernels = (64-bit SCALAR base of the tile) + (32-bit unsigned per-lane byte offset).
// Written as ((b * H + y) * W + x) * cs in size_t per access, every load and store of a tile loop cost a 64 x 32-bit multiply
// in vector registers (v_mad_i64_i32, 2 v_mul_lo_u32, v_mad_u64_u32: quarter-rate instructions, ~18 issue slots per access)
// and the tile coordinates two float-reciprocal divisions on uniform values -- about a third of the VALU time of kernels
// whose VALU is ~70 % busy (SQ_ACTIVE_INST_VALU, profiles/r02_sq_counters.csv).  Instead: (image, tile in image) advance by
// the grid size without a division, the tile row comes from one s_mul_hi_u32 by a host-computed magic number, the tile's
// base address of each tensor is scalar arithmetic, and a lane adds an offset relative to the tile origin -- a constant for
// the stores and epilogue loads of full tiles, one clamp per coordinate and one 24-bit multiply-add (full rate) for a
// staged halo pixel.  Limits: imk_conv_max_pixels() per image, rows < 2^16 (plans are refused above).
// =====================================================================================================
struct PTile { int b, r, ty0, tx0; };            // r: the tile's index inside image b; all four live in scalar registers
__device__ __forceinline__ PTile ptile_at(int b, int r, int tiles_x, unsigned magic_tx) {
    PTile c;
    c.b = __builtin_amdgcn_readfirstlane(b); c.r = __builtin_amdgcn_readfirstlane(r);
    const unsigned ru = (unsigned)c.r;
    const unsigned ty = tiles_x == 1 ? ru : (unsigned)(((unsigned long long)ru * magic_tx) >> 32);   // r * tiles_x < 2^32
    c.ty0 = (int)ty * 16;
    c.tx0 = (int)(ru - ty * (unsigned)tiles_x) * TW;
    return c;
}
__device__ __forceinline__ PTile ptile_next(const PTile &c, int grid_q, int grid_r, int per_img, int tiles_x, unsigned magic_tx) {
    int b = c.b + grid_q, r = c.r + grid_r;      // the grid size = grid_q images + grid_r tiles (host-computed)
    if (r >= per_img) { r -= per_img; ++b; }
    return ptile_at(b, r, tiles_x, magic_tx);
}
// Persistent walk over the tiles of a launch, XCD-aware (round 4).  Workgroups are dealt round-robin over the 8 XCDs (blocks b
// and b + 8 share one: MI355X_MICROARCH.md, "Workgroup dispatch"), each XCD has its own 4 MiB L2, and a plain walk
// (tile = block, block + grid, ...) hands neighbouring tiles to DIFFERENT XCDs -- every halo row / column of a 3x3 input was
// then fetched through the fabric by each of its 2-4 tiles (FETCH_SIZE 1.4-1.6x the algorithmic bytes on the full-resolution
// launches, profiles/r03_pmc_traffic.csv).  Here group g = block & (2^shift - 1) owns the contiguous tile range
// [g * chunk, (g + 1) * chunk) and its workgroups (block >> shift = 0 .. step - 1) sweep it together, a band of `step`
// neighbouring tiles per iteration: the halo re-reads are hits in that XCD's L2.  shift = 0 is the plain walk.  Placement is a
// matter of speed only; any block -> XCD map gives the same results.
// Dynamic form (round 4; kernels instantiated with DYN, launches that carry ImkConvArgs::sched): the workgroups of one launch
// have equal work but finish 12 ... 22 us after its start (per-workgroup clock stamps, tests/gpu_probe/wgstamps.py: the e1 / d9
// forward launch; 93 ... 123 us for the decoder's inference launch) -- the launch ends with its slowest workgroup.  There a
// workgroup's first tile is the static one and every further tile is TAKEN from its group's counter (one returning atomic per tile,
// requested two tiles ahead so that its latency never shows): fast workgroups process more tiles, all finish together.  Only for
// launches whose results do not depend on which workgroup computed which tile (no per-workgroup partial rows): inference.
struct ImkWalk { int chunk, shift, step, q, r; unsigned magic_pi; };      // q, r: step = q images + r tiles; magic_pi: / per_img (0: divide)
// group of a block and the index of its tile range: ranges of one XCD are neighbours (shift 5: range = 4 * XCD + quarter)
__device__ __forceinline__ int imk_walk_range(const ImkWalk &wk, unsigned block) {
    const int grp = (int)(block & ((1u << wk.shift) - 1u));
    return wk.shift == 5 ? (((grp & 7) << 2) | (grp >> 3)) : grp;
}
inline unsigned imk_div_magic(int d) { return d > 1 ? (unsigned)((1ull << 32) / (unsigned)d + 1) : 0u; }
// Tile counters of the dynamic form: IMK_SCHED_HEADS counters per launch, IMK_SCHED_STRIDE bytes apart.  One counter word takes
// ~90 atomics per microsecond, and counters in one 64-byte line share that (the first version, 8 counters in one line, added
// 1.6 ms to a 0.9 ms forward); 32 groups = 4 per XCD on separate lines keep the take off the critical path.
// (IMK_SCHED_HEADS / IMK_SCHED_STRIDE / IMK_SCHED_BYTES: imk_kernels.h)
inline ImkWalk imk_walk_make(int &grid, int n_tiles, int per_img, bool dyn = false) {
    static const bool off = []() { const char *e = getenv("IMK_XCD_WALK"); return e && e[0] == '0'; }();
    ImkWalk w{};
    if (dyn && grid >= 64 && n_tiles >= 64) {     // 32 groups: group & 7 = XCD, group >> 3 = quarter of that XCD's range
        grid &= ~31;
        w.shift = 5; w.chunk = (n_tiles + 31) / 32; w.step = grid / 32;
    } else if (!off && grid >= 64 && n_tiles >= 64) {
        grid &= ~7;
        w.shift = 3; w.chunk = (n_tiles + 7) / 8; w.step = grid / 8;
    } else {
        w.shift = 0; w.chunk = n_tiles; w.step = grid;
    }
    w.q = w.step / per_img; w.r = w.step % per_img;
    w.magic_pi = (per_img > 1 && (long long)n_tiles * per_img < (1ll << 32)) ? imk_div_magic(per_img) : 0u;
    return w;
}

// tile index of the launch -> (image, tile in image, origin); the index is wave-uniform
__device__ __forceinline__ PTile ptile_of_index(int tile, int per_img, int tiles_x, unsigned magic_tx, unsigned magic_pi) {
    const unsigned tu = (unsigned)__builtin_amdgcn_readfirstlane(tile);
    const unsigned b = magic_pi ? (unsigned)(((unsigned long long)tu * magic_pi) >> 32) : (per_img == 1 ? tu : tu / (unsigned)per_img);
    return ptile_at((int)b, (int)(tu - b * (unsigned)per_img), tiles_x, magic_tx);
}
// one tile ticket of counter `head` for the whole WORKGROUP: every lane issues the (bounds-checked) buffer atomic -- no branch, so
// the compiler keeps counting the loop's memory operations -- but only lane 0 of wave 0 is inside the 4-byte buffer and adds
__device__ __forceinline__ int imk_take_ticket(unsigned *head, bool chosen) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(head, 0, 4, 0x00020000);
    return __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, rs, chosen ? 0 : 0x7ffffff0, 0, 1 /* sc0: return the old value */);
}

// base of pixel (y, x) of image b in a tensor with pitch_b bytes per pixel: scalar; may point in front of the tensor (y, x = -1)
__device__ __forceinline__ const char *pix_base(const void *ptr, int b, int hh, int ww, int y, int x, unsigned pitch_b) {
    return reinterpret_cast<const char *>(ptr) + ((long long)(b * hh + y) * ww + x) * (long long)pitch_b;
}

// The staged input of one tile: scalar bases of the source tensor(s) at the halo tile's origin (-1 at the image border) and
// the window of its rows / columns that lie inside the image.
template <int LM>
struct PSrc {
    const char *b_in, *b_in2;
    unsigned row2_b, w2, csb;
    int oy, ox, oyl, oxl, pix0, nb, hw, lo_y, hi_y, lo_x, hi_x;
};
// CSB: bytes per pixel of the fp16 input tensor when the kernel knows them at compile time, 0 = in.cs_in * 2 at run time
template <int LM, unsigned CSB>
__device__ __forceinline__ PSrc<LM> psrc_of(const ImkInput &in, const PTile &tc, int halo, int HT, int WT, int H, int W) {
    PSrc<LM> s{};
    const int oy = tc.ty0 - halo, ox = tc.tx0 - halo;
    const unsigned csb = CSB ? CSB : (unsigned)in.cs_in * 2u;
    s.csb = csb;
    s.oy = oy; s.ox = ox;
    s.lo_y = oy < 0 ? -oy : 0; s.hi_y = min(HT - 1, H - 1 - oy);
    s.lo_x = ox < 0 ? -ox : 0; s.hi_x = min(WT - 1, W - 1 - ox);
    if constexpr (LM == LM_RAW || LM == LM_AFFINE || LM == LM_BNBWD) {
        s.b_in = pix_base(in.in, tc.b, H, W, oy, ox, csb);
        if constexpr (LM == LM_BNBWD) s.b_in2 = pix_base(in.in2, tc.b, H, W, oy, ox, csb);
    } else if constexpr (LM == LM_POOL) {
        const int H2 = in.src_h ? in.src_h : 2 * H, W2 = in.src_w ? in.src_w : 2 * W;
        s.b_in = pix_base(in.in, tc.b, H2, W2, 2 * oy, 2 * ox, csb);
        s.w2 = (unsigned)W2;
        s.row2_b = s.w2 * csb;
    } else if constexpr (LM == LM_UPADD) {
        s.oyl = oy >> 1; s.oxl = ox >> 1;
        s.b_in = pix_base(in.in, tc.b, H / 2, W / 2, s.oyl, s.oxl, csb);
        s.b_in2 = pix_base(in.in2, tc.b, H, W, oy, ox, csb);
    } else {                                        // uint8 pixels: nb bytes each
        s.nb = LM == LM_STEM ? in.u8_c : in.cin;
        s.hw = H * W;
        s.pix0 = oy * W + ox;                       // may be negative; pix0 + (a clamped pixel's offset) never is
        s.b_in = reinterpret_cast<const char *>(in.in) + (long long)tc.b * s.hw * s.nb;
    }
    return s;
}
// Chunk c8 of the halo tile's pixel (py, px), clamped into the image (unconditional load); returns whether it was inside.
template <int LM, unsigned CSB>
__device__ __forceinline__ bool psrc_load(const PSrc<LM> &s, int py, int px, int c8, int W, RawChunk<LM> &r) {
    const int ry = min(max(py, s.lo_y), s.hi_y), rx = min(max(px, s.lo_x), s.hi_x);
    auto bytes = [&](unsigned pixels) { return CSB ? pixels * CSB : __umul24(pixels, s.csb); };   // tile-relative: < 2^24
    if constexpr (LM == LM_RAW || LM == LM_AFFINE || LM == LM_BNBWD) {
        const unsigned o = bytes(__umul24(ry, W) + rx) + c8 * 16;
        r.v[0] = *reinterpret_cast<const f16x8 *>(s.b_in + o);
        if constexpr (LM == LM_BNBWD) r.v[1] = *reinterpret_cast<const f16x8 *>(s.b_in2 + o);
    } else if constexpr (LM == LM_POOL) {
        const unsigned o = bytes(__umul24(2 * ry, s.w2) + 2 * rx) + c8 * 16;
        r.v[0] = *reinterpret_cast<const f16x8 *>(s.b_in + o);
        r.v[1] = *reinterpret_cast<const f16x8 *>(s.b_in + o + s.csb);
        r.v[2] = *reinterpret_cast<const f16x8 *>(s.b_in + o + s.row2_b);
        r.v[3] = *reinterpret_cast<const f16x8 *>(s.b_in + o + s.row2_b + s.csb);
    } else if constexpr (LM == LM_UPADD) {
        const int yl = ((s.oy + ry) >> 1) - s.oyl, xl = ((s.ox + rx) >> 1) - s.oxl;
        r.v[0] = *reinterpret_cast<const f16x8 *>(s.b_in + bytes(__umul24(yl, W / 2) + xl) + c8 * 16);
        r.v[1] = *reinterpret_cast<const f16x8 *>(s.b_in2 + bytes(__umul24(ry, W) + rx) + c8 * 16);
    } else {
        // One unaligned dword instead of nb byte loads (RGB images: the byte loads made the stem's weight gradient
        // issue-bound), branch-free like every load of the prefetches: the last pixels of an image take their dword a few
        // bytes early, so nothing past the image is read (H * W * nb >= 4).
        typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
        const int pidx = s.pix0 + (int)(__umul24(ry, W) + rx);           // pixel index inside its image
        const int rem = __mul24(s.hw - pidx, s.nb);                       // bytes from this pixel to the end of the image
        const int back = rem >= 4 ? 0 : 4 - rem;
        const uint32_t v = *reinterpret_cast<const u32_unaligned *>(s.b_in + (unsigned)(__mul24(pidx, s.nb) - back)) >> (8 * back);
#pragma unroll
        for (int j = 0; j < 4; ++j) r.b[j] = (j < s.nb) ? ((v >> (8 * j)) & 0xffu) : 0u;
    }
    return ry == py && rx == px;
}

}  // namespace
