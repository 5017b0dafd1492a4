// Implicit-GEMM convolution kernels for the tiny U-Net on gfx950 (MI355X): forward / dgrad (one kernel,
// different packed weights) and wgrad.  NHWC fp16 activations, fp32 accumulation on the matrix cores.
//
// Forward/dgrad:  D[co][pixel] = sum_k Wp[co][k] * X[k][pixel],  k = (tap, input channel).
//   * one workgroup (4 waves) = one TH x 16 output tile of one image; the input tile (+1 halo for 3x3) is
//     staged ONCE into LDS as fp16 [pixel][channel chunk of 8] with the producer's BatchNorm / max-pool /
//     upsample+add / u8->float applied on the way in, so those layers never make their own HBM pass;
//   * the pixel operand (MFMA B) is one ds_read_b128 per lane per k-step straight from that tile: the 8
//     consecutive k of a lane are the 8 channels of one chunk of one (shifted) pixel, so im2col is free;
//     pixel stride imk_lds_pitch(chunks) (imk_stage.h) => conflict-free across the hardware's 16-lane groups of ds_read_b128;
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
#include <cstdlib>
#include "imk_kernels.h"
#include "imk_elem.h"

#include "imk_stage.h"

#ifndef IMK_ABL
#define IMK_ABL 0
#endif
#ifndef IMK_INF_WAVES
#define IMK_INF_WAVES 4      // waves per SIMD the 8-channel inference instantiations of conv_pipe_kernel are compiled for (probe builds: 5)
#endif
IMK_STAMP_TABLE(conv)
IMK_WGSTAMP_TABLE(conv)

namespace {


// =====================================================================================================
// forward / dgrad
// =====================================================================================================
// Channel passes: the K dimension is walked in n_pass passes of nc8p chunks (imk_pass_chunks above), each pass staging
// its channel slice of the tile and of the packed weights; any width up to 512 channels runs this way.
// k order = (pass, tap, chunk in pass) -- see pack_conv_batched_kernel.
struct ImkConvGeom { int tiles_x, tiles_y, mt_total, nc8, nc8p, n_pass, ps, nsp; unsigned magic_tx, magic_pi; };   // magic_pi 0: plain division
// LDS offset of a chained launch's second weight set: behind the first stage's regions and the output tile that reuses them
__host__ __device__ inline size_t imk_chain_w2_offset(size_t stage1_bytes, size_t out_bytes) {
    return ((stage1_bytes > out_bytes ? stage1_bytes : out_bytes) + 15) & ~(size_t)15;
}

// bx = spatial tile, by = group of MT output-channel tiles (the launch grid, or a slice of a fused launch's 1-D grid)
// (Measured and dropped for the launches that do not fill the chip -- deep layers at batch 32: staging batches of 6-12
// items per thread, 1.386 vs 1.369 ms per step; 1024-thread workgroups with one wave per tile row, 1.450 ms.)
template <int TH, int MT, int LM, bool CHAIN = false>
__device__ __forceinline__ void conv_mfma_body(const ImkConvArgs &a, const ImkConvGeom &gm, int bx, int by) {
    const int tiles_x = gm.tiles_x, tiles_y = gm.tiles_y, mt_total = gm.mt_total, nc8 = gm.nc8, nc8p = gm.nc8p;
    const int n_pass = gm.n_pass, ps = gm.ps, nsp = gm.nsp;
    constexpr int NW = 4, NT = 64 * NW;   // waves / threads per workgroup
    constexpr int P = TH / NW;            // pixel groups (tile rows) per wave
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int ks3 = (a.ksize == 3);
    const int halo = ks3 ? 1 : 0;
    const int T = ks3 ? 9 : 1;
    const int HT = TH + 2 * halo, WT = TW + 2 * halo;
    uint8_t *s_tile = smem;
    float *s_aff = reinterpret_cast<float *>(smem + (size_t)HT * WT * ps * 16);
    f16 *s_w = reinterpret_cast<f16 *>(s_aff + 4 * a.x.cs_in);   // [MT][nsp][512] packed weight fragments of one pass
    // CHAIN: the second conv's weight fragments [mt2][ns2][512], behind everything the first stage and its output tile use
    f16 *s_w2 = reinterpret_cast<f16 *>(smem + imk_chain_w2_offset((size_t)HT * WT * ps * 16 + 4 * (size_t)a.x.cs_in * sizeof(float) +
                                                                   (size_t)MT * nsp * 1024, (size_t)TH * 16 * (MT * 16 + 8) * sizeof(f16)));
    const int t = threadIdx.x;
    const int H = a.H, W = a.W;
    // addresses: scalar base of the tile + 32-bit lane offsets (PTile / PSrc above); TH = 16 rows per tile
    const int per_img = tiles_x * tiles_y;
    const int tb = gm.magic_pi ? (int)(((unsigned long long)(unsigned)bx * gm.magic_pi) >> 32) : bx / per_img;
    const PTile tc = ptile_at(tb, bx - tb * per_img, tiles_x, gm.magic_tx);
    const PSrc<LM> src = psrc_of<LM, 0>(a.x, tc, halo, HT, WT, H, W);
    const int ns_total = n_pass * nsp;

    IMK_STAMP_BEGIN(conv, 10000 + LM * 1000 + MT * 100 + n_pass * 10 + ks3);
    stage_affine_table(a.x, s_aff);   // visible after the barrier inside the first staging batch
    bool aff_pending = (LM != LM_RAW && LM != LM_U8);

    const int lane = t & 63, wave = t >> 6, n = lane & 15, g = lane >> 4;
    const int ct0 = by * MT;
    int base[P];
#pragma unroll
    for (int p = 0; p < P; ++p) base[p] = ((wave * P + p) * WT + n) * ps * 16;
    f32x4 acc[MT][P];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int p = 0; p < P; ++p) acc[m][p] = f32x4{0, 0, 0, 0};
    for (int pass = 0; pass < n_pass; ++pass) {
        const int c8_lo = pass * nc8p;
        const int nc8_cur = min(nc8p, nc8 - c8_lo);
        if (pass) __syncthreads();   // everyone is done with the previous pass's tile and weights
        {
            // One cooperative copy of this workgroup's weight fragments into LDS: the k-loop then never waits for L2
            // (deep layers have 18-72 k-steps; a global fragment load per k-step costs a full L2 latency each).
            // The copy is asynchronous (global_load_lds_dwordx4: global -> LDS without registers, one contiguous KB per
            // wave and instruction), so all of it is in flight while the input tile is staged below.
            // (no division in the copy loop: one (channel tile, chunk) walk per m)
            const int n16m = nsp * 64;       // 16-byte chunks per channel tile; a multiple of 64: wave-uniform bounds
            if (CHAIN && pass == 0) {        // the chained 1x1's fragments: contiguous in its pack, landed by the first barrier
                const int nc8_2 = a.cs_out / 8;
                const int n16 = ((a.cout2 + 15) / 16) * imk_cdiv_d(nc8_2, imk_pass_chunks(nc8_2)) * 64;
                for (int j = t; j < n16; j += NT)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.wpk2 + (size_t)j * 8),
                                                     (__attribute__((address_space(3))) void *)(s_w2 + (size_t)(j - lane) * 8), 16, 0, 0);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                int ct = by * MT + m;
                if (ct >= mt_total) ct = mt_total - 1;   // padding rows of the last group: outputs are never stored
                const f16 *src = a.wpk + ((size_t)ct * ns_total + (size_t)pass * nsp) * 512;
                f16 *dstm = s_w + (size_t)m * n16m * 8;
                for (int j = t; j < n16m; j += NT)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)j * 8),
                                                     (__attribute__((address_space(3))) void *)(dstm + (size_t)(j - lane) * 8), 16, 0, 0);
            }
        }
        // ---- stage this pass's channel slice of the input tile ------------------------------------------
        const int n_items = HT * WT * nc8_cur;
        // Batches of items per thread: all global loads of a batch are issued back to back (unconditionally, with
        // clamped coordinates: no branch between them), the transforms (BN / pool / up+add / BN backward) follow, so a
        // thread pays one memory latency per batch.  Deep layers have 10-20 items per thread: 1-2 batches.
        if constexpr (LM == LM_U8) {     // uint8 input with 5-8 channels (never the shipped configs): simple path
            for (int i = t; i < n_items; i += NT) {
                const int pix = i / nc8_cur, c8 = i - pix * nc8_cur;
                const int py = pix / WT, px = pix - py * WT;
                const int y = tc.ty0 + py - halo, x = tc.tx0 + px - halo;
                f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (y >= 0 && y < H && x >= 0 && x < W) v = load_chunk(a.x, tc.b, y, x, H, W, c8_lo + c8, s_aff);
                *reinterpret_cast<f16x8 *>(s_tile + (pix * ps + c8) * 16) = v;
            }
        } else {
            constexpr int BATCH = 4;
            // divisions by the small run-time divisors nc8_cur (1..IMK_PASS_CAP) and WT (16 / 18) as multiply-shift:
            // floor(n * ceil(2^16 / d) / 2^16) == n / d for n * d < 2^16 (here n < 1500, d <= 18)
            const int mg_c = (65536 + nc8_cur - 1) / nc8_cur, mg_w = (65536 + WT - 1) / WT;
            const int n_batches = (n_items + BATCH * NT - 1) / (BATCH * NT);   // uniform trip count (barrier inside)
            for (int bt = 0; bt < n_batches; ++bt) {
                const int i0 = t + bt * BATCH * NT;
                RawChunk<LM> r[BATCH];
                int dst[BATCH], c8s[BATCH];
                unsigned ok = 0;
#pragma unroll
                for (int u = 0; u < BATCH; ++u) {
                    const int i = i0 + u * NT;
                    const int ii = i < n_items ? i : t;          // idle slots repeat this thread's first item
                    const int pix = (ii * mg_c) >> 16;            // ii / nc8_cur, exact for these ranges (see mg_c)
                    const int c8 = ii - pix * nc8_cur;
                    const int py = (pix * mg_w) >> 16, px = pix - py * WT;   // pix / WT
                    dst[u] = i < n_items ? (pix * ps + c8) * 16 : -1;
                    c8s[u] = c8_lo + c8;
                    if (psrc_load<LM, 0>(src, py, px, c8s[u], W, r[u])) ok |= 1u << u;
                }
                if (aff_pending) { __syncthreads(); aff_pending = false; }   // affine table (uniform branch)
#pragma unroll
                for (int u = 0; u < BATCH; ++u) {
                    if (dst[u] >= 0) {
                        f16x8 v = raw_transform<LM>(r[u], s_aff, a.x.cs_in, c8s[u], a.x.cin, a.x.u8_div);
                        if (!(ok & (1u << u))) v = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                        *reinterpret_cast<f16x8 *>(s_tile + dst[u]) = v;
                    }
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the asynchronous weight copy has landed
        __syncthreads();
        if (pass == 0) IMK_STAMP(1);

        // ---- MFMA loop over (tap, chunk) of this pass ----------------------------------------------------------
        const int nq = T * nc8p;          // the packing pads every pass to nc8p chunks (zero weights beyond nc8_cur)
        int q = g;                        // this lane group's k-slot (tap, chunk) index at the next step to load
        const int mg_q = (65536 + nc8p - 1) / nc8p;    // q / nc8p as multiply-shift (q < 9 * 4 + 4, nc8p <= IMK_PASS_CAP)
        // Software-pipelined over the k-steps with two register sets of fragments: the LDS (or L2) reads of step s + 1
        // are issued before the MFMAs of step s, so their latency runs under the matrix pipe instead of in front of it
        // (one wave per SIMD and workgroup: nothing else would hide it).  The loop body has no conditional load or MFMA
        // (see load_frag), so the compiler's wait counts leave exactly the younger set in flight.
        auto k_loop = [&]() {
            int sn = 0;     // next k-step to load
            const f16 *wsrc[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                wsrc[m] = s_w + ((size_t)m * nsp * 64 + lane) * 8;
            }
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
            auto load_frag = [&](f16x8 (&af)[MT], f16x8 (&bf)[P]) {
                const int tap = (q * mg_q) >> 16, c8 = q - tap * nc8p;
                const int ty = ks3 ? (tap * 11) >> 5 : 0, tx = ks3 ? tap - 3 * ty : 0;   // tap / 3 for tap < 16
                const bool vq = (q < nq) && (c8 < nc8_cur);
                const int off = vq ? ((ty * WT + tx) * ps + c8) * 16 : 0;     // invalid k-slots: zero weights
                // steps past the end (the second half of the last pair when nsp is odd, and the look-ahead of the last
                // iteration) re-read the last step and get a zero weight fragment: no branch, so all MFMAs of the loop
                // sit in one basic block with one set of accumulators
                const unsigned live = sn < nsp ? 0xFFFFFFFFu : 0u;
                const int sc = min(sn, nsp - 1);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const u32x4 w = *reinterpret_cast<const u32x4 *>(wsrc[m] + (size_t)sc * 512) & live;
                    af[m] = __builtin_bit_cast(f16x8, w);
                }
#pragma unroll
                for (int p = 0; p < P; ++p) bf[p] = *reinterpret_cast<const f16x8 *>(s_tile + base[p] + off);
                ++sn;
                q += 4;
            };
            auto mma = [&](const f16x8 (&af)[MT], const f16x8 (&bf)[P]) {
#pragma unroll
                for (int p = 0; p < P; ++p)
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[m], bf[p], acc[m][p], 0, 0, 0);
            };
            f16x8 a0[MT], b0[P], a1[MT], b1[P];
            load_frag(a0, b0);
            for (int s = 0; s < nsp; s += 2) {
                load_frag(a1, b1);
                mma(a0, b0);
                load_frag(a0, b0);
                mma(a1, b1);
            }
        };
        k_loop();
        if (pass == 0) IMK_STAMP(2);
    }
    IMK_STAMP(3);

    // ---- epilogue ---------------------------------------------------------------------------------
    // A lane owns 4 channels of one pixel per (m, p): stored directly, that is 8 bytes every cs_out * 2 bytes -- a third of
    // every 64-byte line.  So the tile goes through LDS ([pixel][MT * 16 channels], the region the input tile occupied)
    // and leaves as 16-byte chunks, MT * 32 contiguous bytes per pixel.  The ReLU masks and BN outputs that the gradient
    // epilogues read are all requested up front (one latency, not one per (m, p)).
    const int x = tc.tx0 + n;
    const int my = H - 1 - tc.ty0, mx = W - 1 - tc.tx0;            // last row / column of the image, relative to the tile
    float s1[MT][4], s2[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[m][r] = s2[m][r] = 0.f;
    const bool dystat = !CHAIN && (a.epi != EP_RELU) && a.dystat_z && a.stats_partial;
    const bool want_stats = !CHAIN && (((a.epi == EP_RELU) && a.stats_partial) || dystat);
    // biases of this lane's channels: one batch of loads, in flight across the barrier below
    float bias[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = (ct0 + m) * 16 + 4 * g + r;
            bias[m][r] = (a.epi == EP_RELU && a.bias && co < a.cout) ? a.bias[co] : 0.f;
        }
    constexpr int OPITCH = MT * 16 + 8;                       // halfs per pixel row of the output tile in LDS
    f16 *s_out = reinterpret_cast<f16 *>(smem);
    __syncthreads();                                           // every wave is done reading the input tile
    // one straight-line copy of the (m, p) loops per epilogue kind (the kind is uniform: decided once, outside)
    auto write_tile = [&](auto EPI_T, auto STAT_T, const f32x4 (&ac)[MT][P], const float (&bs)[MT][4], int cs_o) {
        constexpr int EPI = decltype(EPI_T)::value;
        constexpr int STAT = decltype(STAT_T)::value;          // 0 none, 1 sum / sum of squares, 2 sum dy / sum dy*z
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int co0 = (ct0 + m) * 16 + 4 * g;
            // the ReLU mask / BN output this channel tile's gradient epilogue reads: the P loads go out together
            f16x4 mk[P], zz[P];
            if (EPI == EP_MASK || STAT == 2) {
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    // dead lanes (pixels outside the image, padding channels) read a clamped, valid address
                    const unsigned o = __umul24(__umul24(min(wave * P + p, my), W) + min(n, mx), (unsigned)cs_o * 2u) + (co0 < cs_o ? co0 * 2 : 0);
                    if (EPI == EP_MASK) mk[p] = *reinterpret_cast<const f16x4 *>(pix_base(a.mask, tc.b, H, W, tc.ty0, tc.tx0, (unsigned)cs_o * 2u) + o);
                    if (STAT == 2) zz[p] = *reinterpret_cast<const f16x4 *>(pix_base(a.dystat_z, tc.b, H, W, tc.ty0, tc.tx0, (unsigned)cs_o * 2u) + o);
                }
            }
#pragma unroll
            for (int p = 0; p < P; ++p) {
                f16x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (EPI == EP_RELU) v[r] = (f16)fmaxf(ac[m][p][r] + bs[m][r], 0.f);
                    else if (EPI == EP_MASK) v[r] = ((float)mk[p][r] > 0.f) ? (f16)ac[m][p][r] : (f16)0.f;
                    else v[r] = (f16)ac[m][p][r];
                }
                *reinterpret_cast<f16x4 *>(s_out + ((wave * P + p) * 16 + n) * OPITCH + m * 16 + 4 * g) = v;
                if (STAT) {
                    const int y = tc.ty0 + wave * P + p;
                    const float live = (co0 < cs_o && y < H && x < W) ? 1.f : 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float f = live * (float)v[r];
                        s1[m][r] += f;
                        s2[m][r] += f * (STAT == 2 ? (float)zz[p][r] : f);
                    }
                }
            }
        }
    };
    // copy-out: 16-byte chunks, (MT * 2) per pixel; channel tiles beyond cs_o are skipped
    auto copy_out = [&](f16 *out, int cs_o) {
        const int cpp = MT * 2;                                // chunks per pixel
        const int n_live = min(cpp, (cs_o - ct0 * 16) / 8);    // cs_o is a multiple of 8
        char *ob = const_cast<char *>(pix_base(out, tc.b, H, W, tc.ty0, tc.tx0, (unsigned)cs_o * 2u)) + ct0 * 32;
        for (int i = t; i < TH * 16 * cpp; i += NT) {
            const int pixl = i / cpp, ch = i - pixl * cpp;       // cpp = MT * 2: a power of two
            const int py = pixl >> 4, px = pixl & 15;
            if (ch < n_live && py <= my && px <= mx)
                *reinterpret_cast<f16x8 *>(ob + __umul24(__umul24(py, W) + px, (unsigned)cs_o * 2u) + ch * 16) =
                    *reinterpret_cast<const f16x8 *>(s_out + pixl * OPITCH + ch * 8);
        }
    };
    auto reduce_stats = [&](int cs_o) {
        __syncthreads();  // everyone is done reading the tile; reuse its LDS
        float *s_red = reinterpret_cast<float *>(smem);  // [NW waves][2][16*MT]
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
            if (co < cs_o) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) v += s_red[(w * 2 + which) * 16 * MT + c];
                a.stats_partial[(size_t)bx * 2 * cs_o + which * cs_o + co] = v;
            }
        }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    if constexpr (CHAIN) {
        // ---- chained 1x1 conv (Conv3x3+ReLU -> Conv1x1+ReLU of a block): this workgroup holds ALL channels of its tile, so
        // the second conv's pixel operand is the first's output tile where the epilogue put it -- each wave reads back the
        // 64 pixels it wrote itself (no barrier), 8 consecutive channels per lane and k-step, against the regular packed
        // weights of the 1x1 (one k-step per channel pass), staged next to the tile at kernel entry.
        write_tile(I0{}, I0{}, acc, bias, a.cs_out);
        const int nc8_2 = a.cs_out / 8, nc8p2 = imk_pass_chunks(nc8_2), ns2 = imk_cdiv_d(nc8_2, nc8p2);
        const int mt2 = (a.cout2 + 15) / 16;
        f32x4 acc2[MT][P];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int p = 0; p < P; ++p) acc2[m][p] = f32x4{0, 0, 0, 0};
        float bias2[MT][4];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = m * 16 + 4 * g + r;
                bias2[m][r] = co < a.cout2 ? a.bias2[co] : 0.f;
            }
        for (int s2i = 0; s2i < ns2; ++s2i) {
            const int c8 = s2i * nc8p2 + g;
            const int off = (g < nc8p2 && c8 < nc8_2) ? c8 * 8 : 0;     // invalid k-slots: zero weights
            f16x8 bf[P];
#pragma unroll
            for (int p = 0; p < P; ++p) bf[p] = *reinterpret_cast<const f16x8 *>(s_out + ((wave * P + p) * 16 + n) * OPITCH + off);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (m < mt2) {
                    const f16x8 af = *reinterpret_cast<const f16x8 *>(s_w2 + ((size_t)(m * ns2 + s2i) * 64 + lane) * 8);
#pragma unroll
                    for (int p = 0; p < P; ++p) acc2[m][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[p], acc2[m][p], 0, 0, 0);
                }
            }
        }
        if (a.out) {                       // training: the backward pass needs the intermediate
            __syncthreads();
            copy_out(a.out, a.cs_out);
            __syncthreads();               // (without the copy-out nobody reads another wave's rows: no barrier)
        }
        IMK_STAMP(4);
        if (a.stats_partial) write_tile(I0{}, I1{}, acc2, bias2, a.cs_out2); else write_tile(I0{}, I0{}, acc2, bias2, a.cs_out2);
        __syncthreads();
        copy_out(a.out2, a.cs_out2);
        IMK_STAMP(5);
        if (a.stats_partial) reduce_stats(a.cs_out2);
    } else {
        if (a.epi == EP_RELU) { if (want_stats) write_tile(I0{}, I1{}, acc, bias, a.cs_out); else write_tile(I0{}, I0{}, acc, bias, a.cs_out); }
        else if (a.epi == EP_MASK) { if (dystat) write_tile(I2{}, I2{}, acc, bias, a.cs_out); else write_tile(I2{}, I0{}, acc, bias, a.cs_out); }
        else { if (dystat) write_tile(I1{}, I2{}, acc, bias, a.cs_out); else write_tile(I1{}, I0{}, acc, bias, a.cs_out); }
        __syncthreads();
        IMK_STAMP(4);
        copy_out(a.out, a.cs_out);
        IMK_STAMP(5);
        if (want_stats) reduce_stats(a.cs_out);  // workgroup-uniform branch
    }
    IMK_STAMP_END(6);
}

template <int TH, int MT, int LM, bool CHAIN = false>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ImkConvArgs a, ImkConvGeom gm) {
    conv_mfma_body<TH, MT, LM, CHAIN>(a, gm, blockIdx.x, blockIdx.y);
}

// =====================================================================================================
// forward / dgrad, persistent + register-prefetch pipelined variant for the wide, shallow layers
// (<= 16 input channels, <= 16 output channels: levels 0/1 at alpha <= 1, where 80 % of the bytes are).
// Those layers are pure streaming (a 16x16x8 tile is 4 KB in, 4 KB out, 12 MFMAs), so what limits them
// is memory latency per workgroup, not bandwidth or math.  Differences to conv_mfma_kernel:
//   * grid = (resident workgroups), each walks tiles blockIdx.x, +gridDim.x, ... ;
//   * the global loads of tile i+1 are issued into registers BEFORE the MFMAs / stores of tile i, so every
//     resident workgroup always has a tile of loads in flight;
//   * the packed weights (<= 5 k-steps) and the per-lane LDS offsets are loaded once per workgroup.
// =====================================================================================================
// PAIR (layers with <= 8 output channels, i.e. level 0 at alpha = 0.5): a 16-row MFMA output would be half empty, so
// the two halves of the K dimension carry two different tile rows instead: k-slot groups g = 0,1 hold (tap, chunk)
// pairs of row A, g = 2,3 the same pairs of row B; weight rows 0-7 are non-zero only in the first half, rows 8-15 (the
// same 8 output channels again) only in the second.  D rows 0-7 are then row A's channels, rows 8-15 row B's: every
// lane owns 4 channels of one pixel and the epilogue runs with all 64 lanes instead of 32.
//
// Memory pipeline.  gfx950 returns vector-memory results in issue order and counts loads and stores in one counter
// (vmcnt), and the compiler can only wait for "all but the N youngest" when N is the same on every path.  So inside the
// loop every global access is unconditional and their number per iteration is a compile-time constant:
//   * out-of-image halo pixels and idle staging slots load a clamped in-image address and are zeroed at staging time;
//   * the prefetch of the last iteration re-reads the current tile;
//   * the epilogue kind (EPI, DYSTAT), the chain's intermediate store (CHAIN = 2: none) and full tiles (FULL: H and W
//     multiples of 16 and every lane owning real output channels, so no guard around any store) are template
//     parameters, and the epilogue's own loads (ReLU mask, BN output z) are issued BEFORE the prefetch.
// With that the waits become vmcnt(#prefetch loads) before the epilogue and vmcnt(#stores) before staging, i.e. the
// prefetch stays in flight across the MFMAs and the epilogue and the stores across the next staging.
// CHAIN: 0 none, 1 chained 1x1 with the intermediate stored (training), 2 chained, intermediate not stored (inference).
// WG (dgrad of a 1x1 conv behind a BatchNorm, ReLU mask = the conv's own input x; LM_BNBWD, EP_MASK, full tiles): the launch
// also accumulates the conv's weight / bias gradient.  x is staged in LDS next to the gradient tile (the epilogue then takes
// its mask from there instead of from global memory), and per tile every wave adds its 64 pixels to
// dW[ci][co] += x^T . dA with transposed LDS reads (ds_read_tr16_b64) -- 4 MFMAs and 8 LDS reads per wave and tile.
// WG = 2: the same for a 1x1 conv that READS a BatchNorm output (the U-Net's output layer; LM_RAW, EP_PLAIN, DYSTAT):
// x = fp16(z * sc + sh) with z the tensor the BN-gradient statistics need anyway -- staged raw (the epilogue's z comes from
// LDS, too), the affine applied to the transposed reads with the lane's channel constants.
// LDS of conv_pipe_kernel without the PRE stage's extra tile: tile, affine table, statistics, u8 rows, fused-wgrad x tile
// (+ slack: its transposed reads reach 16 bytes past a pixel); at least the fused 3x3 weight gradient's final reduction
constexpr size_t pipe_lds_base(int nc8, bool pair, int wg) {
    size_t b = (size_t)18 * 18 * imk_lds_pitch(nc8) * 16 + (8 * 16 + 4 * 2 * 16) * sizeof(float) + 1024 +
               ((wg && wg != 4) ? (size_t)(wg == 3 ? 18 * 18 : 256) * ((pair ? 1 : 2) | 1) * 16 + 64 : 0);
    if (wg == 3 && b < 4 * 5 * 256 * sizeof(float)) b = 4 * 5 * 256 * sizeof(float);
    return (b + 15) & ~(size_t)15;
}

// PRE (inference): the conv's input is itself a Conv1x1 + ReLU + BatchNorm of the tensor `a.x` describes (the input block in
// front of the first encoder conv, unet.py:4-9; the decoder's Conv1x1 on upsample + skip, unet.py:32-35) and is computed here,
// per tile incl. halo, on the matrix cores: `a.x` (uint8 image or up+add, PRE = its chunks per pixel) is staged into a second
// LDS tile, one MFMA per 32 (pair layout) / 16 pixels with the 1x1's REGULAR packed weights gives its channels, the lanes add
// bias, ReLU, round to fp16, apply the BatchNorm, zero what lies outside the image and write the 3x3's input tile.  The
// intermediate tensor is neither written nor read, and every value is bit-identical to the two-launch path (same MFMA, same
// operand layout per pixel).
// DYN: the workgroups take their tiles from per-group counters (a.sched; ImkWalk, dynamic form) -- launches without per-workgroup
// partial rows only (EP_RELU without statistics, no fused weight gradient: inference)
template <int LM, int NC8, int CHAIN, bool PAIR, int EPI, bool DYSTAT, bool FULL, int WG = 0, int PRE = 0, bool DYN = false>
// (inference launches of the 8-channel layers: four workgroups per CU, i.e. <= 128 registers -- the decoder's launch sat at 134 and ran
//  three waves per SIMD: 0.891 -> 0.873 ms per 256-image forward; the 16-channel and the training variants gain nothing from a cap)
__global__ __launch_bounds__(256, (DYN && NC8 == 1) ? IMK_INF_WAVES : 1) void conv_pipe_kernel(ImkConvArgs a, int tiles_x, int tiles_y, int n_tiles,
                                                        unsigned magic_tx, ImkWalk wk) {
    static_assert(!DYN || (EPI == EP_RELU && !DYSTAT && WG == 0), "dynamic walk: results must not depend on the tile -> workgroup map");
    static_assert(PRE == 0 || (WG == 0 && EPI == EP_RELU && !DYSTAT && (LM == LM_U8 || LM == LM_UPADD)), "pre-stage: inference forward only");
    static_assert(WG != 1 || (LM == LM_BNBWD && CHAIN == 0 && EPI == EP_MASK && !DYSTAT && FULL), "fused wgrad: 1x1 dgrad behind a BatchNorm");
    static_assert(WG != 2 || (LM == LM_RAW && CHAIN == 0 && EPI == EP_PLAIN && DYSTAT && FULL), "fused wgrad: 1x1 dgrad in front of a BatchNorm");
    static_assert(WG != 3 || (LM == LM_RAW && CHAIN == 0 && EPI == EP_PLAIN && DYSTAT && FULL), "fused wgrad: 3x3 dgrad in front of a BatchNorm");
    // WG = 4 (round 4): no weight gradient -- the 1x1 dgrad that produces the gradient of an upsample + add tensor also emits its 2x2
    // sums (= the gradient of the lower block's BatchNorm output) with their BatchNorm-backward statistics (ImkConvArgs::sum2_out)
    static_assert(WG != 4 || (LM == LM_BNBWD && CHAIN == 0 && EPI == EP_PLAIN && !DYSTAT && FULL && PRE == 0), "2x2-sum epilogue: 1x1 dgrad of a decoder's first conv");
    constexpr int P = PAIR ? 2 : 4;             // MFMA column blocks per wave: 4 tile rows, one or two per block
    constexpr int PS = imk_lds_pitch(NC8);      // pixel stride in 16-byte chunks (imk_stage.h)
    constexpr int NCI = PRE ? PRE : NC8;        // chunks per pixel of the STAGED tensor (PRE: the 1x1's input)
    constexpr int PSI = imk_lds_pitch(NCI);
    constexpr int MAX_ITEMS = (18 * 18 * NCI + 255) / 256;
    constexpr int MAX_NS = PAIR ? 3 * NC8 : (9 * NC8 + 3) / 4;       // (PAIR 1x1: (NC8 + 1) / 2 <= 3 NC8)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int ks3 = (a.ksize == 3);
    const int halo = ks3 ? 1 : 0;
    const int HT = 16 + 2 * halo, WT = TW + 2 * halo;
    const int n_items = HT * WT * NCI;
    // PAIR, 3x3: the block's two tile rows are vertically adjacent, so they share 2 of their 3 input rows -- the k-slots are the UNION of
    // their taps, 4 input rows x 3 columns = 12 (tap, chunk) slots per chunk: 3 NC8 k-steps instead of ceil(9 NC8 / 2) (NC8 = 1: 3, not 5)
    const bool shared_taps = PAIR && ks3;
    const int nq = (shared_taps ? 12 : (ks3 ? 9 : 1)) * NC8, ns = shared_taps ? 3 * NC8 : (PAIR ? (nq + 1) / 2 : (nq + 3) / 4);
    uint8_t *s_tile = smem;
    uint8_t *s_t0 = smem + pipe_lds_base(NC8, PAIR, WG);                          // PRE: the staged input of the 1x1, [pixel][PSI]
    float *s_pre = reinterpret_cast<float *>(s_t0 + 18 * 18 * PSI * 16);          // PRE: [bias | scale | shift][16]
    uint8_t *s_stage = PRE ? s_t0 : s_tile;
    float *s_aff = reinterpret_cast<float *>(smem + 18 * 18 * PS * 16);
    float *s_red = s_aff + 8 * 16;              // [4 waves][2][16] (the affine table has up to 7 rows of 16: LM_STEM)
    const int t = threadIdx.x;
    IMK_STAMP_BEGIN(conv, 40000 + LM * 1000 + WG * 100 + CHAIN * 10 + (DYSTAT ? 1 : 0));
    IMK_WGSTAMP_BEGIN(conv, 40000 + LM * 1000 + WG * 100 + CHAIN * 10 + (DYSTAT ? 1 : 0));
    const int lane = t & 63, wave = t >> 6, n = lane & 15, g = lane >> 4;
    const int set = PAIR ? (g >> 1) : 0;        // PAIR: which of the block's two tile rows this lane feeds and owns
    const int H = a.H, W = a.W;
    constexpr int cs_in = NCI * 8;              // == a.x.cs_in (checked at launch): the affine table's rows sit at compile-time offsets
    const int per_img = tiles_x * tiles_y;
    auto tile_row = [&](int p) { return wave * 4 + (PAIR ? 2 * p + set : p); };
    constexpr unsigned CSB = NCI * 16;          // bytes per pixel of the (fp16) input tensor: cs_in = NCI * 8

    // packed weights and per-lane LDS offsets of every k-step: once per workgroup
    f16x8 af[MAX_NS];
    int off[MAX_NS];
    // k-slots beyond the real taps carry zero WEIGHTS; their pixel operand may be any finite tile data, so
    // those lanes simply read offset 0 (no per-read select).
#pragma unroll
    for (int s = 0; s < MAX_NS; ++s) {
        const int q = (PAIR && !shared_taps) ? 2 * s + (g & 1) : 4 * s + g;
        const bool vq = (s < ns) && (q < nq);
        const int tap = q / NC8, c8 = q - tap * NC8;          // shared taps: tap = (input row 0..3 of the pair's window) * 3 + column
        const int ty = ks3 ? tap / 3 : 0, tx = ks3 ? tap - 3 * (tap / 3) : 0;
        off[s] = vq ? ((ty * WT + tx) * PS + c8) * 16 : 0;
        af[s] = (s < ns) ? *reinterpret_cast<const f16x8 *>(a.wpk + ((size_t)s * 64 + lane) * 8) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    int base[P];
#pragma unroll
    for (int p = 0; p < P; ++p) base[p] = ((shared_taps ? wave * 4 + 2 * p : tile_row(p)) * WT + n) * PS * 16;   // shared taps: from the pair's upper row
    // this thread's staging items (constant over tiles): LDS offset and position inside the halo tile.  Idle slots
    // (beyond the tile's item count) repeat slot 0's pixel, so that their (unconditional) load is a cache hit.
    int it_lds[MAX_ITEMS], it_py[MAX_ITEMS], it_px[MAX_ITEMS], it_c8[MAX_ITEMS];
#pragma unroll
    for (int k = 0; k < MAX_ITEMS; ++k) {
        const int i = t + 256 * k;
        const bool live = i < n_items;
        const int pix = (live ? i : t % n_items) / NCI;
        it_c8[k] = (live ? i : t % n_items) - pix * NCI;
        it_py[k] = pix / WT;
        it_px[k] = pix - it_py[k] * WT;
        it_lds[k] = live ? (pix * PSI + it_c8[k]) * 16 : -1;
    }
    float bias[4] = {0, 0, 0, 0}, bias2[4] = {0, 0, 0, 0};
    const int co0 = PAIR ? 4 * (g & 1) : 4 * g;
    if (EPI == EP_RELU && a.bias)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[r] = (co0 + r < a.cout) ? a.bias[co0 + r] : 0.f;
    f16x8 af2 = {0, 0, 0, 0, 0, 0, 0, 0};
    if (CHAIN) {
        af2 = *reinterpret_cast<const f16x8 *>(a.wpk2 + (size_t)lane * 8);
#pragma unroll
        for (int r = 0; r < 4; ++r) bias2[r] = (co0 + r < a.cout2) ? a.bias2[co0 + r] : 0.f;
    }
    const bool want_stats = DYSTAT || WG == 4 || ((EPI == EP_RELU) && a.stats_partial);
    // PRE: the 1x1's weight fragment (one k-step: its input has at most 16 channels), its constants in LDS, and the pixels
    // this lane's accumulator rows belong to in each of the wave's NJ MFMAs (pair layout: 2 x 16 pixels per MFMA)
    constexpr int NJ = PAIR ? 3 : 6;            // 4 waves x NJ x (32 | 16) = 384 >= 18 x 18 pixels
    f16x8 af_pre = {0, 0, 0, 0, 0, 0, 0, 0};
    int pre_pix[PRE ? NJ : 1], pre_yx[PRE ? NJ : 1];
    const int pre_c8 = PAIR ? (g & 1) : g;      // the chunk of the staged pixel this lane group feeds (zero weights beyond NCI)
    const int pre_cb = PAIR ? 4 * (g & 1) : 4 * g;   // this lane's 4 output channels
    if constexpr (PRE) {
        af_pre = *reinterpret_cast<const f16x8 *>(a.pre_wpk + (size_t)lane * 8);
        if (t < 16) {
            s_pre[t] = t < a.pre_cout ? a.pre_bias[t] : 0.f;
            s_pre[16 + t] = t < NC8 * 8 ? a.pre_sc[t] : 0.f;
            s_pre[32 + t] = t < NC8 * 8 ? a.pre_sh[t] : 0.f;
        }
        const int n_pix = HT * WT;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int grp = wave * NJ + j;
            const int px = min(PAIR ? (grp * 2 + (g >> 1)) * 16 + n : grp * 16 + n, n_pix - 1);   // past the end: the last pixel again
            const int py = px / WT;
            pre_pix[j] = px;
            pre_yx[j] = (py << 8) | (px - py * WT);
        }
    }
    // FULL also promises that every lane owns real channels (PAIR, or 16-channel outputs): no guard around any store
    const bool lane_out = FULL || co0 < (CHAIN ? a.cs_out2 : a.cs_out);   // this lane's 4 channels exist in the output tensor
    const bool lane_mid = FULL || co0 < a.cs_out;                          // ... in the chain's intermediate

    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};   // BN statistics, accumulated over all tiles of this workgroup
    RawChunk<LM> raw[MAX_ITEMS];
    unsigned valid = 0;
    // uint8 image, full 16 x 16 tiles of a 1x1 conv (the stem): a tile row is 16 * cin contiguous bytes = cin aligned
    // 16-byte segments, so 16 * cin threads fetch the whole tile with one wide load each (instead of cin byte loads
    // per pixel and thread); the bytes go through LDS to the pixel that owns them.
    constexpr bool U8ROWS = (LM == LM_U8) && FULL && !PRE;
    uint8_t *s_u8 = reinterpret_cast<uint8_t *>(s_red + 4 * 2 * 16);   // [16 rows][cin * 16 bytes], cin <= 4
    // WG: x tile [256 px][XS chunks] behind it, and the two persistent accumulators of the weight / bias gradient
    constexpr int NCX = PAIR ? 1 : 2, XS = NCX | 1;     // x = the dgrad's OUTPUT channels: 8 (pair layout) or 16
    uint8_t *s_x = s_u8 + 1024;
    // WG = 3 (3x3 conv on a BatchNorm output): x = fp16(z * sc + sh) WITH its halo, 18 x 18 pixels, zero outside the image
    constexpr int MAX_XI = WG == 3 ? (18 * 18 * NCX + 255) / 256 : NCX;
    f16x8 xr[MAX_XI];
    unsigned xvalid = 0;
    int xi_lds[MAX_XI], xi_py[MAX_XI], xi_px[MAX_XI], xi_c8[MAX_XI];
    if constexpr (WG == 3) {
#pragma unroll
        for (int k = 0; k < MAX_XI; ++k) {
            const int i = t + 256 * k;
            const bool live = i < 18 * 18 * NCX;
            const int ii = live ? i : t;                 // idle slots repeat slot 0 (t < 324 * NCX always)
            const int pix = ii / NCX;
            xi_c8[k] = ii - pix * NCX;
            xi_py[k] = pix / 18;
            xi_px[k] = pix - xi_py[k] * 18;
            xi_lds[k] = live ? (pix * XS + xi_c8[k]) * 16 : -1;
        }
    }
    f32x4 wacc = {0, 0, 0, 0}, bacc = {0, 0, 0, 0};
    f32x4 wtap[WG == 3 ? 9 : 1];
#pragma unroll
    for (int i = 0; i < (WG == 3 ? 9 : 1); ++i) wtap[i] = f32x4{0, 0, 0, 0};
    float x_sc = 0.f, x_sh = 0.f;                       // WG = 2: BatchNorm of this lane's input channel (row n of the A operand)
    if constexpr (WG == 2) { if (n < a.cs_out) { x_sc = a.wg_sc[n]; x_sh = a.wg_sh[n]; } }
    uint4 rowseg = {0, 0, 0, 0};
    const int u8_cin = a.x.cin, u8_nseg = 16 * u8_cin;
    const int u8_t = t < u8_nseg ? t : 0;                 // idle threads repeat segment 0 (loads are unconditional)
    const int u8_row = u8_t / u8_cin, u8_seg = u8_t - u8_row * u8_cin;
    const unsigned u8_off = (unsigned)(u8_row * W * u8_cin + u8_seg * 16);
    // thread t <-> pixel t of a full tile (WG = 1, 2), in bytes of a tensor with the output's channel stride
    const unsigned cso_b = (unsigned)a.cs_out * 2u;
    const unsigned xw_off = (unsigned)((t >> 4) * W + (t & 15)) * cso_b;
    // (Round 5, measured and removed: an interior-tile path -- 3 of 4 tiles have their whole halo window inside the image and need
    //  neither the clamps of psrc_load nor the inside / outside select of the staging -- chosen per tile by a scalar branch, same loads,
    //  bit-identical outputs: the 256-image ISIC forward went 0.843-0.846 -> 0.867-0.872 ms, SUIM 1.052-1.057 -> 1.059-1.067, with or
    //  without a second copy of the first-stage loop.  The clamps were never on the critical path; the extra branches and the
    //  rescheduled LDS waits were.  profiles/r05_notes.md.)
    auto issue = [&](const PTile &tc) {
        if constexpr (U8ROWS) {
            rowseg = *reinterpret_cast<const uint4 *>(pix_base(a.x.in, tc.b, H, W, tc.ty0, tc.tx0, (unsigned)u8_cin) + u8_off);
            return;
        }
        const PSrc<LM> src = psrc_of<LM, CSB>(a.x, tc, halo, HT, WT, H, W);
        valid = 0;
#pragma unroll
        for (int k = 0; k < MAX_ITEMS; ++k) {
#if IMK_ABL & 1
            const bool in_img = psrc_load<LM, CSB>(src, 1, 1, 0, W, raw[k]) || true;      // every lane the same (cached) address
#else
            const bool in_img = psrc_load<LM, CSB>(src, it_py[k], it_px[k], it_c8[k], W, raw[k]);
#endif
            valid |= ((it_lds[k] >= 0 && in_img) ? 1u : 0u) << k;
        }
        if constexpr (WG == 1 || WG == 2) {      // thread t <-> pixel t of the (full) tile
            const char *px = pix_base(WG == 1 ? a.mask : a.dystat_z, tc.b, H, W, tc.ty0, tc.tx0, cso_b) + xw_off;
#pragma unroll
            for (int q = 0; q < NCX; ++q) xr[q] = *reinterpret_cast<const f16x8 *>(px + q * 16);
        }
        if constexpr (WG == 3) {
            const char *bz = pix_base(a.dystat_z, tc.b, H, W, src.oy, src.ox, cso_b);      // 3x3: the same halo window as the gradient tile
            xvalid = 0;
#pragma unroll
            for (int k = 0; k < MAX_XI; ++k) {
                const int ry = min(max(xi_py[k], src.lo_y), src.hi_y), rx = min(max(xi_px[k], src.lo_x), src.hi_x);
                const bool ok = xi_lds[k] >= 0 && ry == xi_py[k] && rx == xi_px[k];
                xr[k] = *reinterpret_cast<const f16x8 *>(bz + __umul24(__umul24(ry, W) + rx, cso_b) + xi_c8[k] * 16);
                xvalid |= (ok ? 1u : 0u) << k;
            }
        }
    };
    // this lane's output pixels (column block p: tile row tile_row(p), column n) as byte offsets from the tile origin; constant
    // over full tiles.  o1: tensors with the output's channel stride (out, ReLU mask, BatchNorm output), o2: the chain's output.
    const unsigned cso2_b = CHAIN ? (unsigned)a.cs_out2 * 2u : 0u;
    unsigned o1c[P], o2c[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const unsigned lp = (unsigned)(tile_row(p) * W + n);
        o1c[p] = lp * cso_b + co0 * 2;
        o2c[p] = lp * cso2_b + co0 * 2;
    }

    // this workgroup's tiles: tile, tile + wk.step, ... inside its group's range (ImkWalk, imk_stage.h); a workgroup without a
    // tile (possible when the ranges are uneven) prefetches the launch's last tile and writes zero statistics rows
    const int wgrp = imk_walk_range(wk, blockIdx.x);
    int tile = wgrp * wk.chunk + (int)(blockIdx.x >> wk.shift);
    const int tile_end = min(n_tiles, (wgrp + 1) * wk.chunk);
    const int tile0 = min(tile, n_tiles - 1);
    PTile tc = ptile_at(tile0 / per_img, tile0 % per_img, tiles_x, magic_tx);
    issue(tc);
    // DYN: tickets of this group's counter number the tiles behind the statically assigned first ones; s_tk[2] hands the ticket
    // wave 0 received to the whole workgroup (written before a barrier, read after it; two slots: one per tile parity)
    __shared__ int s_tk[2];
    const bool tk_lane = DYN && t == 0;
    unsigned *const tk_head = DYN ? a.sched + wgrp * (IMK_SCHED_STRIDE / 4) : nullptr;
    const int tk_base = wgrp * wk.chunk + wk.step;
    int it_par = 0, tk_pending = 0;       // tk_pending: the ticket requested one tile ago, handed over one tile later -- a returning
    if constexpr (DYN) {                  // atomic takes 1-3 us under load, about a tile's time: two tiles of look-ahead hide it
        const int tk = imk_take_ticket(tk_head, tk_lane);
        tk_pending = imk_take_ticket(tk_head, tk_lane);
        if (tk_lane) s_tk[0] = tk;
    }
    stage_affine_table(a.x, s_aff);       // behind the first tile's loads: one exposed memory latency for both, not two
    if constexpr (WG == 3) {              // BatchNorm of the conv's input (LM_RAW leaves the table free)
        if (t < a.cs_out) { s_aff[t] = a.wg_sc[t]; s_aff[16 + t] = a.wg_sh[t]; }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): weights, biases and the first tile are in; nothing older is pending in the loop
    __syncthreads();                      // affine table visible
    while (tile < tile_end) {
        // registers -> LDS (BN / pool / up+add / u8 conversion applied here)
        if constexpr (U8ROWS) {
            if (t < u8_nseg) *reinterpret_cast<uint4 *>(s_u8 + t * 16) = rowseg;
            __syncthreads();
            const uint8_t *pb = s_u8 + (t >> 4) * u8_nseg + (t & 15) * u8_cin;   // pixel t of the tile
            f16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (f16)((j < 4 && j < u8_cin) ? (float)pb[j] / a.x.u8_div : 0.0f);
            *reinterpret_cast<f16x8 *>(s_tile + it_lds[0]) = v;
        } else {
#pragma unroll
            for (int k = 0; k < MAX_ITEMS; ++k) {
                if (it_lds[k] >= 0) {
#if IMK_ABL & 4
                    f16x8 v;
                    if constexpr (LM == LM_U8 || LM == LM_STEM) v = raw_transform<LM>(raw[k], s_aff, cs_in, it_c8[k], a.x.cin, a.x.u8_div, a.x.u8_c);
                    else v = raw[k].v[0];
#else
                    f16x8 v = raw_transform<LM>(raw[k], s_aff, cs_in, it_c8[k], a.x.cin, a.x.u8_div, a.x.u8_c);
#endif
                    if (!(valid & (1u << k))) v = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    *reinterpret_cast<f16x8 *>(s_stage + it_lds[k]) = v;
                }
            }
            if constexpr (WG == 1 || WG == 2) {
#pragma unroll
                for (int q = 0; q < NCX; ++q) *reinterpret_cast<f16x8 *>(s_x + (t * XS + q) * 16) = xr[q];
            }
            if constexpr (WG == 3) {      // the conv's input: BatchNorm applied, zero padding outside the image
#pragma unroll
                for (int k = 0; k < MAX_XI; ++k) {
                    if (xi_lds[k] >= 0) {
                        f16x8 v = affine8(xr[k], s_aff + xi_c8[k] * 8, s_aff + 16 + xi_c8[k] * 8);
                        if (!(xvalid & (1u << k))) v = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                        *reinterpret_cast<f16x8 *>(s_x + xi_lds[k]) = v;
                    }
                }
            }
        }
        __syncthreads();
        // output pixel of each column block; partial tiles clamp the coordinates used for LOADS (stores are guarded)
        const char *b_o1 = pix_base(a.out, tc.b, H, W, tc.ty0, tc.tx0, cso_b);
        const char *b_o2 = CHAIN ? pix_base(a.out2, tc.b, H, W, tc.ty0, tc.tx0, cso2_b) : nullptr;
        const char *b_mk = (EPI == EP_MASK && WG != 1) ? pix_base(a.mask, tc.b, H, W, tc.ty0, tc.tx0, cso_b) : nullptr;
        const char *b_zq = (DYSTAT && WG != 2) ? pix_base(a.dystat_z, tc.b, H, W, tc.ty0, tc.tx0, cso_b) : nullptr;
        unsigned o1[P], o1l[P], o2[P];     // o1l: for loads, lanes without real channels read channel 0
        bool inb[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if constexpr (FULL) {
                o1[p] = o1l[p] = o1c[p]; o2[p] = o2c[p]; inb[p] = true;
            } else {
                const int my = H - 1 - tc.ty0, mx = W - 1 - tc.tx0, r = tile_row(p);
                inb[p] = r <= my && n <= mx;
                const unsigned lp = __umul24(min(r, my), W) + min(n, mx);
                const unsigned ob = __umul24(lp, cso_b);
                o1[p] = ob + co0 * 2;
                o1l[p] = ob + (lane_out ? co0 * 2 : 0);
                o2[p] = __umul24(lp, cso2_b) + co0 * 2;
            }
        }
        f16x4 mk[P], zq[P];
        if (EPI == EP_MASK) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if constexpr (WG == 1) mk[p] = *reinterpret_cast<const f16x4 *>(s_x + ((tile_row(p) * 16 + n) * XS) * 16 + co0 * 2);
                else mk[p] = *reinterpret_cast<const f16x4 *>(b_mk + o1l[p]);
            }
        }
        if (DYSTAT) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if constexpr (WG == 2) zq[p] = *reinterpret_cast<const f16x4 *>(s_x + ((tile_row(p) * 16 + n) * XS) * 16 + co0 * 2);
                else zq[p] = *reinterpret_cast<const f16x4 *>(b_zq + o1l[p]);
            }
        }
        // WG = 4: the lower block's BatchNorm input at this lane's 2x2 window (tile rows 2 q, 2 q + 1 -> low-res row q of the tile)
        constexpr int NQ = PAIR ? P : P / 2;          // windows (in y) per lane
        f16x4 zlo[WG == 4 ? NQ : 1];
        unsigned olo[WG == 4 ? NQ : 1];
        const char *b_lo = nullptr;
        if constexpr (WG == 4) {
            const char *bz = pix_base(a.sum2_z, tc.b, H >> 1, W >> 1, tc.ty0 >> 1, tc.tx0 >> 1, cso_b);
            b_lo = pix_base(a.sum2_out, tc.b, H >> 1, W >> 1, tc.ty0 >> 1, tc.tx0 >> 1, cso_b);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                olo[q] = (unsigned)((wave * 2 + q) * (W >> 1) + (n >> 1)) * cso_b + co0 * 2;
                zlo[q] = *reinterpret_cast<const f16x4 *>(bz + olo[q]);
            }
        }
        int next;
        PTile tn;
        int tk_new = 0;
        if constexpr (DYN) {
            next = tk_base + __builtin_amdgcn_readfirstlane(s_tk[it_par]);
            tn = next < tile_end ? ptile_of_index(next, per_img, tiles_x, magic_tx, wk.magic_pi) : tc;
        } else {
            next = tile + wk.step;
            tn = next < tile_end ? ptile_next(tc, wk.q, wk.r, per_img, tiles_x, magic_tx) : tc;
        }
        issue(tn);                                // in flight during the MFMAs, the epilogue and its stores
        if constexpr (DYN) tk_new = imk_take_ticket(tk_head, tk_lane);     // the tile after next: its latency runs under this tile
        if constexpr (PRE && !(IMK_ABL & 8)) {
            // rows / columns of the halo tile that lie inside the image (the 3x3 pads its INPUT with zeros, not the 1x1's)
            const int oy = tc.ty0 - halo, ox = tc.tx0 - halo;
            const int lo_y = oy < 0 ? -oy : 0, hi_y = min(HT - 1, H - 1 - oy), lo_x = ox < 0 ? -ox : 0, hi_x = min(WT - 1, W - 1 - ox);
            const f32x4 pb = *reinterpret_cast<const f32x4 *>(s_pre + pre_cb), psc = *reinterpret_cast<const f32x4 *>(s_pre + 16 + pre_cb),
                        psh = *reinterpret_cast<const f32x4 *>(s_pre + 32 + pre_cb);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const f16x8 bfp = *reinterpret_cast<const f16x8 *>(s_t0 + (pre_pix[j] * PSI + (pre_c8 < NCI ? pre_c8 : 0)) * 16);
                const f32x4 pa = __builtin_amdgcn_mfma_f32_16x16x32_f16(af_pre, bfp, f32x4{0, 0, 0, 0}, 0, 0, 0);
                const int py = pre_yx[j] >> 8, px = pre_yx[j] & 255;
                const bool inside = py >= lo_y && py <= hi_y && px >= lo_x && px <= hi_x;
                const float pbv[4] = {pb[0], pb[1], pb[2], pb[3]};
                const f16x4 z4 = imk_bias_relu4(pa, pbv);                                                          // the 1x1's stored output
                unsigned wd[2];
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const f16x2 z2 = {z4[r], z4[r + 1]};
                    const f16x2 w2 = imk_affine2(z2, f32x2{psc[r], psc[r + 1]}, f32x2{psh[r], psh[r + 1]});      // as the staged form (imk_common.h)
                    wd[r >> 1] = inside ? __builtin_bit_cast(unsigned, w2) : 0u;                                   // (one select per pair)
                }
                const f16x4 w = __builtin_bit_cast(f16x4, uint2{wd[0], wd[1]});
                if (pre_cb < NC8 * 8)
                    *reinterpret_cast<f16x4 *>(s_tile + (pre_pix[j] * PS + (pre_cb >> 3)) * 16 + (pre_cb & 7) * 2) = w;
#ifdef IMK_PRE_DEBUG      // probe builds (tests/gpu_probe/pre_dump.py): the first stage's values of the tile's own pixels, for a per-pixel diff
                if (a.out && pre_cb < NC8 * 8 && py >= 1 && py <= 16 && px >= 1 && px <= 16 && inside) {
                    const size_t gp = ((size_t)(tc.b * H + tc.ty0 + py - 1) * W + tc.tx0 + px - 1) * (NC8 * 8) + pre_cb;
                    f16x4 zv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) zv[r] = (f16)fmaxf(pa[r] + pb[r], 0.f);
                    *reinterpret_cast<f16x4 *>(const_cast<f16 *>(a.mask) + gp) = zv;      // pre-BatchNorm (what the 1x1's own launch stores)
                    *reinterpret_cast<f16x4 *>(a.out + gp) = w;                            // after the BatchNorm (the 3x3's input)
                }
#endif
            }
            __syncthreads();      // the 3x3's input tile is complete
        }
        if constexpr (WG == 3) {
            // as below, with the 9 taps: dW[tap][ci][co] += x[pixel + tap][ci] * dA[pixel][co]; both tiles carry a halo here
            const int qq = n >> 2, pp = n & 3;
            f16x8 ones;
#pragma unroll
            for (int j = 0; j < 8; ++j) ones[j] = (f16)(n == 0 ? 1.0f : 0.0f);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int row = wave * 4 + 2 * kk + (g >> 1), xx = 4 * (g & 1) + qq;
                const uint8_t *pb = s_tile + (size_t)((row + 1) * 18 + xx + 1) * PS * 16 + pp * 8;     // dA, interior pixel
                const h4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(LDS_PTR(h4, pb));
                const h4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(LDS_PTR(h4, pb + 8 * PS * 16));
                f16x8 bfw;
#pragma unroll
                for (int e = 0; e < 4; ++e) { bfw[e] = (f16)b0[e]; bfw[4 + e] = (f16)b1[e]; }
                const uint8_t *pa = s_x + (size_t)(row * 18 + xx) * XS * 16 + pp * 8;                    // x at tap (0, 0)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const uint8_t *p = pa + (size_t)((tap / 3) * 18 + tap % 3) * XS * 16;
                    const h4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(LDS_PTR(h4, p));
                    const h4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(LDS_PTR(h4, p + 8 * XS * 16));
                    f16x8 afw;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { afw[e] = (f16)a0[e]; afw[4 + e] = (f16)a1[e]; }
                    wtap[tap] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afw, bfw, wtap[tap], 0, 0, 0);
                }
                bacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, bfw, bacc, 0, 0, 0);
            }
        }
        if constexpr (WG == 1 || WG == 2) {
            // this wave's 4 tile rows = 2 k-steps of 32 pixels; k-slot <-> pixel map as in wgrad_mfma_body (lane group g:
            // elements 0-3 = pixels x = 4(g&1) + 0..3 of row r0 + (g >> 1), elements 4-7 the pixels 8 further right)
            const int qq = n >> 2, pp = n & 3;
            f16x8 ones;
#pragma unroll
            for (int j = 0; j < 8; ++j) ones[j] = (f16)(n == 0 ? 1.0f : 0.0f);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int pi = (wave * 4 + 2 * kk + (g >> 1)) * 16 + 4 * (g & 1) + qq;
                const uint8_t *pb = s_tile + (size_t)pi * PS * 16 + pp * 8;      // dA[pixel][co]
                const uint8_t *pa = s_x + (size_t)pi * XS * 16 + pp * 8;         // x[pixel][ci]
                const h4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(LDS_PTR(h4, pb));
                const h4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(LDS_PTR(h4, pb + 8 * PS * 16));
                const h4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(LDS_PTR(h4, pa));
                const h4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(LDS_PTR(h4, pa + 8 * XS * 16));
                f16x8 bfw, afw;
#pragma unroll
                for (int e = 0; e < 4; ++e) { bfw[e] = (f16)b0[e]; bfw[4 + e] = (f16)b1[e]; afw[e] = (f16)a0[e]; afw[4 + e] = (f16)a1[e]; }
                if constexpr (WG == 2) {      // x = the BatchNorm output: fp16(z * sc + sh), what LM_AFFINE staging computes
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const f16x2 r2 = imk_affine2(f16x2{afw[e], afw[e + 1]}, f32x2{x_sc, x_sc}, f32x2{x_sh, x_sh});
                        afw[e] = r2[0]; afw[e + 1] = r2[1];
                    }
                }
                wacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(afw, bfw, wacc, 0, 0, 0);
                bacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, bfw, bacc, 0, 0, 0);   // column sums -> bias gradient
            }
        }
        f32x4 acc[P];
#pragma unroll
        for (int p = 0; p < P; ++p) acc[p] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < ((IMK_ABL & 16) ? 1 : MAX_NS); ++s) {
            if (s < ns) {
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const f16x8 bf = *reinterpret_cast<const f16x8 *>(s_tile + base[p] + off[s]);
                    acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[s], bf, acc[p], 0, 0, 0);
                }
            }
        }
        if (CHAIN) {
            // stage 1: h = relu(acc + b) in fp16 (stored only in training); stage 2 on the register tile:
            // the lane's 4 channels of pixel n are exactly k-slots (g, 0..3) of the next MFMA's B operand.
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const f16x4 hv = imk_bias_relu4(acc[p], bias);
                if (CHAIN == 1 && (FULL || inb[p]) && lane_mid) *reinterpret_cast<f16x4 *>(const_cast<char *>(b_o1) + o1[p]) = hv;
                const f16x8 bf2 = {hv[0], hv[1], hv[2], hv[3], 0, 0, 0, 0};
                const f32x4 a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(af2, bf2, f32x4{0, 0, 0, 0}, 0, 0, 0);
                const f16x4 v = imk_bias_relu4(a2, bias2);
                if ((FULL || inb[p]) && lane_out && (!(IMK_ABL & 2) || a.H < 0)) {
                    *reinterpret_cast<f16x4 *>(const_cast<char *>(b_o2) + o2[p]) = v;
                    if (want_stats)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float f = (float)v[r]; s1[r] += f; s2[r] += f * f; }
                }
            }
        } else {
            float hsum[WG == 4 ? P : 1][4];
#pragma unroll
            for (int p = 0; p < P; ++p) {
                f16x4 v;
                if (EPI == EP_RELU) {
                    v = imk_bias_relu4(acc[p], bias);
                } else if (EPI == EP_MASK) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = ((float)mk[p][r] > 0.f) ? (f16)acc[p][r] : (f16)0.f;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (f16)acc[p][r];
                }
                if ((FULL || inb[p]) && lane_out) {
                    *reinterpret_cast<f16x4 *>(const_cast<char *>(b_o1) + o1[p]) = v;
                    if (DYSTAT) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float f = (float)v[r]; s1[r] += f; s2[r] += f * (float)zq[p][r]; }
                    } else if (WG != 4 && want_stats) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float f = (float)v[r]; s1[r] += f; s2[r] += f * f; }
                    }
                }
                if constexpr (WG == 4) {
                    // 2x2 sums of the rounded outputs, (v00 + v01) + (v10 + v11) in fp32 like bn_bwd_prep_kernel<2>: the left / right
                    // neighbour is lane ^ 1; the row below is the other tile row of the pair (lane ^ 32) or the next accumulator block
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float f = (float)v[r]; hsum[p][r] = f + __shfl_xor(f, 1, 64); }
                }
            }
            if constexpr (WG == 4) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    f16x4 d;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float tot = PAIR ? hsum[q][r] + __shfl_xor(hsum[q][r], 32, 64) : hsum[2 * q][r] + hsum[2 * q + 1][r];
                        d[r] = (f16)tot;
                    }
                    if ((n & 1) == 0 && (!PAIR || set == 0)) {      // one lane per window stores and counts
                        *reinterpret_cast<f16x4 *>(const_cast<char *>(b_lo) + olo[q]) = d;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float f = (float)d[r]; s1[r] += f; s2[r] += f * (float)zlo[q][r]; }
                    }
                }
            }
        }
        if constexpr (DYN) { it_par ^= 1; if (tk_lane) s_tk[it_par] = tk_pending; tk_pending = tk_new; }
        __syncthreads();   // tile reads done: the LDS tile may be overwritten
        tile = next;
        tc = tn;
    }
    if constexpr (WG == 3) {    // [10][256]: 9 taps + bias, reduced over the 4 waves in two rounds of 5 (20 KB of LDS)
        float *s_acc = reinterpret_cast<float *>(smem);
        float *dst = a.wg_partial + (size_t)blockIdx.x * 10 * 256;
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const f32x4 v = (rd * 5 + i < 9) ? wtap[(rd * 5 + i) % 9] : bacc;
#pragma unroll
                for (int r = 0; r < 4; ++r) s_acc[(wave * 5 + i) * 256 + r * 64 + lane] = v[r];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 5; ++i)
                dst[(rd * 5 + i) * 256 + t] = s_acc[(0 * 5 + i) * 256 + t] + s_acc[(1 * 5 + i) * 256 + t] + s_acc[(2 * 5 + i) * 256 + t] +
                                              s_acc[(3 * 5 + i) * 256 + t];
        }
        __syncthreads();
    }
    if constexpr (WG == 1 || WG == 2) {    // the 4 waves' weight / bias gradient accumulators -> this workgroup's partial row [2][256]
        float *s_acc = reinterpret_cast<float *>(smem);          // [4][2][256]: the tile region is free now
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s_acc[(wave * 2 + 0) * 256 + r * 64 + lane] = wacc[r];
            s_acc[(wave * 2 + 1) * 256 + r * 64 + lane] = bacc[r];
        }
        __syncthreads();
        float *dst = a.wg_partial + (size_t)blockIdx.x * 2 * 256;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            dst[i * 256 + t] = s_acc[(0 * 2 + i) * 256 + t] + s_acc[(1 * 2 + i) * 256 + t] + s_acc[(2 * 2 + i) * 256 + t] +
                               s_acc[(3 * 2 + i) * 256 + t];
        __syncthreads();   // s_acc overlaps the statistics scratch (s_red) that the block below writes
    }
    if (want_stats) {      // one partial row per workgroup
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v1 = wave_sum<16>(s1[r]), v2 = wave_sum<16>(s2[r]);
            if (n == 0) { s_red[(wave * 2 + 0) * 16 + 4 * g + r] = v1; s_red[(wave * 2 + 1) * 16 + 4 * g + r] = v2; }   // PAIR: [set][8]
        }
        __syncthreads();
        if (t < 32) {
            const int which = t >> 4, c = t & 15;
            const int cs_st = CHAIN ? a.cs_out2 : a.cs_out;
            if (c < cs_st) {
                float v = (s_red[(0 * 2 + which) * 16 + c] + s_red[(1 * 2 + which) * 16 + c]) +
                          (s_red[(2 * 2 + which) * 16 + c] + s_red[(3 * 2 + which) * 16 + c]);
                if (PAIR) v += (s_red[(0 * 2 + which) * 16 + c + 8] + s_red[(1 * 2 + which) * 16 + c + 8]) +
                               (s_red[(2 * 2 + which) * 16 + c + 8] + s_red[(3 * 2 + which) * 16 + c + 8]);
                a.stats_partial[(size_t)blockIdx.x * 2 * cs_st + which * cs_st + c] = v;
            }
        }
    }
    IMK_WGSTAMP_END();
    IMK_STAMP_END(1);
}

// =====================================================================================================
// forward / dgrad of the full- and half-resolution layers with 17-32 channels on either side (alpha in (1, 2]: the later
// generations of the IM+ width schedule, Cityscapes/11_Cityscapes_IM+.py:48; EvalNet's towers at ALPHA_EVALNET = 2).
// The same persistent, register-prefetching scheme as conv_pipe_kernel, generalised where that kernel is specialised:
// MT = 1 or 2 output-channel tiles per workgroup (accumulators [MT][4]), up to 4 input chunks per pixel (NC8), and the
// packed weights -- up to 2 x 9 fragments -- in LDS instead of registers (copied once per workgroup; one ds_read_b128 per
// channel tile and k-step), which keeps the kernel at 3-4 waves per SIMD.  No pair layout, chaining or fused weight
// gradient here.  Before round 2 these layers ran on the per-tile kernel (one tile per workgroup, weights re-copied per
// tile, no overlap of staging and MFMAs): 2.3x the step time from alpha 1 to 1.25 for 1.56x the flops.
// =====================================================================================================
// CHAIN2 (inference; round 3): the block's Conv1x1+ReLU (at most 32 output channels) as a second stage on the 3x3's output tile,
// which never leaves the chip.  Each wave parks its 64 pixels x 32 channels in a private LDS slab ([pixel][32 + 8] halfs: the
// 16-byte reads of 16 pixels hit 64 different banks) and reads them back as the B operand of one more MFMA per output tile --
// the k order of the 1x1's regular forward pack, so the result is bit-identical to the two launches.  Replaces the per-tile
// chain (conv_mfma_kernel<..., CHAIN>) for these widths: persistent workgroups, next tile's loads in flight, weights staged once.
template <int LM, int NC8, int MT, int EPI, bool DYSTAT, bool FULL, bool CHAIN2 = false>
__global__ __launch_bounds__(256) void conv_wide_kernel(ImkConvArgs a, int tiles_x, int tiles_y, int n_tiles,
                                                        unsigned magic_tx, ImkWalk wk) {
    constexpr int P = 4;
    constexpr int PS = imk_lds_pitch(NC8);
    constexpr int MAX_ITEMS = (18 * 18 * NC8 + 255) / 256;
    constexpr int MAX_NS = (9 * NC8 + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int ks3 = (a.ksize == 3);
    const int halo = ks3 ? 1 : 0;
    const int HT = 16 + 2 * halo, WT = TW + 2 * halo;
    const int n_items = HT * WT * NC8;
    const int nq = (ks3 ? 9 : 1) * NC8, ns = (nq + 3) / 4;
    uint8_t *s_tile = smem;
    float *s_aff = reinterpret_cast<float *>(smem + 18 * 18 * PS * 16);        // up to 4 x 32 floats (LM_UPADD)
    float *s_red = s_aff + 4 * 32;                                             // [4 waves][2][16 * MT]
    f16 *s_w = reinterpret_cast<f16 *>(s_red + 4 * 2 * 16 * MT);               // [MT][ns][512]
    constexpr int MIDP = 48;                                                   // halfs per pixel of the chain's slab: 6 chunks (imk_lds_pitch(4))
    const int t = threadIdx.x;
    f16 *s_mid = s_w + (size_t)MT * ns * 512 + (t >> 6) * 64 * MIDP;           // CHAIN2: this wave's [64 pixels][MIDP]
    IMK_STAMP_BEGIN(conv, 50000 + LM * 1000 + MT * 10 + (DYSTAT ? 1 : 0));
    const int lane = t & 63, wave = t >> 6, n = lane & 15, g = lane >> 4;
    const int H = a.H, W = a.W;
    const int cs_in = a.x.cs_in;
    const int per_img = tiles_x * tiles_y;

    int off[MAX_NS];
#pragma unroll
    for (int s = 0; s < MAX_NS; ++s) {
        const int q = 4 * s + g;
        const bool vq = (s < ns) && (q < nq);          // k-slots beyond the real taps carry zero weights: read offset 0
        const int tap = q / NC8, c8 = q - tap * NC8;
        const int ty = ks3 ? tap / 3 : 0, tx = ks3 ? tap - 3 * (tap / 3) : 0;
        off[s] = vq ? ((ty * WT + tx) * PS + c8) * 16 : 0;
    }
    int base[P];
#pragma unroll
    for (int p = 0; p < P; ++p) base[p] = ((wave * 4 + p) * WT + n) * PS * 16;
    int it_lds[MAX_ITEMS], it_py[MAX_ITEMS], it_px[MAX_ITEMS], it_c8[MAX_ITEMS];
#pragma unroll
    for (int k = 0; k < MAX_ITEMS; ++k) {
        const int i = t + 256 * k;
        const bool live = i < n_items;
        const int pix = (live ? i : t % n_items) / NC8;
        it_c8[k] = (live ? i : t % n_items) - pix * NC8;
        it_py[k] = pix / WT;
        it_px[k] = pix - it_py[k] * WT;
        it_lds[k] = live ? (pix * PS + it_c8[k]) * 16 : -1;
    }
    // packed weights -> LDS, once per workgroup (the per-tile kernel's fragment layout: [channel tile][k-step][lane][8])
    {
        const int n16 = MT * ns * 64;                   // 16-byte chunks
        const int mt_have = (a.cout + 15) / 16;
        for (int j = t; j < n16; j += 256) {
            const int m = j / (ns * 64);
            f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (m < mt_have) v = *reinterpret_cast<const f16x8 *>(a.wpk + (size_t)j * 8);
            *reinterpret_cast<f16x8 *>(s_w + (size_t)j * 8) = v;
        }
    }
    float bias[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = 16 * m + 4 * g + r;
            bias[m][r] = (EPI == EP_RELU && a.bias && co < a.cout) ? a.bias[co] : 0.f;
        }
    const bool want_stats = DYSTAT || ((EPI == EP_RELU) && a.stats_partial);      // CHAIN2 (training): statistics of the 1x1's output
    const int cs_st = CHAIN2 ? a.cs_out2 : a.cs_out;                               // channels of the tensor the statistics describe
    bool lane_out[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) lane_out[m] = 16 * m + 4 * g < a.cs_out;      // this lane's 4 channels exist in the output tensor
    // CHAIN2: the 1x1's A fragments (its regular forward pack: [channel tile][k-step 0][lane][8]), bias, output geometry
    f16x8 a2[2] = {f16x8{0, 0, 0, 0, 0, 0, 0, 0}, f16x8{0, 0, 0, 0, 0, 0, 0, 0}};
    float bias2[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    bool lane_out2[2] = {false, false};
    const unsigned cso2_b = CHAIN2 ? (unsigned)a.cs_out2 * 2u : 0u;
    if constexpr (CHAIN2) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            if (16 * m < a.cout2) a2[m] = *reinterpret_cast<const f16x8 *>(a.wpk2 + ((size_t)m * 64 + lane) * 8);
            lane_out2[m] = 16 * m + 4 * g < a.cs_out2;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int co = 16 * m + 4 * g + r; bias2[m][r] = co < a.cout2 ? a.bias2[co] : 0.f; }
        }
    }

    float s1[MT][4], s2[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[m][r] = s2[m][r] = 0.f;
    RawChunk<LM> raw[MAX_ITEMS];
    unsigned valid = 0;
    constexpr unsigned CSB = NC8 * 16;          // bytes per pixel of the (fp16) input tensor
    auto issue = [&](const PTile &tc) {         // addresses: see PTile / PSrc above conv_pipe_kernel
        const PSrc<LM> src = psrc_of<LM, CSB>(a.x, tc, halo, HT, WT, H, W);
        valid = 0;
#pragma unroll
        for (int k = 0; k < MAX_ITEMS; ++k) {
            const bool in_img = psrc_load<LM, CSB>(src, it_py[k], it_px[k], it_c8[k], W, raw[k]);
            valid |= ((it_lds[k] >= 0 && in_img) ? 1u : 0u) << k;
        }
    };
    // this lane's output pixels (tile row wave * 4 + p, column n) as byte offsets from the tile origin (channel 0): constant
    // over full tiles
    const unsigned cso_b = (unsigned)a.cs_out * 2u;
    unsigned o1c[P];
#pragma unroll
    for (int p = 0; p < P; ++p) o1c[p] = (unsigned)((wave * 4 + p) * W + n) * cso_b;

    // this workgroup's tiles: tile, tile + wk.step, ... inside its group's range (ImkWalk, imk_stage.h); a workgroup without a
    // tile (possible when the ranges are uneven) prefetches the launch's last tile and writes zero statistics rows
    const int wgrp = imk_walk_range(wk, blockIdx.x);
    int tile = wgrp * wk.chunk + (int)(blockIdx.x >> wk.shift);
    const int tile_end = min(n_tiles, (wgrp + 1) * wk.chunk);
    const int tile0 = min(tile, n_tiles - 1);
    PTile tc = ptile_at(tile0 / per_img, tile0 % per_img, tiles_x, magic_tx);
    issue(tc);
    stage_affine_table(a.x, s_aff);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): weights, biases and the first tile are in
    __syncthreads();                      // affine table and weights visible
    while (tile < tile_end) {
#pragma unroll
        for (int k = 0; k < MAX_ITEMS; ++k) {
            if (it_lds[k] >= 0) {
                f16x8 v = raw_transform<LM>(raw[k], s_aff, cs_in, it_c8[k], a.x.cin, a.x.u8_div);
                if (!(valid & (1u << k))) v = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                *reinterpret_cast<f16x8 *>(s_tile + it_lds[k]) = v;
            }
        }
        __syncthreads();
        const char *b_o1 = CHAIN2 ? pix_base(a.out2, tc.b, H, W, tc.ty0, tc.tx0, cso2_b) : pix_base(a.out, tc.b, H, W, tc.ty0, tc.tx0, cso_b);
        const char *b_mk = EPI == EP_MASK ? pix_base(a.mask, tc.b, H, W, tc.ty0, tc.tx0, cso_b) : nullptr;
        const char *b_zq = DYSTAT ? pix_base(a.dystat_z, tc.b, H, W, tc.ty0, tc.tx0, cso_b) : nullptr;
        unsigned o1[P];        // byte offset of the lane's pixels in the tensor this launch writes (CHAIN2: the 1x1's output)
        unsigned o_mid[P];     // CHAIN2 in training: the same pixels in the 3x3's own output tensor
        const char *b_mid = (CHAIN2 && a.out) ? pix_base(a.out, tc.b, H, W, tc.ty0, tc.tx0, cso_b) : nullptr;
        bool inb[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if constexpr (FULL) {
                o1[p] = CHAIN2 ? (unsigned)((wave * 4 + p) * W + n) * cso2_b : o1c[p]; inb[p] = true;
                o_mid[p] = o1c[p];
            } else {        // partial tiles clamp the coordinates used for LOADS (stores are guarded)
                const int my = H - 1 - tc.ty0, mx = W - 1 - tc.tx0, r = wave * 4 + p;
                inb[p] = r <= my && n <= mx;
                o1[p] = __umul24(__umul24(min(r, my), W) + min(n, mx), CHAIN2 ? cso2_b : cso_b);
                o_mid[p] = __umul24(__umul24(min(r, my), W) + min(n, mx), cso_b);
            }
        }
        f16x4 mk[MT][P], zq[MT][P];
        if (EPI == EP_MASK) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int p = 0; p < P; ++p) mk[m][p] = *reinterpret_cast<const f16x4 *>(b_mk + o1[p] + (lane_out[m] ? 32 * m + 8 * g : 0));
        }
        if (DYSTAT) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int p = 0; p < P; ++p) zq[m][p] = *reinterpret_cast<const f16x4 *>(b_zq + o1[p] + (lane_out[m] ? 32 * m + 8 * g : 0));
        }
        const int next = tile + wk.step;
        const PTile tn = next < tile_end ? ptile_next(tc, wk.q, wk.r, per_img, tiles_x, magic_tx) : tc;
        issue(tn);                                // in flight during the MFMAs, the epilogue and its stores
        f32x4 acc[MT][P];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int p = 0; p < P; ++p) acc[m][p] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < MAX_NS; ++s) {
            if (s < ns) {
                f16x8 afm[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) afm[m] = *reinterpret_cast<const f16x8 *>(s_w + ((size_t)(m * ns + s) * 64 + lane) * 8);
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const f16x8 bf = *reinterpret_cast<const f16x8 *>(s_tile + base[p] + off[s]);
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afm[m], bf, acc[m][p], 0, 0, 0);
                }
            }
        }
        if constexpr (CHAIN2) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    f16x4 v = {0, 0, 0, 0};
                    if (m < MT) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = (f16)fmaxf(acc[m < MT ? m : 0][p][r] + bias[m < MT ? m : 0][r], 0.f);
                    }
                    *reinterpret_cast<f16x4 *>(s_mid + (p * 16 + n) * MIDP + 16 * m + 4 * g) = v;
                    if (a.out && m < MT && (FULL || inb[p]) && lane_out[m < MT ? m : 0])     // training: the backward pass reads the 3x3's output
                        *reinterpret_cast<f16x4 *>(const_cast<char *>(b_mid) + o_mid[p] + 32 * m + 8 * g) = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const f16x8 bf2 = *reinterpret_cast<const f16x8 *>(s_mid + (p * 16 + n) * MIDP + 8 * g);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    if (16 * m < a.cs_out2) {
                        const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[m], bf2, f32x4{0, 0, 0, 0}, 0, 0, 0);
                        f16x4 v;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = (f16)fmaxf(d[r] + bias2[m][r], 0.f);
                        if ((FULL || inb[p]) && lane_out2[m]) {
                            *reinterpret_cast<f16x4 *>(const_cast<char *>(b_o1) + o1[p] + 32 * m + 8 * g) = v;
                            if (want_stats && m < MT) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) { const float f = (float)v[r]; s1[m < MT ? m : 0][r] += f; s2[m < MT ? m : 0][r] += f * f; }
                            }
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int p = 0; p < P; ++p) {
                f16x4 v;
                if (EPI == EP_RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (f16)fmaxf(acc[m][p][r] + bias[m][r], 0.f);
                } else if (EPI == EP_MASK) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = ((float)mk[m][p][r] > 0.f) ? (f16)acc[m][p][r] : (f16)0.f;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (f16)acc[m][p][r];
                }
                if ((FULL || inb[p]) && lane_out[m]) {
                    *reinterpret_cast<f16x4 *>(const_cast<char *>(b_o1) + o1[p] + 32 * m + 8 * g) = v;
                    if (DYSTAT) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float f = (float)v[r]; s1[m][r] += f; s2[m][r] += f * (float)zq[m][p][r]; }
                    } else if (want_stats) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float f = (float)v[r]; s1[m][r] += f; s2[m][r] += f * f; }
                    }
                }
            }
        }
        __syncthreads();   // tile reads done: the LDS tile may be overwritten
        tile = next;
        tc = tn;
    }
    if (want_stats) {      // one partial row per workgroup
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v1 = wave_sum<16>(s1[m][r]), v2 = wave_sum<16>(s2[m][r]);
                if (n == 0) {
                    s_red[(wave * 2 + 0) * 16 * MT + 16 * m + 4 * g + r] = v1;
                    s_red[(wave * 2 + 1) * 16 * MT + 16 * m + 4 * g + r] = v2;
                }
            }
        __syncthreads();
        if (t < 2 * 16 * MT) {
            const int which = t / (16 * MT), c = t - which * 16 * MT;
            if (c < cs_st) {
                const float v = (s_red[(0 * 2 + which) * 16 * MT + c] + s_red[(1 * 2 + which) * 16 * MT + c]) +
                                (s_red[(2 * 2 + which) * 16 * MT + c] + s_red[(3 * 2 + which) * 16 * MT + c]);
                a.stats_partial[(size_t)blockIdx.x * 2 * cs_st + which * cs_st + c] = v;
            }
        }
    }
    IMK_STAMP_END(1);
}

// =====================================================================================================
// wgrad
// =====================================================================================================

// LM = how the conv's input is materialised (same modes as the forward).  Persistent over tiles with the next
// tile's global loads issued into registers before the MFMAs of the current one (same scheme as conv_pipe_kernel).
struct ImkWgradGeom { int tiles_x, tiles_y, n_tiles, cit_n, cot_n, nci, nco; ImkWalk wk; };

// bx = split (walks tiles bx, bx + nbx, ...), by = (input-channel tile, output-channel tile) pair of nby
// The prefetch loads are unconditional (clamped coordinates, idle slots repeat slot 0) and BNB (BatchNorm backward on the
// gradient operand: a second tensor to load) is a template parameter: with a load inside a lane- or launch-dependent
// branch the compiler protects the reuse of its destination registers with `s_waitcnt vmcnt(0)` in front of EVERY load of
// the next tile, i.e. the seven loads of a prefetch went out one memory latency after the other (cycle stamps: 3.6-4.3 k
// of a workgroup's 8 k cycles per tile).
// KS3: 3x3 conv (9 tap accumulators + the one of the bias gradient) or 1x1 (2): the 1x1 form needs 32 registers and 32 KB of
// LDS less -- these kernels share the chip with the backward chain, whose next launch can only start where registers and LDS
// are left
template <int LM, bool BNB, bool KS3>
__device__ __forceinline__ void wgrad_mfma_body(const ImkWgradArgs &a, const ImkWgradGeom &gm, int bx, int by, int nbx, int nby) {
    const int tiles_x = gm.tiles_x, tiles_y = gm.tiles_y, n_tiles = gm.n_tiles, cit_n = gm.cit_n, cot_n = gm.cot_n;
    const int nc8_in = gm.nci, nc8_out = gm.nco;
    (void)tiles_y; (void)cit_n;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int ks3 = KS3 ? 1 : 0;
    constexpr int NACC = KS3 ? 10 : 2;          // taps + bias
    const int halo = ks3 ? 1 : 0;
    const int HT = 16 + 2 * halo, WT = TW + 2 * halo;
    constexpr int T = KS3 ? 9 : 1;
    f16 *s_x = reinterpret_cast<f16 *>(smem);
    f16 *s_d = s_x + 18 * 18 * WG_STRIDE_H;
    float *s_aff = reinterpret_cast<float *>(s_d + 256 * WG_STRIDE_H);
    const int t = threadIdx.x;
    const int pair = by;
    const int cit = pair / cot_n, cot = pair - cit * cot_n;
    const int H = a.H, W = a.W;
    const int lane = t & 63, wave = t >> 6, g = lane >> 4, i16 = lane & 15, qq = i16 >> 2, pp = i16 & 3;

    IMK_STAMP_BEGIN(conv, 30000 + LM * 10 + (BNB ? 1 : 0));
    stage_affine_table(a.x, s_aff);
    float *s_coef = s_aff + 4 * a.x.cs_in;          // [A | B | C] of the dA-side BatchNorm backward (optional)
    constexpr bool bnbwd = BNB;
    if (bnbwd)
        for (int i = t; i < 3 * a.cs_out; i += 256) s_coef[i] = a.dA_coef[i];

    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    f16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (f16)(i16 == 0 ? 1.0f : 0.0f);

    // staging items of this thread (constant over tiles): up to 3 chunks of the x slice, 2 of the dA slice.  A slice
    // is 16 channels = 2 chunks per pixel; when the tensor has only one chunk there (8-channel layers: level 0 at
    // alpha = 0.5, where most of the pixels are), items are pixels, so every thread carries live items and the
    // always-zero second chunk of the LDS slices is written once, before the loop.
    constexpr int NX = 3, ND = 2;
    const int cx = min(2, nc8_in - 2 * cit), cd = min(2, nc8_out - 2 * cot);   // live chunks per pixel: 1 or 2
    const int n_x = HT * WT * cx, n_d = 256 * cd;
    int x_lds[NX], x_py[NX], x_px[NX], x_c8[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        const int i = t + 256 * k;
        const int ii = i < n_x ? i : t;      // idle slots repeat slot 0 (t < n_x always): their load is a cache hit
        const int pix = cx == 2 ? ii >> 1 : ii, ch = cx == 2 ? (ii & 1) : 0;
        x_c8[k] = 2 * cit + ch;
        x_py[k] = pix / WT;
        x_px[k] = pix - x_py[k] * WT;
        x_lds[k] = (i < n_x) ? pix * WG_STRIDE_H + ch * 8 : -1;
    }
    int d_lds[ND], d_py[ND], d_px[ND], d_c8[ND];
#pragma unroll
    for (int k = 0; k < ND; ++k) {
        const int i = t + 256 * k;
        const int ii = i < n_d ? i : t;
        const int pix = cd == 2 ? ii >> 1 : ii, ch = cd == 2 ? (ii & 1) : 0;
        d_c8[k] = 2 * cot + ch;
        d_py[k] = pix >> 4;
        d_px[k] = pix & 15;
        d_lds[k] = (i < n_d) ? pix * WG_STRIDE_H + ch * 8 : -1;
    }
    // the dead second chunks read as zeros for every tile (nothing else ever writes them)
    if (cx < 2)
        for (int i = t; i < 18 * 18; i += 256) *reinterpret_cast<f16x8 *>(s_x + i * WG_STRIDE_H + 8) = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (cd < 2) *reinterpret_cast<f16x8 *>(s_d + t * WG_STRIDE_H + 8) = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    RawChunk<LM> xr[NX];
    f16x8 dr[ND], dz[ND];
    unsigned vx = 0, vd = 0;
    auto issue = [&](int tile) {
        const TileCoord tc = tile_coord(tile, tiles_x, tiles_y, 16);
        vx = vd = 0;
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int y = tc.ty0 + x_py[k] - halo, x = tc.tx0 + x_px[k] - halo;
            const bool ok = x_lds[k] >= 0 && y >= 0 && y < H && x >= 0 && x < W;
            raw_load<LM>(a.x, tc.b, min(max(y, 0), H - 1), min(max(x, 0), W - 1), H, W, x_c8[k], xr[k]);
            vx |= (ok ? 1u : 0u) << k;
        }
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            const int y = tc.ty0 + d_py[k], x = tc.tx0 + d_px[k];
            const bool ok = d_lds[k] >= 0 && y < H && x < W;
            const size_t o = ((size_t)(tc.b * H + min(y, H - 1)) * W + min(x, W - 1)) * a.cs_out + d_c8[k] * 8;
            dr[k] = *reinterpret_cast<const f16x8 *>(a.dA + o);
            if (bnbwd) dz[k] = *reinterpret_cast<const f16x8 *>(a.dA_z + o);
            vd |= (ok ? 1u : 0u) << k;
        }
    };

    // this split's tiles: an XCD-aware walk over the split index (ImkWalk, imk_stage.h; nbx is a multiple of 8 there, so
    // bx & 7 is the block's XCD group whatever by is)
    const ImkWalk wk = gm.wk;
    const int wgrp = imk_walk_range(wk, (unsigned)bx);
    int tile = wgrp * wk.chunk + (bx >> wk.shift);
    const int tile_end = min(n_tiles, (wgrp + 1) * wk.chunk);
    issue(tile < n_tiles ? tile : n_tiles - 1);
    __syncthreads();   // affine table visible
    while (tile < tile_end) {
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            if (x_lds[k] >= 0) {
                f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (vx & (1u << k)) v = raw_transform<LM>(xr[k], s_aff, a.x.cs_in, x_c8[k], a.x.cin, a.x.u8_div);
                *reinterpret_cast<f16x8 *>(s_x + x_lds[k]) = v;
            }
        }
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            if (d_lds[k] < 0) continue;
            f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (vd & (1u << k)) {
                v = dr[k];
                if (bnbwd) {
                    const float *A = s_coef + d_c8[k] * 8, *Bc = A + a.cs_out, *Cc = Bc + a.cs_out;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float zf = (float)dz[k][j];
                        v[j] = zf > 0.f ? (f16)(A[j] * (float)dr[k][j] + Bc[j] * zf + Cc[j]) : (f16)0.f;
                    }
                }
            }
            *reinterpret_cast<f16x8 *>(s_d + d_lds[k]) = v;
        }
        __syncthreads();
        const int next = tile + wk.step;
        issue(next < tile_end ? next : tile);    // in flight during the MFMAs below (the last one re-reads this tile)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int r0 = 2 * (wave + 4 * kk);          // tile rows r0, r0+1 form this k-step's 32 pixels
            const int row = r0 + (g >> 1);
            // k-slot <-> pixel map (any bijection works as long as both operands use it): lane group g's elements 0-3 are
            // pixels x = 4(g&1) + 0..3 and elements 4-7 the pixels 8 further right.  The two groups of a 32-lane half then
            // read 8 consecutive pixels of one row = 256 contiguous bytes = every LDS bank once (the previous map,
            // x = 8(g&1) + 0..3 at a 48-byte pitch, put two pixels of a half on the same banks: SQ_LDS_BANK_CONFLICT equal to
            // SQ_ACTIVE_INST_LDS in profiles/r01_sq_counters.csv).
            const int xx = 4 * (g & 1) + qq;             // +8 for the second read
            // B operand: dA[pixel][co]
            const f16 *pb = s_d + (row * 16 + xx) * WG_STRIDE_H + 4 * pp;
            const h4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(LDS_PTR(h4, pb));
            const h4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(LDS_PTR(h4, pb + 8 * WG_STRIDE_H));
            f16x8 bf;
#pragma unroll
            for (int e = 0; e < 4; ++e) { bf[e] = (f16)b0[e]; bf[4 + e] = (f16)b1[e]; }
            const f16 *pa = s_x + (row * WT + xx) * WG_STRIDE_H + 4 * pp;
#pragma unroll
            for (int tap = 0; tap < T; ++tap) {
                {
                    const int ty = ks3 ? tap / 3 : 0, tx = ks3 ? tap % 3 : 0;
                    const f16 *p = pa + (ty * WT + tx) * WG_STRIDE_H;
                    const h4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(LDS_PTR(h4, p));
                    const h4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(LDS_PTR(h4, p + 8 * WG_STRIDE_H));
                    f16x8 af;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { af[e] = (f16)a0[e]; af[4 + e] = (f16)a1[e]; }
                    acc[tap] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, acc[tap], 0, 0, 0);
                }
            }
            acc[NACC - 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, bf, acc[NACC - 1], 0, 0, 0);  // column sums -> bias grad
        }
        __syncthreads();   // tile reads done before the next tile overwrites LDS
        tile = next;
    }
    // ---- reduce the 4 waves' accumulators through LDS, write this workgroup's partial ----------------
    float *s_acc = reinterpret_cast<float *>(smem);  // [4][NACC][256]
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s_acc[(wave * NACC + i) * 256 + r * 64 + lane] = acc[i][r];
    __syncthreads();
    float *dst = a.partial + ((size_t)bx * nby + pair) * (T + 1) * 256;
    for (int i = 0; i <= T; ++i)      // row T = the bias gradient
        dst[i * 256 + t] = s_acc[(0 * NACC + i) * 256 + t] + s_acc[(1 * NACC + i) * 256 + t] +
                           s_acc[(2 * NACC + i) * 256 + t] + s_acc[(3 * NACC + i) * 256 + t];
    IMK_STAMP_END(1);
}

template <int LM, bool BNB, bool KS3>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(ImkWgradArgs a, ImkWgradGeom gm) {
    wgrad_mfma_body<LM, BNB, KS3>(a, gm, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y);
}


// ---- the input block's weight gradient as a streaming reduction (round 5) ------------------------------------------------------
// dW[ci][co] = sum_px x[px][ci] * dA[px][co] for the Conv1x1 on the uint8 image (unet.py:5-6: 1-4 input channels, <= 8 outputs),
// dA = the BatchNorm backward + ReLU mask of (dy, z) formed on load -- the LAST weight gradient of a step, alone on the chip with the
// optimizer waiting behind it.  The MFMA kernel staged 256-pixel tiles through LDS for a 3 x 8 result and moved its 70 MB at 2.9 TB/s;
// here a thread walks pixels (two in flight), keeps the 4 x 8 + 8 sums in registers, and a workgroup leaves one partial row in the
// layout wgf_stage1 / 2 expect (element r * 64 + l <-> ci = 4 (l >> 4) + r, co = l & 15; the bias row's element co).  Same operands
// as the matrix-core form (x and dA rounded to fp16, products and sums in fp32), another summation order.
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const uint8_t *__restrict__ x, int cin, float u8_div, const f16 *__restrict__ dy,
                                                         const f16 *__restrict__ z, const float *__restrict__ coef /*[3][8]*/,
                                                         long long n_pix, float *__restrict__ partial) {
    __shared__ float s_c[3][8];
    __shared__ float s_red[4][40];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t < 24) s_c[t >> 3][t & 7] = coef[t];
    __syncthreads();
    float A[8], Bc[8], Cc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { A[j] = s_c[0][j]; Bc[j] = s_c[1][j]; Cc[j] = s_c[2][j]; }
    float acc[4][8], bsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { bsum[j] = 0.f; acc[0][j] = acc[1][j] = acc[2][j] = acc[3][j] = 0.f; }
    const float inv = 1.0f / u8_div;
    const long long stride = (long long)gridDim.x * 256;
    struct Px { f16x8 d, zz; uint32_t xb; };
    auto load = [&](long long p, Px &q) {
        q.d = *reinterpret_cast<const f16x8 *>(dy + p * 8);
        q.zz = *reinterpret_cast<const f16x8 *>(z + p * 8);
        // one unaligned dword per pixel (byte loads made this gradient issue-bound: imk_stage.h); the tensor's last pixels take theirs a
        // few bytes early, so nothing behind the images is read
        typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
        const long long rem = (n_pix - p) * cin;
        const int back = rem >= 4 ? 0 : (int)(4 - rem);
        const uint32_t v = *reinterpret_cast<const u32_unaligned *>(x + p * cin - back) >> (8 * back);
        q.xb = cin >= 4 ? v : (v & ((1u << (8 * cin)) - 1u));
    };
    auto use = [&](const Px &q) {
        float xf[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) xf[c] = (float)(f16)((float)((q.xb >> (8 * c)) & 0xffu) * inv);      // == x / u8_div in fp16 for every byte (imk_stage.h)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float zf = (float)q.zz[j];
            const float g = zf > 0.f ? (float)(f16)(A[j] * (float)q.d[j] + Bc[j] * zf + Cc[j]) : 0.f;
            bsum[j] += g;
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c][j] += xf[c] * g;
        }
    };
    long long p = (long long)blockIdx.x * 256 + t;
    if (p < n_pix) {
        Px cur, nxt;
        load(p, cur);
        while (true) {
            const long long pn = p + stride;
            const bool more = pn < n_pix;
            load(more ? pn : p, nxt);
            use(cur);
            if (!more) break;
            cur = nxt;
            p = pn;
        }
    }
    // lanes (butterfly), then the four waves in order: fixed association
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = wave_sum<64>(acc[c][j]);
            if (lane == 0) s_red[wave][c * 8 + j] = v;
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = wave_sum<64>(bsum[j]);
        if (lane == 0) s_red[wave][32 + j] = v;
    }
    __syncthreads();
    auto tot = [&](int i) { return (s_red[0][i] + s_red[1][i]) + (s_red[2][i] + s_red[3][i]); };
    float *row = partial + (size_t)blockIdx.x * 2 * 256;
    const int r = t >> 6, l = t & 63, ci = 4 * (l >> 4) + r, co = l & 15;
    row[t] = (ci < 4 && co < 8) ? tot(ci * 8 + co) : 0.f;
    row[256 + t] = t < 8 ? tot(32 + t) : 0.f;
}

// ---- batched, deterministic reduction of all layers' weight-gradient partials -----------------------------
__device__ __forceinline__ int wgf_find_job(const ImkWgFinalJobs &jobs, int idx, bool stage1) {
    int j = 0;
    for (int k = 1; k < jobs.n; ++k) {
        const int begin = stage1 ? jobs.j[k].work1_begin : jobs.j[k].tile_begin;
        if (idx >= begin) j = k;
    }
    return j;
}

// Round 5: 16-byte loads.  A (tile, chunk) work item is 256 floats x `chunk` splits; thread (e4 = t & 63, sg = t >> 6) sums the
// float4 e4 of splits sg, sg + 4, ... (four 16-byte loads in flight per 16 splits instead of sixteen 4-byte ones: a wave's load is
// 1 KB, not 256 B), the four sub-sums are combined in a fixed order through LDS.  Deterministic; another association than rounds 2-4.
__global__ __launch_bounds__(256) void wgf_stage1_kernel(ImkWgFinalJobs jobs) {
    IMK_STAMP_BEGIN(conv, 21);
    __shared__ float4 s_q[4][64];
    const int jn = wgf_find_job(jobs, blockIdx.x, true);
    const ImkWgFinalJob &jb = jobs.j[jn];
    const int local = blockIdx.x - jb.work1_begin;
    const int tile = local / jb.n_chunks, chunk = local - tile * jb.n_chunks;
    const int t = threadIdx.x, e4 = t & 63, sg = t >> 6;
    const int s0 = chunk * jb.chunk, s1 = min(jb.n_split, s0 + jb.chunk);    // jb.chunk: a multiple of WG_RED_CHUNK
    const size_t stride4 = (size_t)jb.n_tiles * 64;
    const float4 *p = reinterpret_cast<const float4 *>(jb.partial) + (size_t)tile * 64 + e4;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int sb = s0; sb < s1; sb += WG_RED_CHUNK) {
        float4 v[WG_RED_CHUNK / 4];
#pragma unroll
        for (int i = 0; i < WG_RED_CHUNK / 4; ++i) v[i] = p[(size_t)min(sb + sg + 4 * i, s1 - 1) * stride4];   // unconditional: all in flight
#pragma unroll
        for (int i = 0; i < WG_RED_CHUNK / 4; ++i) {
            const float live = (sb + sg + 4 * i < s1) ? 1.f : 0.f;
            acc.x += live * v[i].x; acc.y += live * v[i].y; acc.z += live * v[i].z; acc.w += live * v[i].w;
        }
    }
    s_q[sg][e4] = acc;
    __syncthreads();
    if (sg == 0) {
        const float4 a = s_q[0][e4], b = s_q[1][e4], c = s_q[2][e4], d = s_q[3][e4];
        float4 r;
        r.x = (a.x + b.x) + (c.x + d.x); r.y = (a.y + b.y) + (c.y + d.y); r.z = (a.z + b.z) + (c.z + d.z); r.w = (a.w + b.w) + (c.w + d.w);
        reinterpret_cast<float4 *>(jb.red)[((size_t)chunk * jb.n_tiles + tile) * 64 + e4] = r;
    }
    IMK_STAMP_END(1);
}

// Stage 2: one 1024-thread block per tile, sub-group sg = t >> 8 sums chunks sg, sg + 4, ... (at most 16 each).  Round 5: only as
// many loads as the job has chunks -- the deep layers of the wide nets have thousands of tiles with 1-4 chunks each, and sixteen
// clamped loads per thread for every one of them made this launch 5 % of a Cityscapes alpha = 2 step (42 us per launch).
template <int N>
__device__ __forceinline__ float wgf_sum_chunks(const float *p, size_t stride, int sg, int n_chunks) {
    float v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = p[(size_t)min(sg + 4 * i, n_chunks - 1) * stride];   // unconditional loads
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < N; ++i) acc += (sg + 4 * i < n_chunks) ? v[i] : 0.f;
    return acc;
}

__global__ __launch_bounds__(1024) void wgf_stage2_kernel(ImkWgFinalJobs jobs, const float *__restrict__ inv_scale_ptr,
                                                          float *__restrict__ found_inf) {
    __shared__ float s_p[4][256];
    const int jn = wgf_find_job(jobs, blockIdx.x, false);
    const ImkWgFinalJob &jb = jobs.j[jn];
    const int tile = blockIdx.x - jb.tile_begin;
    const int e = threadIdx.x & 255, sg = threadIdx.x >> 8;
    const size_t stride = (size_t)jb.n_tiles * 256;
    const float *p = jb.red + (size_t)tile * 256 + e;
    const int per = (jb.n_chunks + 3) >> 2;            // chunks per sub-group (uniform over the block); same order of additions for any N >= per
    float acc;
    if (per <= 1) acc = wgf_sum_chunks<1>(p, stride, sg, jb.n_chunks);
    else if (per <= 2) acc = wgf_sum_chunks<2>(p, stride, sg, jb.n_chunks);
    else if (per <= 4) acc = wgf_sum_chunks<4>(p, stride, sg, jb.n_chunks);
    else if (per <= 8) acc = wgf_sum_chunks<8>(p, stride, sg, jb.n_chunks);
    else acc = wgf_sum_chunks<16>(p, stride, sg, jb.n_chunks);
    s_p[sg][e] = acc;
    __syncthreads();
    if (sg != 0) return;
    float s = ((s_p[0][e] + s_p[1][e]) + (s_p[2][e] + s_p[3][e])) * *inv_scale_ptr;
    const int pair = tile / (jb.T + 1), tap = tile - pair * (jb.T + 1);
    const int cit = pair / jb.cot_n, cot = pair - cit * jb.cot_n;
    const int r = e >> 6, lane = e & 63;
    const int m = 4 * (lane >> 4) + r, nn = lane & 15;
    const int ci = cit * 16 + m, co = cot * 16 + nn;
    if (tap < jb.T) {
        if (ci < jb.cin && co < jb.cout) {
            if (!isfinite(s)) *found_inf = 1.0f;
            jb.dw[((size_t)tap * jb.cin + ci) * jb.cout + co] = s;
        }
    } else if (cit == 0 && m == 0 && co < jb.cout) {
        if (!isfinite(s)) *found_inf = 1.0f;
        jb.db[co] = s;
    }
}

// all conv layers of a model in one launch: blockIdx.y = job
__global__ __launch_bounds__(256) void pack_conv_batched_kernel(ImkPackJobs jobs) {
    IMK_STAMP_BEGIN(conv, 23);
    if (jobs.ctl && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) imk_ctl_end_step(jobs.ctl, jobs.stats);
    const ImkPackJob &jb = jobs.j[blockIdx.y];
    if (jb.transposed == 2) {   // chain operand of a 1x1 conv: one k-step, lane (m, g), j < 4 <-> W[ci = 4g + j][co = m]
        for (int i = blockIdx.x * 256 + threadIdx.x; i < 512; i += gridDim.x * 256) {
            const int j = i & 7, lane = (i >> 3) & 63;
            const int m = lane & 15, g = lane >> 4;
            float v = 0.f;
            if (!jb.pair) {
                const int ci = 4 * g + j;
                if (j < 4 && ci < jb.cin && m < jb.cout) v = jb.w[(size_t)ci * jb.cout + m];
            } else {        // rows 0-7: row A's 8 outputs from k-slot groups 0,1; rows 8-15: row B's from groups 2,3
                const int ci = 4 * (g & 1) + j, co = m & 7;
                if (j < 4 && (m >> 3) == (g >> 1) && ci < jb.cin && co < jb.cout) v = jb.w[(size_t)ci * jb.cout + co];
            }
            jb.dst[i] = (f16)v;
        }
        return;
    }
    const int T = jb.ksize == 3 ? 9 : 1;
    const int m_dim = jb.transposed ? jb.cin : jb.cout, k_dim = jb.transposed ? jb.cout : jb.cin;
    if (jb.pair && T == 9) {    // 3x3, pair layout with shared taps (conv_pipe_kernel): k-step s, group g carry slot q = 4s + g =
        // (window row r = 0..3, column c, chunk) of the 4 x 3 window the pair's two pixel rows read; accumulator rows 0-7 are the
        // upper pixel row's channels (kernel row r), rows 8-15 the lower one's (kernel row r - 1): zero where that is outside 0..2
        const int nc8 = ((k_dim + 7) & ~7) / 8;
        const int ns = 3 * nc8;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < ns * 512; i += gridDim.x * 256) {
            const int j = i & 7, lane = (i >> 3) & 63, st = i >> 9;
            const int m = lane & 15, g = lane >> 4;
            const int q = 4 * st + g;
            const int ut = q / nc8, c8 = q - ut * nc8;
            const int r = ut / 3, c = ut - 3 * r, ky = r - (m >> 3);
            const int mi = m & 7, ki = c8 * 8 + j;
            float v = 0.f;
            if (ky >= 0 && ky <= 2 && mi < m_dim && ki < k_dim) {
                const int tap = ky * 3 + c;
                if (!jb.transposed) v = jb.w[((size_t)tap * jb.cin + ki) * jb.cout + mi];
                else v = jb.w[((size_t)(T - 1 - tap) * jb.cin + mi) * jb.cout + ki];
            }
            jb.dst[i] = (f16)v;
        }
        return;
    }
    if (jb.pair) {              // 1x1: one 16-row block, k-step s: group g carries chunk q = 2s + (g & 1) of row g >> 1
        const int nc8 = ((k_dim + 7) & ~7) / 8;
        const int nq = T * nc8, ns = (nq + 1) / 2;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < ns * 512; i += gridDim.x * 256) {
            const int j = i & 7, lane = (i >> 3) & 63, st = i >> 9;
            const int m = lane & 15, g = lane >> 4;
            const int q = 2 * st + (g & 1);
            const int tap = q / nc8, c8 = q - tap * nc8;
            const int mi = m & 7, ki = c8 * 8 + j;
            float v = 0.f;
            if (q < nq && (m >> 3) == (g >> 1) && mi < m_dim && ki < k_dim) {
                if (!jb.transposed) v = jb.w[((size_t)tap * jb.cin + ki) * jb.cout + mi];
                else v = jb.w[((size_t)(T - 1 - tap) * jb.cin + mi) * jb.cout + ki];
            }
            jb.dst[i] = (f16)v;
        }
        return;
    }
    const int nc8 = ((k_dim + 7) & ~7) / 8;
    const int nc8p = imk_pass_chunks(nc8), nsp = (T * nc8p + 3) / 4;
    const int ns = imk_cdiv_d(nc8, nc8p) * nsp;
    const int total = ((m_dim + 15) / 16) * ns * 512;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int j = i & 7, lane = (i >> 3) & 63;
        const int cs = i >> 9;
        const int s = cs % ns, ct = cs / ns;
        const int m = lane & 15, g = lane >> 4;
        const int pass = s / nsp, q = 4 * (s - pass * nsp) + g;
        const int tap = q / nc8p, c8 = pass * nc8p + (q - tap * nc8p);
        const int mi = ct * 16 + m, ki = c8 < nc8 ? c8 * 8 + j : k_dim;
        float v = 0.f;
        if (tap < T && mi < m_dim && ki < k_dim) {
            if (!jb.transposed) v = jb.w[((size_t)tap * jb.cin + ki) * jb.cout + mi];
            else v = jb.w[((size_t)(T - 1 - tap) * jb.cin + mi) * jb.cout + ki];
        }
        jb.dst[i] = (f16)v;
    }
    IMK_STAMP_END(1);
}

}  // namespace

int imk_wgf_add_job(ImkWgFinalJobs &jobs, float *partial, int n_split, int ksize, int cin, int cout, float *dw, float *db) {
    if (jobs.n >= IMK_WGF_MAX_JOBS) return IMK_EUNSUPPORTED;
    const int T = ksize == 3 ? 9 : 1;
    const int cit_n = (imk_pad8(cin) + 15) / 16, cot_n = (imk_pad8(cout) + 15) / 16;
    ImkWgFinalJob &jb = jobs.j[jobs.n++];
    jb.partial = partial;
    jb.n_split = n_split;
    jb.chunk = WG_RED_CHUNK;
    while (imk_cdiv(n_split, jb.chunk) > 64) jb.chunk *= 2;      // stage 2 sums at most 64 chunks
    jb.n_chunks = imk_cdiv(n_split, jb.chunk);
    jb.n_tiles = cit_n * cot_n * (T + 1);
    jb.red = partial + (size_t)n_split * jb.n_tiles * 256;
    jb.T = T; jb.cin = cin; jb.cout = cout; jb.cot_n = cot_n;
    jb.dw = dw; jb.db = db;
    jb.work1_begin = jobs.total_work1;
    jb.tile_begin = jobs.total_tiles;
    jobs.total_work1 += jb.n_tiles * jb.n_chunks;
    jobs.total_tiles += jb.n_tiles;
    return IMK_OK;
}

int imk_launch_wgrad_finalize_jobs(const ImkWgFinalJobs &jobs, const float *inv_scale_ptr, float *found_inf, hipStream_t stream) {
    if (jobs.n <= 0) return IMK_OK;
    double wgf_bytes = 0;
    for (int i = 0; i < jobs.n; ++i)   // partials read once, chunk sums written and read once, gradients written
        wgf_bytes += ((double)jobs.j[i].n_split + 2.0 * jobs.j[i].n_chunks + 1.0) * jobs.j[i].n_tiles * 256 * 4;
    ImkProfScope prof(PF_WGF, wgf_bytes, stream);
    double b1 = 0, b2 = 0;            // per launch: stage 1 reads the partial rows and writes the chunk sums, stage 2 reads those and writes the gradients
    for (int i = 0; i < jobs.n; ++i) {
        b1 += ((double)jobs.j[i].n_split + jobs.j[i].n_chunks) * jobs.j[i].n_tiles * 256 * 4;
        b2 += ((double)jobs.j[i].n_chunks + 1.0) * jobs.j[i].n_tiles * 256 * 4;
    }
    imk_prof_work(b1);
    imk_klaunch(wgf_stage1_kernel, dim3(jobs.total_work1), dim3(256), 0, stream, jobs);
    IMK_LAUNCH_CHECK();
    imk_prof_work(b2);
    imk_klaunch(wgf_stage2_kernel, dim3(jobs.total_tiles), dim3(1024), 0, stream, jobs, inv_scale_ptr, found_inf);
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}

int imk_launch_pack_jobs(const ImkPackJobs &jobs, hipStream_t stream) {
    if (jobs.n <= 0) return IMK_OK;
    double pk_bytes = 0;
    for (int i = 0; i < jobs.n; ++i) pk_bytes += (double)jobs.j[i].ksize * jobs.j[i].ksize * jobs.j[i].cin * jobs.j[i].cout * 6;
    ImkProfScope prof(PF_STEP_TAIL, pk_bytes, stream);
    // blocks per job: the widest layer (147 k fragment slots at alpha = 0.5, 590 k at alpha = 1) then has 2-9 slots per thread --
    // the kernel sits at the very end of the step's dependent chain, and a slot is a chain of index divisions and one load
    // Round 5: wider nets get more blocks (EvalNet alpha 2 / Cityscapes alpha 2: 2.4 M slots in the widest layer = 144 per thread on
    // 64 blocks, 44 us per launch): about 8 slots per thread of the widest job, 64 ... 1024 blocks per job (a block whose job has
    // fewer slots leaves at once).  IMK_PACK_BLOCKS fixes the count.
    static const int bpj_env = []() { const char *e = getenv("IMK_PACK_BLOCKS"); return e ? atoi(e) : 0; }();
    int bpj = bpj_env;
    if (bpj <= 0) {
        long long widest = 0;
        for (int i = 0; i < jobs.n; ++i) {
            const long long slots = (long long)jobs.j[i].ksize * jobs.j[i].ksize * ((jobs.j[i].cin + 15) & ~15) * ((jobs.j[i].cout + 15) & ~15);
            widest = slots > widest ? slots : widest;
        }
        bpj = (int)((widest + 2047) / 2048);
        bpj = bpj < 64 ? 64 : (bpj > 1024 ? 1024 : bpj);
    }
    imk_klaunch(pack_conv_batched_kernel, dim3(dim3(bpj, jobs.n)), dim3(256), 0, stream, jobs);
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}

// -----------------------------------------------------------------------------------------------------
static inline int odd_ps(int nc8) { return imk_lds_pitch(nc8); }   // (name kept: the per-tile kernel's pitch; odd until round 4)

// Tile height: 16 rows unless the LDS tile would exceed 64 KB (wide layers), then 8.
static inline int conv_tile_h(int cs_in, int ksize) {
    const int halo = ksize == 3 ? 1 : 0;
    const size_t b16 = (size_t)(16 + 2 * halo) * (TW + 2 * halo) * odd_ps(imk_pass_chunks(cs_in / 8)) * 16;
    return b16 > 64 * 1024 ? 8 : 16;
}

int imk_conv_num_tiles(int B, int H, int W, int cs_in, int ksize) {
    const int per_tile = B * imk_cdiv(H, conv_tile_h(cs_in, ksize)) * imk_cdiv(W, TW);
    const int gemm = imk_conv_gemm_num_tiles(B, H, W);    // the GEMM-class kernel's rows (8 x 16 tiles), whichever kernel runs
    return per_tile > gemm ? per_tile : gemm;
}

// ---- optional per-launch event timing (bench.py roofline) ---------------------------------------------
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <cxxabi.h>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>
namespace {
struct ProfRec { hipEvent_t e0, e1; int variant; double bytes, flops; };
struct ProfTot { long long launches = 0; double bytes = 0, flops = 0; };
const char *const kFamilyNames[PF_COUNT] = {
    "conv_mfma_kernel<16, 1>", "conv_mfma_kernel<16, 2>", "conv_mfma_kernel<16, 4>", "conv_mfma_kernel<8, 1>", "conv_mfma_kernel<8, 2>",
    "conv_mfma_kernel<8, 4>", "conv_pipe_kernel", "wgrad_mfma_kernel", "bn_bwd_prep_kernel", "bn_bwd_coef_kernel", "bn_finalize_kernel",
    "wgf_stage1+2_kernel", "head_kernel", "head_loss_kernel", "step_tail", "im_kernel", "conv_gemm_kernel", "wgrad_gemm_kernel"};
// marker dispatch for kernel traces: its grid size (64 * id work-items) carries the id
__global__ void imk_mark_kernel(int id) { (void)id; }
}  // namespace
// Measurement context (include/imk.h: imk_prof_*): owned by the caller, bound to the thread that launches -- the library itself
// keeps no mutable global, only this per-thread binding
struct imk_prof {
    int period = 0;                    // time every k-th hooked launch
    long counter = 0;
    std::vector<ProfRec> recs;         // recorded launches since the last collect
    std::vector<hipEvent_t> pool;      // recycled events
    bool totals_on = false;            // sum every hooked launch per kernel name (imk_prof_totals_*)
    std::map<std::string, ProfTot> totals;
    hipEvent_t event() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e;
        (void)hipEventCreate(&e);
        return e;
    }
};
static thread_local imk_prof *t_prof = nullptr;

extern "C" int imk_prof_create(int period, imk_prof **out) {
    IMK_CHECK_ARG(out && period >= 0);
    imk_prof *p = new (std::nothrow) imk_prof();
    if (!p) return IMK_EINVAL;
    p->period = period;
    *out = p;
    return IMK_OK;
}
extern "C" void imk_prof_destroy(imk_prof *p) {
    if (!p) return;
    if (t_prof == p) { t_prof = nullptr; imk_tls_totals_on = false; }
    for (auto &r : p->recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (auto e : p->pool) (void)hipEventDestroy(e);
    delete p;
}
extern "C" int imk_prof_bind(imk_prof *p) { t_prof = p; imk_tls_totals_on = p && p->totals_on; return IMK_OK; }
extern "C" int imk_prof_unbind(imk_prof *p) { if (p && t_prof == p) { t_prof = nullptr; imk_tls_totals_on = false; return 1; } return 0; }
extern "C" int imk_prof_set_period(imk_prof *p, int period) { IMK_CHECK_ARG(p && period >= 0); p->period = period; return IMK_OK; }

// name of a kernel as rocprofv3's kernel trace prints it, minus "void ", the anonymous namespace and the parameter list (what
// profiles/summarize.py's short() leaves): resolved once per function through the runtime's own symbol table
static const std::string &kernel_short_name(const void *kern, hipStream_t stream) {
    static std::mutex mu;
    static std::unordered_map<const void *, std::string> names;
    std::lock_guard<std::mutex> lk(mu);
    auto it = names.find(kern);
    if (it != names.end()) return it->second;
    std::string n = "?";
    if (const char *m = hipKernelNameRefByPtr(kern, stream)) {
        int st = 0;
        char *d = abi::__cxa_demangle(m, nullptr, nullptr, &st);
        n = (st == 0 && d) ? d : m;
        free(d);
        if (st != 0) {      // (_Float16 parameters defeat the demangler -- rocprofv3's too: the plain name, as summarize.py's short() cuts it)
            const char *tag = "_ZN12_GLOBAL__N_1";
            const size_t at = n.find(tag);
            if (at != std::string::npos) {
                size_t i = at + strlen(tag), len = 0;
                while (i < n.size() && n[i] >= '0' && n[i] <= '9') len = 10 * len + (size_t)(n[i++] - '0');
                if (len > 0 && i + len <= n.size()) n = n.substr(i, len);
            }
        }
        for (const char *drop : {"void ", "(anonymous namespace)::"})
            for (size_t at; (at = n.find(drop)) != std::string::npos;) n.erase(at, strlen(drop));
        // cut the parameter list: the first '(' outside the template argument brackets
        int depth = 0;
        for (size_t i = 0; i < n.size(); ++i) {
            if (n[i] == '<') ++depth;
            else if (n[i] == '>') --depth;
            else if (n[i] == '(' && depth == 0) { n.resize(i); break; }
        }
        if (n.size() > 90) n.resize(90);
    }
    return names.emplace(kern, n).first->second;
}
static thread_local struct { double bytes, flops; bool armed; } t_work = {0.0, 0.0, false};
void imk_prof_work(double bytes, double flops) { t_work.bytes = bytes; t_work.flops = flops; t_work.armed = true; }
void imk_prof_note_launch(const void *kern, hipStream_t stream) {
    imk_prof *p = t_prof;
    if (!p || !p->totals_on) return;
    ProfTot &t = p->totals[kernel_short_name(kern, stream)];
    t.launches += 1;
    if (t_work.armed) { t.bytes += t_work.bytes; t.flops += t_work.flops; t_work.armed = false; }
}

int imk_prof_begin(int family, double bytes, hipStream_t stream, double flops, const char *variant) {
    imk_prof *p = t_prof;
    (void)variant;
    if (p && p->totals_on && family >= 0 && family < PF_COUNT) imk_prof_work(bytes, flops);      // the scope's next launch takes it
    if (!p || p->period <= 0 || (p->counter++ % p->period) != 0) return -1;
    ProfRec pr{p->event(), p->event(), family, bytes, flops};
    if (hipEventRecord(pr.e0, stream) != hipSuccess) { p->pool.push_back(pr.e0); p->pool.push_back(pr.e1); return -1; }
    p->recs.push_back(pr);
    return (int)p->recs.size() - 1;
}
void imk_prof_end(int slot, hipStream_t stream) {
    imk_prof *p = t_prof;
    if (p && slot >= 0 && slot < (int)p->recs.size()) (void)hipEventRecord(p->recs[slot].e1, stream);
}

extern "C" int imk_prof_totals_enable(imk_prof *p, int on) {
    IMK_CHECK_ARG(p);
    p->totals.clear();
    p->totals_on = on != 0;
    if (t_prof == p) imk_tls_totals_on = p->totals_on;      // (the launches of the thread the context is bound to are the ones counted)
    t_work.armed = false;
    return IMK_OK;
}
// "name;launches;bytes;flops\n" per kernel name; returns the bytes needed incl. the terminating 0 (call again if > cap)
extern "C" int64_t imk_prof_totals_dump(imk_prof *p, char *buf, int64_t cap) {
    if (!p) return IMK_EINVAL;
    std::string out;
    char line[160];
    for (auto &kv : p->totals) {
        snprintf(line, sizeof line, ";%lld;%.0f;%.0f\n", kv.second.launches, kv.second.bytes, kv.second.flops);
        out += kv.first; out += line;
    }
    if (buf && cap > 0) {
        const size_t n = std::min((size_t)cap - 1, out.size());
        memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return (int64_t)out.size() + 1;
}
// A marker dispatch on `stream` (kernel imk_mark_kernel, grid size 64 * id): lets a kernel trace be cut at the timed region
extern "C" int imk_prof_mark(int id, void *stream) {
    IMK_CHECK_ARG(id > 0 && id < 65536);
    imk_klaunch(imk_mark_kernel, dim3(id), dim3(64), 0, (hipStream_t)stream, id);
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}

extern "C" int imk_prof_collect(imk_prof *p, int64_t *count, double *ms, double *bytes, double *flops) {
    IMK_CHECK_ARG(p && count && ms && bytes);
    static_assert(IMK_PROF_VARIANTS == PF_COUNT, "include/imk.h and imk_common.h disagree");
    for (int v = 0; v < IMK_PROF_VARIANTS; ++v) { count[v] = 0; ms[v] = 0; bytes[v] = 0; if (flops) flops[v] = 0; }
    for (auto &r : p->recs) {
        float t = 0.f;
        if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess) {
            count[r.variant] += 1; ms[r.variant] += t; bytes[r.variant] += r.bytes;
            if (flops) flops[r.variant] += r.flops;
        }
        p->pool.push_back(r.e0); p->pool.push_back(r.e1);
    }
    p->recs.clear();
    return IMK_OK;
}

// Launch geometry of the per-tile kernel for one conv.
// n / d = (n * div_magic(d)) >> 32 for d > 1 and n * d < 2^32 (a tile's index inside its image: pipe_fits); d = 1 is handled
// by the kernels
static inline unsigned div_magic(int d) { return d > 1 ? (unsigned)((1ull << 32) / (unsigned)d + 1) : 0u; }

struct ConvLaunch { ImkConvGeom gm; int th, mt; int gx, gy; size_t lds; };

static int plan_conv_mfma(const ImkConvArgs &a, ConvLaunch &L) {
    const int TH = conv_tile_h(a.x.cs_in, a.ksize);
    const int halo = a.ksize == 3 ? 1 : 0;
    const int nc8 = a.x.cs_in / 8;
    const int nc8p = imk_pass_chunks(nc8), n_pass = imk_cdiv_d(nc8, nc8p);
    const int ps = odd_ps(nc8p);
    const int T = a.ksize == 3 ? 9 : 1;
    const int nsp = (T * nc8p + 3) / 4;
    const int mt_total = (a.cout + 15) / 16;
    const int tiles_x = imk_cdiv(a.W, TW), tiles_y = imk_cdiv(a.H, TH);
    const size_t tile_bytes = (size_t)(TH + 2 * halo) * (TW + 2 * halo) * ps * 16;
    const size_t stats_bytes = 4 * 2 * 16 * 4 * sizeof(float);   // [4 waves][2][16 * MT <= 64]
    size_t lds = tile_bytes + 4 * (size_t)a.x.cs_in * sizeof(float);
    if (lds < stats_bytes) lds = stats_bytes;
    const size_t lds_base = lds;
    // output-channel tiles per workgroup: as many as possible while the grid still has >= 512 workgroups
    // (deep layers have few tiles; they are latency-bound, so parallelism beats operand reuse there)
    const int n_sp = a.B * tiles_x * tiles_y;
    int mt = mt_total >= 4 ? 4 : (mt_total >= 2 ? 2 : 1);
    while (mt > 1 && n_sp * imk_cdiv(mt_total, mt) < 512) mt >>= 1;
    // the weight fragments of one pass sit in LDS next to the tile (1 KB per (channel tile, k-step); a pass has at most
    // 9 k-steps, so this always fits)
    if (a.wpk2) {   // chained 1x1: one workgroup owns all channels of its tile (imk_conv_can_chain has checked the sizes)
        const int mt2 = (a.cout2 + 15) / 16;
        mt = (mt_total <= 2 && mt2 <= 2) ? 2 : 4;
        if (mt_total > mt || mt2 > mt || TH != 16) return IMK_EUNSUPPORTED;
    }
    lds = lds_base + (size_t)mt * nsp * 1024;
    const size_t out_bytes = (size_t)TH * 16 * (mt * 16 + 8) * sizeof(f16);   // the epilogue's output tile reuses the LDS
    if (a.wpk2) {
        const int nc8_2 = a.cs_out / 8;
        lds = imk_chain_w2_offset(lds_base + (size_t)mt * nsp * 1024, out_bytes) +
              (size_t)((a.cout2 + 15) / 16) * imk_cdiv_d(nc8_2, imk_pass_chunks(nc8_2)) * 1024;
    }
    if (lds < out_bytes) lds = out_bytes;
    if (lds > 160 * 1024) return IMK_EUNSUPPORTED;
    if (!((long long)a.H * a.W < imk_conv_max_pixels() && a.W < (1 << 16))) return IMK_EUNSUPPORTED;   // tile-address arithmetic (PTile / PSrc)
    const long long per_img = (long long)tiles_x * tiles_y;
    L.gm = ImkConvGeom{tiles_x, tiles_y, mt_total, nc8, nc8p, n_pass, ps, nsp, div_magic(tiles_x),
                       (per_img > 1 && (long long)n_sp * per_img < (1ll << 32)) ? div_magic((int)per_img) : 0u};
    L.th = TH; L.mt = mt; L.gx = n_sp; L.gy = imk_cdiv(mt_total, mt); L.lds = lds;
    return IMK_OK;
}

template <typename K>
static int set_lds_limit(K kern, size_t lds) {
    if (lds > 64 * 1024)  // above the default dynamic-LDS limit: opt in (idempotent, no sync)
        IMK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return IMK_OK;
}

static int launch_conv_mfma(const ImkConvArgs &a, hipStream_t stream) {
    ConvLaunch L{};
    int rc = plan_conv_mfma(a, L);
    if (rc) return rc;
    const dim3 grid(L.gx, L.gy);
    auto launch = [&](auto kern) -> int {
        int r = set_lds_limit(kern, L.lds);
        if (r) return r;
        imk_klaunch(kern, dim3(grid), dim3(256), L.lds, stream, a, L.gm);
        return IMK_OK;
    };
    ImkProfScope prof(PF_CONV_MFMA + (L.th == 8 ? 3 : 0) + (L.mt == 4 ? 2 : (L.mt == 2 ? 1 : 0)), imk_conv_algorithmic_bytes(a), stream, imk_conv_flops(a));
    if (a.wpk2) {   // Conv3x3+ReLU -> Conv1x1+ReLU in one launch: encoder blocks pool on load, decoder blocks read a BatchNorm output
        if (L.gy != 1 || L.th != 16) return IMK_EUNSUPPORTED;
        if (a.x.lmode == LM_POOL) rc = L.mt == 4 ? launch(conv_mfma_kernel<16, 4, LM_POOL, true>) : launch(conv_mfma_kernel<16, 2, LM_POOL, true>);
        else if (a.x.lmode == LM_AFFINE) rc = L.mt == 4 ? launch(conv_mfma_kernel<16, 4, LM_AFFINE, true>) : launch(conv_mfma_kernel<16, 2, LM_AFFINE, true>);
        else return IMK_EUNSUPPORTED;
        if (rc) return rc;
        if (a.stats_rows) *a.stats_rows = L.gx;
        IMK_LAUNCH_CHECK();
        return IMK_OK;
    }
#define IMK_MFMA_MT(TH, LM) \
    (L.mt == 4 ? launch(conv_mfma_kernel<TH, 4, LM>) : (L.mt == 2 ? launch(conv_mfma_kernel<TH, 2, LM>) : launch(conv_mfma_kernel<TH, 1, LM>)))
#define IMK_MFMA_TH(LM) (L.th == 16 ? IMK_MFMA_MT(16, LM) : IMK_MFMA_MT(8, LM))
    switch (a.x.lmode) {
        case LM_RAW: rc = IMK_MFMA_TH(LM_RAW); break;
        case LM_AFFINE: rc = IMK_MFMA_TH(LM_AFFINE); break;
        case LM_POOL: rc = IMK_MFMA_TH(LM_POOL); break;
        case LM_UPADD: rc = IMK_MFMA_TH(LM_UPADD); break;
        case LM_BNBWD: rc = IMK_MFMA_TH(LM_BNBWD); break;
        default: rc = IMK_MFMA_TH(LM_U8); break;
    }
#undef IMK_MFMA_TH
#undef IMK_MFMA_MT
    if (rc) return rc;
    if (a.stats_rows) *a.stats_rows = L.gx;
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}

// Limits of conv_pipe_kernel's address arithmetic (24-bit multiplies of tile-relative pixel offsets, 32-bit magic division of
// a tile's index inside its image): images below 2^24 pixels, rows below 2^16.  Plans are refused above that
// (imk_conv_max_pixels), so the layouts decided at plan time (pair layout, chains) never meet an image that does not fit.
static bool pipe_fits(const ImkConvArgs &a) { return (long long)a.H * a.W < imk_conv_max_pixels() && a.W < (1 << 16); }


static bool dyn_walk_on() {
    static const bool on = []() { const char *e = getenv("IMK_DYN_WALK"); return !(e && e[0] == '0'); }();
    return on;
}

template <int LM, int NC8, int CHAIN, bool PAIR, int EPI, bool DYSTAT, bool FULL, int WG = 0, int PRE = 0, bool DYN = false>
static int launch_conv_pipe_k(const ImkConvArgs &a, hipStream_t stream) {
    if constexpr (!DYN && EPI == EP_RELU && !DYSTAT && WG == 0) {      // inference launches with tile counters: the dynamic walk
        if (a.sched && !a.stats_partial && dyn_walk_on() && a.B * imk_cdiv(a.H, 16) * imk_cdiv(a.W, TW) >= 2048)
            return launch_conv_pipe_k<LM, NC8, CHAIN, PAIR, EPI, DYSTAT, FULL, WG, PRE, true>(a, stream);
    }
    if (a.x.cs_in != (PRE ? PRE : NC8) * 8) return IMK_EINVAL;      // the kernel takes the input's channel stride from its template arguments
    static int blocks_per_cu = 0;   // occupancy of this instantiation, queried once
    const size_t lds = pipe_lds_base(NC8, PAIR, WG) + (PRE ? (size_t)18 * 18 * imk_lds_pitch(PRE) * 16 + 3 * 16 * sizeof(float) : 0);
    auto kern = conv_pipe_kernel<LM, NC8, CHAIN, PAIR, EPI, DYSTAT, FULL, WG, PRE, DYN>;
    if (blocks_per_cu == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, lds) != hipSuccess || nb < 1) nb = 4;
        static const int cap = []() { const char *e = getenv("IMK_PIPE_BLOCKS_PER_CU"); return e ? atoi(e) : 8; }();
        blocks_per_cu = nb > cap ? cap : nb;
    }
    const int tiles_x = imk_cdiv(a.W, TW), tiles_y = imk_cdiv(a.H, 16);
    const int n_tiles = a.B * tiles_x * tiles_y;
    int grid = 256 * blocks_per_cu;
    if (grid > n_tiles) grid = n_tiles;
    const ImkWalk wk = imk_walk_make(grid, n_tiles, tiles_x * tiles_y, DYN);
    if (DYN && wk.shift != 5) return IMK_EINVAL;      // (>= 2048 tiles: always the 32-group walk)
    ImkProfScope prof(PF_CONV_PIPE, imk_conv_algorithmic_bytes(a), stream, imk_conv_flops(a));
    imk_klaunch(kern, dim3(grid), dim3(256), lds, stream, a, tiles_x, tiles_y, n_tiles, div_magic(tiles_x), wk);
    IMK_LAUNCH_CHECK();
    if (a.stats_rows) *a.stats_rows = grid;
    return IMK_OK;
}

// Forward convs (EP_RELU): every load mode but BNBWD, optionally chained.  Gradient convs (EP_PLAIN / EP_MASK, with or
// without the BN-gradient statistics): input is a raw gradient or a BN backward on load; never chained.
template <int NC8, bool PAIR, bool FULL>
static int launch_conv_pipe_v(const ImkConvArgs &a, hipStream_t stream) {
    if (a.pre_wpk) {        // 1x1 first stage + 3x3 + chained 1x1 (inference); instantiated for the pair layout (imk_conv_can_prestage)
        if constexpr (PAIR && NC8 == 1) {
#ifndef IMK_PRE_DEBUG     // (the probe build passes dump tensors in a.out / a.mask)
            if (a.out) return IMK_EUNSUPPORTED;
#endif
            if (a.epi != EP_RELU || !a.wpk2 || a.x.lmode != LM_UPADD || a.x.cs_in != 8) return IMK_EUNSUPPORTED;
            return launch_conv_pipe_k<LM_UPADD, 1, 2, true, EP_RELU, false, FULL, 0, 1>(a, stream);
        } else {
            return IMK_EUNSUPPORTED;
        }
    }
    if (a.epi == EP_RELU) {
        const int chain = a.wpk2 ? (a.out ? 1 : 2) : 0;
#define IMK_PIPE_FWD(LM)                                                                              \
        (chain == 0 ? launch_conv_pipe_k<LM, NC8, 0, PAIR, EP_RELU, false, FULL>(a, stream)           \
         : chain == 1 ? launch_conv_pipe_k<LM, NC8, 1, PAIR, EP_RELU, false, FULL>(a, stream)         \
                      : launch_conv_pipe_k<LM, NC8, 2, PAIR, EP_RELU, false, FULL>(a, stream))
        switch (a.x.lmode) {
            case LM_RAW: return IMK_PIPE_FWD(LM_RAW);
            case LM_AFFINE: return IMK_PIPE_FWD(LM_AFFINE);
            case LM_POOL: return IMK_PIPE_FWD(LM_POOL);
            case LM_UPADD: return IMK_PIPE_FWD(LM_UPADD);
            case LM_U8: return IMK_PIPE_FWD(LM_U8);
            case LM_STEM:   // inference chains only (Conv3x3 -> Conv1x1 without the intermediate)
                return chain == 2 ? launch_conv_pipe_k<LM_STEM, NC8, 2, PAIR, EP_RELU, false, FULL>(a, stream) : IMK_EUNSUPPORTED;
            default: return IMK_EUNSUPPORTED;
        }
#undef IMK_PIPE_FWD
    }
    if (a.wpk2) return IMK_EUNSUPPORTED;
    if (a.sum2_out) {        // 1x1 dgrad + the 2x2 sums of its output with their BatchNorm-backward statistics (imk_conv_can_sum2)
        if constexpr (FULL) {
            if (a.wg_partial || a.ksize != 1 || a.x.lmode != LM_BNBWD || a.epi != EP_PLAIN || !a.sum2_z || !a.stats_partial || a.dystat_z ||
                (a.H & 1) || (a.W & 1))
                return IMK_EUNSUPPORTED;
            return launch_conv_pipe_k<LM_BNBWD, NC8, 0, PAIR, EP_PLAIN, false, true, 4>(a, stream);
        } else {
            return IMK_EUNSUPPORTED;
        }
    }
    if (a.wg_partial) {      // dgrad + weight gradient of a 1x1 conv in one launch (callers check imk_conv_can_fuse_wgrad)
        if constexpr (FULL) {
            const bool dys = a.dystat_z && a.stats_partial;
            if (a.ksize == 3) {
                if (a.x.lmode == LM_RAW && a.epi == EP_PLAIN && dys && a.wg_sc && a.wg_sh)
                    return launch_conv_pipe_k<LM_RAW, NC8, 0, PAIR, EP_PLAIN, true, true, 3>(a, stream);
                return IMK_EUNSUPPORTED;
            }
            if (a.x.lmode == LM_BNBWD && a.epi == EP_MASK && !dys)
                return launch_conv_pipe_k<LM_BNBWD, NC8, 0, PAIR, EP_MASK, false, true, 1>(a, stream);
            if (a.x.lmode == LM_RAW && a.epi == EP_PLAIN && dys && a.wg_sc && a.wg_sh)
                return launch_conv_pipe_k<LM_RAW, NC8, 0, PAIR, EP_PLAIN, true, true, 2>(a, stream);
            return IMK_EUNSUPPORTED;
        } else {
            return IMK_EUNSUPPORTED;
        }
    }
    const bool dystat = a.dystat_z && a.stats_partial;
#define IMK_PIPE_BWD(LM)                                                                                         \
    (a.epi == EP_MASK ? (dystat ? launch_conv_pipe_k<LM, NC8, 0, PAIR, EP_MASK, true, FULL>(a, stream)           \
                                : launch_conv_pipe_k<LM, NC8, 0, PAIR, EP_MASK, false, FULL>(a, stream))         \
                      : (dystat ? launch_conv_pipe_k<LM, NC8, 0, PAIR, EP_PLAIN, true, FULL>(a, stream)          \
                                : launch_conv_pipe_k<LM, NC8, 0, PAIR, EP_PLAIN, false, FULL>(a, stream)))
    switch (a.x.lmode) {
        case LM_RAW: return IMK_PIPE_BWD(LM_RAW);
        case LM_BNBWD: return IMK_PIPE_BWD(LM_BNBWD);
        default: return IMK_EUNSUPPORTED;
    }
#undef IMK_PIPE_BWD
}

static int launch_conv_pipe_any(const ImkConvArgs &a, hipStream_t stream) {
    const bool all_ch = a.pair || (a.cs_out == 16 && (!a.wpk2 || a.cs_out2 == 16));
    const bool u8_ok = a.x.lmode != LM_U8 || a.pre_wpk || (a.ksize == 1 && ((uintptr_t)a.x.in & 15) == 0);   // FULL + u8: row-segment loads
    const bool full = (a.H % 16 == 0) && (a.W % TW == 0) && all_ch && u8_ok;
    const bool nc1 = (a.pre_wpk ? imk_pad8(a.pre_cout) : a.x.cs_in) == 8;      // chunks per pixel of THIS conv's input
#define IMK_PIPE_SEL(NC8)                                                                                   \
    (a.pair ? (full ? launch_conv_pipe_v<NC8, true, true>(a, stream) : launch_conv_pipe_v<NC8, true, false>(a, stream)) \
            : (full ? launch_conv_pipe_v<NC8, false, true>(a, stream) : launch_conv_pipe_v<NC8, false, false>(a, stream)))
    return nc1 ? IMK_PIPE_SEL(1) : IMK_PIPE_SEL(2);
#undef IMK_PIPE_SEL
}

template <int LM, int NC8, int MT, int EPI, bool DYSTAT, bool FULL, bool CHAIN2 = false>
static int launch_conv_wide_k(const ImkConvArgs &a, hipStream_t stream) {
    static int blocks_per_cu = 0;
    const int ns = ((a.ksize == 3 ? 9 : 1) * NC8 + 3) / 4;
    const size_t lds = (size_t)18 * 18 * imk_lds_pitch(NC8) * 16 + (4 * 32 + 4 * 2 * 16 * MT) * sizeof(float) + (size_t)MT * ns * 1024 +
                       (CHAIN2 ? (size_t)4 * 64 * 48 * sizeof(f16) : 0);
    auto kern = conv_wide_kernel<LM, NC8, MT, EPI, DYSTAT, FULL, CHAIN2>;
    { int r = set_lds_limit(kern, lds); if (r) return r; }
    if (blocks_per_cu == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, lds) != hipSuccess || nb < 1) nb = 2;
        blocks_per_cu = nb > 8 ? 8 : nb;
    }
    const int tiles_x = imk_cdiv(a.W, TW), tiles_y = imk_cdiv(a.H, 16);
    const int n_tiles = a.B * tiles_x * tiles_y;
    int grid = 256 * blocks_per_cu;
    if (grid > n_tiles) grid = n_tiles;
    const ImkWalk wk = imk_walk_make(grid, n_tiles, tiles_x * tiles_y);
    ImkProfScope prof(PF_CONV_PIPE, imk_conv_algorithmic_bytes(a), stream, imk_conv_flops(a));
    imk_klaunch(kern, dim3(grid), dim3(256), lds, stream, a, tiles_x, tiles_y, n_tiles, div_magic(tiles_x), wk);
    IMK_LAUNCH_CHECK();
    if (a.stats_rows) *a.stats_rows = grid;
    return IMK_OK;
}

template <int NC8, int MT, bool FULL>
static int launch_conv_wide_v(const ImkConvArgs &a, hipStream_t stream) {
    if (a.epi == EP_RELU) {
        switch (a.x.lmode) {
            case LM_RAW: return launch_conv_wide_k<LM_RAW, NC8, MT, EP_RELU, false, FULL>(a, stream);
            case LM_AFFINE: return launch_conv_wide_k<LM_AFFINE, NC8, MT, EP_RELU, false, FULL>(a, stream);
            case LM_UPADD: return launch_conv_wide_k<LM_UPADD, NC8, MT, EP_RELU, false, FULL>(a, stream);
            case LM_U8: if constexpr (NC8 == 1) return launch_conv_wide_k<LM_U8, 1, MT, EP_RELU, false, FULL>(a, stream); else return IMK_EUNSUPPORTED;
            default: return IMK_EUNSUPPORTED;
        }
    }
    const bool dystat = a.dystat_z && a.stats_partial;
#define IMK_WIDE_BWD(LM)                                                                                          \
    (a.epi == EP_MASK ? (dystat ? launch_conv_wide_k<LM, NC8, MT, EP_MASK, true, FULL>(a, stream)                 \
                                : launch_conv_wide_k<LM, NC8, MT, EP_MASK, false, FULL>(a, stream))               \
                      : (dystat ? launch_conv_wide_k<LM, NC8, MT, EP_PLAIN, true, FULL>(a, stream)                \
                                : launch_conv_wide_k<LM, NC8, MT, EP_PLAIN, false, FULL>(a, stream)))
    switch (a.x.lmode) {
        case LM_RAW: return IMK_WIDE_BWD(LM_RAW);
        case LM_BNBWD: return IMK_WIDE_BWD(LM_BNBWD);
        default: return IMK_EUNSUPPORTED;
    }
#undef IMK_WIDE_BWD
}

// 17-32 channels on at least one side, at most 32 on both: the wide persistent kernel (see conv_wide_kernel)
static bool conv_wide_ok(const ImkConvArgs &a) {
    static const bool off = []() { const char *e = getenv("IMK_CONV_WIDE"); return e && e[0] == '0'; }();
    if (off || a.wpk2 || a.x.cs_in > 32 || a.cout > 32 || !pipe_fits(a)) return false;
    if (a.x.lmode == LM_POOL || a.x.lmode == LM_STEM) return false;
    if (a.x.lmode == LM_U8 && a.x.cin > 4) return false;
    return true;
}

// Conv3x3+ReLU -> Conv1x1+ReLU in one launch of the wide kernel (inference: the intermediate is not stored, no statistics)
static bool conv_wide_chain_ok(const ImkConvArgs &a) {
    static const bool off = []() { const char *e = getenv("IMK_WIDE_CHAIN"); return e && e[0] == '0'; }();
    static const bool train_off = []() { const char *e = getenv("IMK_WIDE_CHAIN_TRAIN"); return e && e[0] == '0'; }();
    if (off || !a.wpk2 || a.epi != EP_RELU || a.ksize != 3) return false;
    if ((a.out || a.stats_partial) && (train_off || a.cout2 > a.cout)) return false;   // training: intermediate stored, statistics of the 1x1's output
    if (a.x.cs_in < 16 || a.x.cs_in > 32 || a.cout > 32 || a.cout2 > 32 || (a.x.cs_in <= 16 && a.cout <= 16)) return false;
    if (a.x.lmode != LM_AFFINE) return false;     // (the pooled-input form, alpha 1's second encoder block, gains nothing: 1.170 vs 1.166 ms)
    ImkConvArgs plain = a;
    plain.wpk2 = nullptr;
    return conv_wide_ok(plain);
}
static int launch_conv_wide_chain(const ImkConvArgs &a, hipStream_t stream) {
    const bool full = (a.H % 16 == 0) && (a.W % TW == 0);
    const int nc8 = a.x.cs_in / 8;
    const bool mt2 = a.cout > 16;
#define IMK_WC(NC8V, MTV) (full ? launch_conv_wide_k<LM_AFFINE, NC8V, MTV, EP_RELU, false, true, true>(a, stream)  \
                                : launch_conv_wide_k<LM_AFFINE, NC8V, MTV, EP_RELU, false, false, true>(a, stream))
    switch (nc8) {
        case 2: return mt2 ? IMK_WC(2, 2) : IMK_EUNSUPPORTED;
        case 3: return mt2 ? IMK_WC(3, 2) : IMK_WC(3, 1);
        case 4: return mt2 ? IMK_WC(4, 2) : IMK_WC(4, 1);
        default: return IMK_EUNSUPPORTED;
    }
#undef IMK_WC
}

static int launch_conv_wide_any(const ImkConvArgs &a, hipStream_t stream) {
    const bool full = (a.H % 16 == 0) && (a.W % TW == 0);
    const int nc8 = a.x.cs_in / 8;
    const int mt = a.cout > 16 ? 2 : 1;
#define IMK_WIDE_SEL(NC8V, MTV) (full ? launch_conv_wide_v<NC8V, MTV, true>(a, stream) : launch_conv_wide_v<NC8V, MTV, false>(a, stream))
#define IMK_WIDE_MT(NC8V) (mt == 2 ? IMK_WIDE_SEL(NC8V, 2) : IMK_WIDE_SEL(NC8V, 1))
    switch (nc8) {
        case 1: return IMK_WIDE_MT(1);
        case 2: return IMK_WIDE_MT(2);
        case 3: return IMK_WIDE_MT(3);
        case 4: return IMK_WIDE_MT(4);
        default: return IMK_EUNSUPPORTED;
    }
#undef IMK_WIDE_MT
#undef IMK_WIDE_SEL
}

static bool pipe_enabled() {
    static const bool on = []() { const char *e = getenv("IMK_CONV_PIPE"); return !(e && e[0] == '0'); }();
    return on;
}
static bool pair_enabled() {
    static const bool on = []() { const char *e = getenv("IMK_CONV_PAIR"); return !(e && e[0] == '0'); }();
    return on && pipe_enabled();
}

bool imk_conv_pair_layout(int k_in, int m_out, bool u8_input) {
    return pair_enabled() && imk_pad8(k_in) <= 16 && m_out <= 8 && (!u8_input || k_in <= 4);
}

bool imk_conv_stem_fusable(int u8_c, int ch0, int cout_next) {
    static const bool off = []() { const char *e = getenv("IMK_STEM_FUSE"); return e && e[0] == '0'; }();
    return !off && pipe_enabled() && u8_c <= 4 && ch0 <= 16 && cout_next <= 16;
}

bool imk_conv_can_chain(const ImkConvArgs &a, int cout2) {
    static const bool off = []() { const char *e = getenv("IMK_CONV_CHAIN"); return e && e[0] == '0'; }();
    if (off || !pipe_enabled() || !pipe_fits(a)) return false;
    if (pair_enabled() && a.cout <= 8 && cout2 > 8) return false;   // the two stages must use the same fragment layout
    return a.epi == EP_RELU && a.x.cs_in <= 16 && a.cout <= 16 && cout2 <= 16 && (a.x.lmode != LM_U8 || a.x.cin <= 4);
}

// conv_pipe_kernel<..., PRE>: `main` is a chained 3x3 launch of the shallow kernel without a stored intermediate (inference);
// the 1x1 in front of it reads a uint8 image or upsample + skip with at most 16 channels and uses the same weight layout.
bool imk_conv_can_prestage(const ImkConvArgs &a, int lm_pre, int cin_pre, int cout_pre) {
    static const bool off = []() { const char *e = getenv("IMK_CONV_PRESTAGE"); return e && e[0] == '0'; }();
    if (off || !pipe_enabled() || !pipe_fits(a)) return false;
    if (a.epi != EP_RELU || a.ksize != 3 || !a.wpk2 || a.out) return false;
    // Measured (inference call of 128 images, ms without / with): ISIC (8 channels, pair layout) 0.538 / 0.503 -- the decoder's
    // full- and half-resolution blocks 117.9 -> 104.5 us and 61.0 -> 53.8 us; SUIM (16 channels, plain layout: 170 VGPRs, two
    // waves per SIMD) 1.260 / 1.269; the input block this way 0.507 against 0.503 for the LM_STEM form.  So: up + add in front
    // of pair-layout convs only (both stages <= 8 channels).
    if (lm_pre != LM_UPADD || imk_pad8(cin_pre) != 8 || cout_pre > 8 || a.cout > 8 || a.cout2 > 8) return false;
    return imk_conv_pair_layout(cin_pre, cout_pre, false) && imk_conv_pair_layout(cout_pre, a.cout, false);
}

// The per-tile kernel's chain (conv_mfma_kernel<..., CHAIN>): a 3x3 conv with 17-64 output channels that pools or reads a
// BatchNorm output, followed by a 1x1 with at most as many 16-channel tiles -- the mid / deep blocks.  The second conv uses
// its regular forward pack.  Bit-identical to the two per-tile launches.  Measured (ms per training step B = 32 | inference
// call B = 128; off / on): ISIC 1.104 / 1.095 | 0.637 / 0.592, SUIM 2.052 / 2.028 | 1.463 / 1.360, HeLa 2.022 / 2.002 | 1.386 /
// 1.288, Cityscapes 3.11 / 3.10 | 2.652 / 2.534: inference drops the intermediate tensor's write and read, training keeps
// the write and trades a launch for fewer, longer workgroups.  Layers the 17-32 channel kernel would take (conv_wide_ok)
// chain when the intermediate is not stored or the launch is small (EvalNet's full-resolution towers in training: 2.87 ms
// per step as two conv_wide launches, 2.94 chained).  IMK_CONV_CHAIN_TILE = 0 off, 2 always.
bool imk_conv_can_chain_tile(const ImkConvArgs &a, int cout2, bool store_mid) {
    static const int mode = []() { const char *e = getenv("IMK_CONV_CHAIN_TILE"); return e ? atoi(e) : 1; }();
    if (mode == 0) return false;
    // Where the GEMM-class kernel would take the 3x3 (imk_gemm.hip): IMK_GEMM_OVER_CHAIN = 0 chain anyway, 1 two launches when
    // the intermediate is stored anyway (training), 2 always two launches
    static const int over = []() { const char *e = getenv("IMK_GEMM_OVER_CHAIN"); return e ? atoi(e) : 1; }();
    if (a.epi == EP_RELU && a.ksize == 3 && (a.x.lmode == LM_POOL || a.x.lmode == LM_AFFINE) && a.cout <= 128 &&
        cout2 <= 128 && imk_pad8(cout2) <= (a.cout > 64 ? 128 : 64)) {     // the GEMM-class chain (imk_conv_gemm_chain_ok, before wpk2 is set)
        static const bool gc_off = []() { const char *e = getenv("IMK_GEMM_CHAIN"); return e && e[0] == '0'; }();
        // training (store_mid): IMK_GEMM_CHAIN_TRAIN = 0 off, 2 only where the two convs have the same width (encoder blocks: the
        // chain's statistics rows are then bit for bit those of the 1x1's own launch), 1 (default) everywhere
        static const int gct = []() { const char *e = getenv("IMK_GEMM_CHAIN_TRAIN"); return e ? atoi(e) : 1; }();
        ImkConvArgs plain = a;
        plain.wpk2 = nullptr;
        const bool train_ok = gct == 1 || (gct == 2 && imk_pad8(cout2) == a.cs_out);
        if (!gc_off && (!store_mid || train_ok) && imk_conv_gemm_ok(plain)) return true;
    }
    if (over == 2 || (over == 1 && store_mid)) {
        ImkConvArgs plain = a;
        plain.wpk2 = nullptr;
        if (imk_conv_gemm_ok(plain)) return false;
    }
    const bool pipe_ok = pipe_enabled() && pipe_fits(a) && a.x.cs_in <= 16 && a.cout <= 16;
    if (pipe_ok || a.epi != EP_RELU || a.ksize != 3 || a.cout > 64 || cout2 > 64) return false;
    if (a.x.lmode != LM_POOL && a.x.lmode != LM_AFFINE) return false;
    const bool wide = a.x.cs_in <= 32 && a.cout <= 32 && a.x.lmode != LM_POOL;     // conv_wide_kernel's layers
    // (large launches of these layers in training: two conv_wide launches beat the per-tile chain; the persistent kernel's own
    //  chain -- conv_wide_kernel<..., CHAIN2>, round 3 -- beats both where it applies)
    static const bool wct_off = []() { const char *e = getenv("IMK_WIDE_CHAIN_TRAIN"); return e && e[0] == '0'; }();
    const bool wide_chain = !wct_off && wide && a.x.lmode == LM_AFFINE && a.x.cs_in >= 16 && cout2 <= a.cout && pipe_fits(a);
    if (mode == 1 && wide && store_mid && !wide_chain && (long long)a.B * imk_cdiv(a.H, 16) * imk_cdiv(a.W, TW) > 2048) return false;
    const int mt1 = (a.cout + 15) / 16, mt2 = (cout2 + 15) / 16, mt = (mt1 <= 2 && mt2 <= 2) ? 2 : 4;
    return mt1 <= mt && mt2 <= mt && conv_tile_h(a.x.cs_in, 3) == 16;
}

// Can this dgrad launch also produce the weight gradient of its conv (ImkConvArgs::wg_partial)?  Mirrors the choices of
// imk_launch_conv / launch_conv_pipe_any: pipelined kernel, full tiles, every lane owning real channels.
bool imk_conv_can_fuse_wgrad(const ImkConvArgs &a) {
    static const bool off = []() { const char *e = getenv("IMK_FUSE_WGRAD"); return e && e[0] == '0'; }();
    if (off || !pipe_enabled() || !pipe_fits(a)) return false;
    if (a.wpk2) return false;
    const bool form1 = a.ksize == 1 && a.x.lmode == LM_BNBWD && a.epi == EP_MASK && a.mask && !(a.dystat_z && a.stats_partial);
    const bool form2 = a.x.lmode == LM_RAW && a.epi == EP_PLAIN && a.dystat_z && a.stats_partial;     // 1x1 and 3x3
    // The 3x3 form doubles the launch's matrix and LDS work (20 MFMAs + 40 transposed reads per wave and tile on top of the dgrad's 20):
    // at ISIC the half-resolution 16-channel launch takes 42 us fused against ~20 us for the plain dgrad.  Round 5 measured the
    // alternatives on one box against round 4's library (IMK_FUSE_WGRAD_C3: 0 = never fuse the 3x3 form, 8 = pair layout only, unset =
    // wherever it applies): ISIC / SUIM / HeLa steps within +-0.5 % of each other under every rule -- what the chain gains, the fork and
    // the weight-gradient launch beside it give back (profiles/r05_notes.md).  The default stays "wherever it applies".
    static const int c3_mode = []() { const char *e = getenv("IMK_FUSE_WGRAD_C3"); return e ? atoi(e) : -1; }();
    if (!form1 && !form2) return false;
    if (a.ksize == 3 && (c3_mode == 0 || (c3_mode == 8 && a.cout > 8))) return false;
    if (a.x.cs_in > 16 || a.cout > 16) return false;
    const bool pair = pair_enabled() && a.cout <= 8;
    const bool all_ch = pair || a.cs_out == 16;
    return (a.H % 16 == 0) && (a.W % TW == 0) && all_ch && (pair ? a.cs_out == 8 : true);
}
int imk_conv_fused_wgrad_rows_max() { return 256 * 8; }

// Can this 1x1 dgrad launch also emit the 2x2 sums of its output (ImkConvArgs::sum2_out)?  The pipelined kernel, full tiles, every
// lane owning real channels -- what launch_conv_pipe_any would pick for it.
bool imk_conv_can_sum2(const ImkConvArgs &a) {
    static const bool off = []() { const char *e = getenv("IMK_FUSE_SUM2"); return e && e[0] == '0'; }();
    if (off || !pipe_enabled() || !pipe_fits(a) || a.wpk2 || a.wg_partial) return false;
    if (a.ksize != 1 || a.x.lmode != LM_BNBWD || a.epi != EP_PLAIN || (a.dystat_z && a.stats_partial)) return false;
    if (a.x.cs_in > 16 || a.cout > 16) return false;
    const bool pair = pair_enabled() && a.cout <= 8;
    const bool all_ch = pair ? a.cs_out == 8 : a.cs_out == 16;
    return (a.H % 16 == 0) && (a.W % TW == 0) && all_ch;
}

static bool g_use_pipe = true;   // IMK_CONV_PIPE=0 in the environment falls back to the per-tile kernel (A/B runs)

int imk_launch_conv(const ImkConvArgs &a_in, hipStream_t stream) {
    ImkConvArgs a = a_in;
    IMK_CHECK_ARG(a.x.in && a.wpk && (a.out || a.wpk2) && a.B > 0 && a.H > 0 && a.W > 0);
    IMK_CHECK_ARG(a.ksize == 1 || a.ksize == 3);
    IMK_CHECK_ARG(a.x.cs_in % 8 == 0 && a.cs_out % 8 == 0 && a.x.cs_in >= a.x.cin && a.cs_out >= a.cout);
    IMK_CHECK_ARG(a.x.lmode != LM_U8 || (a.x.cs_in == 8 && a.x.cin <= 8));
    if (a.x.cs_in > 512) return IMK_EUNSUPPORTED;
    static const bool env_checked = []() { const char *e = getenv("IMK_CONV_PIPE"); if (e && e[0] == '0') g_use_pipe = false; return true; }();
    (void)env_checked;
    const bool pipe_ok = g_use_pipe && pipe_fits(a) && a.x.cs_in <= 16 && a.cout <= 16 && (a.x.lmode != LM_U8 || a.x.cin <= 4) &&
                         (!a.pre_wpk || imk_pad8(a.pre_cout) <= 16);
    a.pair = pipe_ok && pair_enabled() && a.cout <= 8;   // must mirror imk_conv_pair_layout
    if (a.pre_wpk && !(pipe_ok && a.wpk2 && a.pre_bias && a.pre_sc && a.pre_sh && a.pre_cout > 0)) return IMK_EUNSUPPORTED;
    if (a.wpk2 && a.pair && a.cout2 > 8) return IMK_EUNSUPPORTED;
    if (a.wpk2) {   // fused second stage (callers check imk_conv_can_chain / imk_conv_can_chain_tile)
        if (a.epi != EP_RELU || !a.out2 || !a.bias2 || a.cs_out2 % 8) return IMK_EUNSUPPORTED;
        if (pipe_ok) return a.cout2 > 16 ? IMK_EUNSUPPORTED : launch_conv_pipe_any(a, stream);
        if (!imk_conv_can_chain_tile(a, a.cout2, a.out != nullptr)) return IMK_EUNSUPPORTED;
        if (conv_wide_chain_ok(a)) return launch_conv_wide_chain(a, stream);
        if (imk_conv_gemm_chain_ok(a)) return imk_launch_conv_gemm(a, stream);
        return launch_conv_mfma(a, stream);
    }
    if (pipe_ok) return launch_conv_pipe_any(a, stream);
    if (a.x.lmode == LM_STEM) return IMK_EUNSUPPORTED;
    if (conv_wide_ok(a)) return launch_conv_wide_any(a, stream);
    if (imk_conv_gemm_ok(a)) return imk_launch_conv_gemm(a, stream);
    return launch_conv_mfma(a, stream);
}

int imk_wgrad_splits(int B, int H, int W, int cin, int cout) {
    const int n_tiles = B * imk_cdiv(H, 16) * imk_cdiv(W, TW);
    const int n_pairs = ((imk_pad8(cin) + 15) / 16) * ((imk_pad8(cout) + 15) / 16);
    static const int target = []() { const char *e = getenv("IMK_WGRAD_WGS"); return e ? atoi(e) : 768; }();
    int s = target / n_pairs;   // ~3 resident workgroups per CU; each walks its tiles with prefetch
    if (s < 1) s = 1;
    if (s > n_tiles) s = n_tiles;
    return s;
}

size_t imk_wgrad_partial_floats(int B, int H, int W, int ksize, int cin, int cout) {
    const int n_pairs = ((imk_pad8(cin) + 15) / 16) * ((imk_pad8(cout) + 15) / 16);
    const int T = ksize == 3 ? 9 : 1;
    size_t ns = (size_t)imk_wgrad_splits(B, H, W, cin, cout);
    // a 1x1 conv between narrow layers may get its weight gradient from its dgrad launch: one row per workgroup of that
    if (imk_pad8(cin) <= 16 && imk_pad8(cout) <= 16 && ns < (size_t)imk_conv_fused_wgrad_rows_max())
        ns = (size_t)imk_conv_fused_wgrad_rows_max();
    // wide layers: the GEMM-class kernel's split count (whichever kernel runs: the size must not depend on a switch)
    if (imk_wgrad_gemm_wide(imk_pad8(cin), imk_pad8(cout), (long long)B * H * W)) {
        for (int lm : {(int)LM_RAW, (int)LM_POOL, (int)LM_AFFINE}) {     // the split count depends on the load mode (pooling: smaller groups; BatchNorm on load: the 3-per-CU form)
            const size_t ng = (size_t)imk_wgrad_gemm_splits(lm, B, H, W, ksize, imk_pad8(cin), imk_pad8(cout));
            if (ns < ng) ns = ng;
        }
    }
    // a Conv1x1 with 24-64 channels on both sides may get it from the fused backward kernel (imk_bwd1.hip): one row per workgroup
    if (ksize == 1 && imk_pad8(cin) >= 24 && imk_pad8(cin) <= 64 && imk_pad8(cout) >= 24 && imk_pad8(cout) <= 64 &&
        ns < (size_t)imk_bwd1x1_rows((long long)B * H * W, imk_pad8(cin), imk_pad8(cout)))
        ns = (size_t)imk_bwd1x1_rows((long long)B * H * W, imk_pad8(cin), imk_pad8(cout));
    // a softmax output layer may get its weight gradient from the fused head kernel (imk_headf.hip): one row per workgroup
    if (ksize == 1 && n_pairs <= 8 && ns < (size_t)imk_loss_blocks((long long)B * H * W)) ns = (size_t)imk_loss_blocks((long long)B * H * W);
    return (ns + (ns + WG_RED_CHUNK - 1) / WG_RED_CHUNK) * n_pairs * (T + 1) * 256;  // partials + stage-1 scratch
}

struct WgradLaunch { ImkWgradGeom gm; int gx, gy; size_t lds; };

static int plan_wgrad(const ImkWgradArgs &a, WgradLaunch &L) {
    IMK_CHECK_ARG(a.x.in && a.dA && a.partial && a.B > 0 && a.H > 0 && a.W > 0 && a.n_split > 0);
    IMK_CHECK_ARG(a.ksize == 1 || a.ksize == 3);
    IMK_CHECK_ARG(a.x.cs_in % 8 == 0 && a.cs_out % 8 == 0);
    if (a.x.lmode == LM_BNBWD) return IMK_EUNSUPPORTED;   // the x side of a wgrad is always a forward tensor
    if (a.x.lmode == LM_U8 && a.x.cin > 4) return IMK_EUNSUPPORTED;
    const int cit_n = (a.x.cs_in + 15) / 16, cot_n = (a.cs_out + 15) / 16;
    const int tiles_x = imk_cdiv(a.W, TW), tiles_y = imk_cdiv(a.H, 16);
    size_t lds = ((size_t)18 * 18 + 256) * WG_STRIDE_H * sizeof(f16) + (4 * (size_t)a.x.cs_in + 3 * (size_t)a.cs_out) * sizeof(float);
    const size_t red = 4 * (size_t)(a.ksize == 3 ? 10 : 2) * 256 * sizeof(float);     // the four waves' accumulators at the end
    if (lds < red) lds = red;
    L.gm = ImkWgradGeom{tiles_x, tiles_y, a.B * tiles_x * tiles_y, cit_n, cot_n, a.x.cs_in / 8, a.cs_out / 8, ImkWalk{}};
    L.gx = a.n_split; L.gy = cit_n * cot_n; L.lds = lds;
    {   // the split count is the caller's (one partial row per split): the XCD-aware walk only where it is a multiple of 8
        int g = a.n_split;
        L.gm.wk = imk_walk_make(g, L.gm.n_tiles, tiles_x * tiles_y);
        if (g != a.n_split) { g = a.n_split; L.gm.wk = ImkWalk{L.gm.n_tiles, 0, g, 0, 0}; }
    }
    return IMK_OK;
}

// forward input read once (per its load mode) + gradient operand(s) read once + the fp32 partials written
static double wgrad_algorithmic_bytes(const ImkWgradArgs &a, const WgradLaunch &L) {
    const double px = (double)a.B * a.H * a.W;
    double in_b;
    switch (a.x.lmode) {
        case LM_POOL: in_b = 4.0 * px * a.x.cs_in * 2; break;
        case LM_UPADD: in_b = 1.25 * px * a.x.cs_in * 2; break;
        case LM_U8: in_b = px * a.x.cin; break;
        default: in_b = px * a.x.cs_in * 2;
    }
    const int T = a.ksize == 3 ? 9 : 1;
    return in_b + px * a.cs_out * 2 * (a.dA_z ? 2 : 1) + (double)L.gx * L.gy * (T + 1) * 256 * 4;
}

int imk_launch_wgrad(const ImkWgradArgs &a, hipStream_t stream) {
    // the input block (uint8 image -> <= 8 channels, BatchNorm behind it): a streaming reduction, not a matrix-core kernel
    static const bool stem_off = []() { const char *e = getenv("IMK_STEM_WGRAD"); return e && e[0] == '0'; }();
    if (!stem_off && a.x.lmode == LM_U8 && a.ksize == 1 && a.dA_z && a.dA_coef && a.cs_out == 8 && a.x.cin <= 4 && a.n_split >= 1) {
        const long long n_pix = (long long)a.B * a.H * a.W;
        ImkProfScope prof(PF_WGRAD, (double)n_pix * (a.x.cin + 32) + (double)a.n_split * 2 * 1024, stream, imk_wgrad_flops(a));
        imk_klaunch(stem_wgrad_kernel, dim3(a.n_split), dim3(256), 0, stream, reinterpret_cast<const uint8_t *>(a.x.in), a.x.cin, a.x.u8_div, a.dA, a.dA_z, a.dA_coef,
                                                        n_pix, a.partial);
        IMK_LAUNCH_CHECK();
        return IMK_OK;
    }
    if (imk_wgrad_gemm_ok(a.x.lmode, a.dA_z != nullptr, a.ksize, a.x.cs_in, a.cs_out, (long long)a.B * a.H * a.W)) return imk_launch_wgrad_gemm(a, stream);
    WgradLaunch L{};
    int rc = plan_wgrad(a, L);
    if (rc) return rc;
    const dim3 grid(L.gx, L.gy);
    ImkProfScope prof(PF_WGRAD, wgrad_algorithmic_bytes(a, L), stream, imk_wgrad_flops(a));
    switch (a.x.lmode) {
#define IMK_WG(LM) do { if (a.ksize == 3) { if (a.dA_z) imk_klaunch(wgrad_mfma_kernel<LM, true, true>, dim3(grid), dim3(256), L.lds, stream, a, L.gm); \
                                           else imk_klaunch(wgrad_mfma_kernel<LM, false, true>, dim3(grid), dim3(256), L.lds, stream, a, L.gm); } \
                        else { if (a.dA_z) imk_klaunch(wgrad_mfma_kernel<LM, true, false>, dim3(grid), dim3(256), L.lds, stream, a, L.gm); \
                               else imk_klaunch(wgrad_mfma_kernel<LM, false, false>, dim3(grid), dim3(256), L.lds, stream, a, L.gm); } } while (0)
        case LM_RAW: IMK_WG(LM_RAW); break;
        case LM_AFFINE: IMK_WG(LM_AFFINE); break;
        case LM_POOL: IMK_WG(LM_POOL); break;
        case LM_UPADD: IMK_WG(LM_UPADD); break;
        default: IMK_WG(LM_U8); break;
#undef IMK_WG
    }
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}

size_t imk_packed_conv_halfs(int ksize, int cin, int cout, int transposed, bool pair) {
    if (transposed == 2) return 512;
    const int T = ksize == 3 ? 9 : 1;
    const int m_dim = transposed ? cin : cout, k_dim = transposed ? cout : cin;
    const int nc8 = imk_pad8(k_dim) / 8;
    if (pair) return (size_t)(T == 9 ? 3 * nc8 : (nc8 + 1) / 2) * 512;
    const int nc8p = imk_pass_chunks(nc8);
    const int ns = imk_cdiv_d(nc8, nc8p) * ((T * nc8p + 3) / 4);
    return (size_t)((m_dim + 15) / 16) * ns * 512;
}

