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
//   weight gradient: wgrad_gemm_kernel in imk_wgemm.hip (64 x 64 channels x 9 taps per workgroup).
//
// Reference layers replaced: Conv2D (+ the BatchNormalization / MaxPooling2D / UpSampling2D + add around it) of
// unet.py:11-43 and evalnet.py:8-21 and their gradients inside model.fit (functions.py:218).
#include <cstdlib>
#include <type_traits>
#include "imk_stage.h"

IMK_STAMP_TABLE(gemm)

namespace {

constexpr int G_PM = 4, G_WM = 2, G_WN = 2;   // pixel groups per wave; waves along pixels / output channels
constexpr int G_TH = G_PM * G_WM;             // tile rows (of 16 pixels)
constexpr int G_BM = 16 * G_TH;               // pixels per workgroup tile

struct GemmGeom {
    int tiles_x, tiles_y, n_sp, gy, mt_total;
    int nc8, nc8p, n_pass, nsp, ns_total;     // packed-weight geometry: chunks, chunks per pass, passes, k-steps per pass / in all
    int spp, cps, n_stage, ps;                // passes per stage, chunks per stage, stages, LDS pixel pitch in chunks (imk_lds_pitch)
    int nc8_2, nc8p_2, ns_2, mt_2;            // CH2: the chained 1x1's k geometry (chunks, chunks per pass, k-steps) and output tiles
};

// CH2 (inference, round 3): the block's Conv1x1+ReLU as a second GEMM on the output tile while it sits in LDS -- the workgroup
// holds ALL output channels of its 128 pixels (gy == 1: cout <= BN), so out2 = relu(W2 . tile + b2) needs nothing else: B operand =
// 16-byte reads of s_out rows (the k order of the 1x1's regular pack), A = that pack from global, result back into s_out and out
// through the same coalesced sweep.  The 3x3's output is never written: one launch and one tensor round trip less per block.
// AD: k-steps the weight fragments run ahead of their MFMAs.  1 (two register buffers) for launches that fill the chip -- three
// workgroups per CU hide the L2 round trip by themselves, and the registers of a deeper ring would cost the third one (round 6:
// two steps ahead everywhere was 5 % SLOWER).  3 (a ring of four; 2 where the load mode's staging registers leave no room) for the launches that cannot: the deep levels at batch 32 run one
// or two workgroups per CU, and the in-kernel stamps show their k-steps taking 467 ns for 115 ns of MFMA -- every step waited for
// the fragment requested one step earlier (tests/gpu_probe/stamps.py, Cityscapes alpha 2: 16 stages of 4.2 us in a 256-workgroup dgrad).
template <int LM, bool KS3, int PN, bool NC4, bool CH2 = false, int AD = 1>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(ImkConvArgs a, GemmGeom gm) {
    constexpr int NT = 256, PM = G_PM, WN = G_WN, TH = G_TH;
    constexpr int BN = 16 * PN * WN;
    constexpr int halo = KS3 ? 1 : 0, T = KS3 ? 9 : 1;
    constexpr int HT = TH + 2 * halo, WT = TW + 2 * halo;
    constexpr int NX = KS3 ? 3 : 4;           // staging items per thread: 10 x 18 pixels x 4 chunks / 128 pixels x 8 chunks
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    // workgroup id -> (spatial tile, output-channel group): the groups of one tile are consecutive on one XCD
    // (Round 5, measured and removed: a persistent walk -- at most 512 / 1 024 workgroups, each taking tile id, id + grid, ..., the next
    //  tile's first-stage loads and weight fragments issued during the current tile's second-to-last stage so that they land under the
    //  epilogue; for the instantiations that have the registers, the others one tile per workgroup.  Bit-identical; Cityscapes alpha 2
    //  step 4.89-4.91 ms against 4.85 for this form, 128-image inference 4.25 against 4.19-4.25, SUIM / HeLa / EvalNet within +-0.5 %
    //  (profiles/r05_ab6.txt): with two workgroups per CU the other workgroup already fills a tile's prologue.  The affine table
    //  moved behind the epilogue's LDS and the walk's state alone cost the one-shot path 4 % in inference.)
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
    IMK_STAMP_BEGIN(gemm, 60000 + LM * 100 + (KS3 ? 10 : 0) + PN + (NC4 ? 0 : 1));

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
    // Weight fragments through a BUFFER load: resource = the packed weights (scalar registers), scalar offset = channel tile m at
    // k-step F, vector offset = lane * 16 -- `buffer_load_dwordx4 v, v_lane16, s[rsrc], s_off offen`: no vector address arithmetic in
    // the k-loop.  (Written as pointer arithmetic the compiler formed a 64-bit VECTOR address per fragment and k-step -- v_lshl_add_u64
    // into the fragment's own destination registers, each behind an s_waitcnt vmcnt for the load still in flight there: round 6.)
    const int ct0 = by * (BN / 16);
    const int wn_u = __builtin_amdgcn_readfirstlane(wn);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16 *>(a.wpk), 0, 0x7fffffff, 0x00020000);
    int wso[PN];
#pragma unroll
    for (int m = 0; m < PN; ++m) {
        int ct = ct0 + wn_u * PN + m;
        if (ct >= gm.mt_total) ct = gm.mt_total - 1;     // padding tiles of the last group: computed, never stored
        wso[m] = ct * ns_total * 1024;                   // < 2^31: the widest pack (512 x 512 x 9 fp16) is 4.7 MB
    }
    const int lane16 = lane * 16;
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
    auto loadA = [&](f16x8 (&af)[PN], int F) {
        const int fo = min(F, ns_total - 1) * 1024;
#pragma unroll
        for (int m = 0; m < PN; ++m) {
            const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16, wso[m] + fo, 0);
            af[m] = __builtin_bit_cast(f16x8, v);
        }
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
    f16x8 af[AD + 1][PN];                                // ring of weight-fragment buffers: fragment i lives in slot i % (AD + 1)
    f16x8 (&a0)[PN] = af[0], (&a1)[PN] = af[1];
    if constexpr (AD == 1) loadA(af[0], 0);
    __syncthreads();                                     // affine table visible
    commit(0, s_tile0);
    issue(min(1, n_stage - 1));
    if constexpr (AD > 1) {
        // the ring is filled BEHIND the second tile's requests, as in every later stage: the first stage's wait for that tile is then
        // "all but the ring's AD x PN loads" like everyone else's (requested in front of them it had to be vmcnt(0) on the path into
        // the loop, and the compiler's merged count for the loop's top drained the ring at EVERY stage)
#pragma unroll
        for (int i = 0; i < AD; ++i) loadA(af[i], i);
    }
    __syncthreads();
    IMK_STAMP(1);
    const int mg_q = (65536 + nc8p - 1) / nc8p;          // q / nc8p as multiply-shift (q < 40)
    int F = 0;                                           // flat k-step index of the stage's first step
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    // One stage = the k-steps of its passes against the staged tile `cur`.  NC4 (4 chunks per pass: every channel count that is
    // a multiple of 32, and most others): k-slot (step s, lane group g) is (tap s, chunk g) -- the tile offset of a step is a
    // SCALAR (tap shift + pass) and the lane's chunk a constant folded into its base; otherwise the general (tap, chunk)
    // decomposition per lane.  The weights of invalid k-slots are zero and every staged chunk is finite (dead chunks are
    // written as zeros), so NC4 needs no validity masks.
    auto run_stage = [&](int st, const uint8_t *cur) {
        const int np = min(spp, gm.n_pass - st * spp);   // passes of this stage
        // k-steps.  A 3x3 with 4 chunks per pass has ONE pass per stage of exactly 9 steps (tap s = step s): a compile-time count, so the
        // deep-look-ahead form's stage body is straight-line code and the compiler can count the loads in flight exactly -- behind a
        // loop of unknown length its s_waitcnt for the NEXT stage's staged tile was vmcnt(0), which drained the weight ring every stage
        constexpr bool NK9 = KS3 && NC4 && AD > 1;
        const int nk = NK9 ? 9 : np * nsp;
        int pi = 0, s = 0;                               // (pass in stage, step in pass) of the next pixel-operand read
        int bg[PM];
#pragma unroll
        for (int p = 0; p < PM; ++p) bg[p] = base[p] + (NC4 ? g * 16 : 0);
        auto loadB = [&](f16x8 (&bf)[PM]) {
            int off;
            if constexpr (NC4) {
                const int ty = KS3 ? (s * 11) >> 5 : 0, tx = KS3 ? s - 3 * ty : 0;          // tap = s (uniform)
                off = ((ty * WT + tx) * ps + pi * 4) * 16;
                if (pi >= np) off = 0;                                                       // look-ahead past the stage: unused
            } else {
                const int q = 4 * s + g;
                const int tap = (q * mg_q) >> 16, c8 = q - tap * nc8p;
                const int ty = KS3 ? (tap * 11) >> 5 : 0, tx = KS3 ? tap - 3 * ty : 0;
                const bool vq = (q < T * nc8p) && ((st * spp + pi) * nc8p + c8 < nc8) && (pi < np);
                off = vq ? ((ty * WT + tx) * ps + pi * nc8p + c8) * 16 : 0;   // invalid k-slots: zero weights, any finite data
            }
#pragma unroll
            for (int p = 0; p < PM; ++p) bf[p] = *reinterpret_cast<const f16x8 *>(cur + bg[p] + off);
            if (++s == nsp) { s = 0; ++pi; }
        };
        if constexpr (AD > 1) {
            // Ring of NA = AD + 1 buffers.  Fragment r of the stage (r = 0: its first k-step) lives in slot r % NA; slots 0 .. AD - 1 hold
            // fragments 0 .. AD - 1 on entry.  Step j requests fragment j + AD into the slot step j - 1 has just consumed.  On exit the
            // next stage's first AD fragments sit in slots nk % NA ...: one register rotation per stage (a few dozen moves for a stage's
            // 144 MFMAs).  Unrolled by lcm(NA, 2) so that every slot and the pixel operand's double buffer are compile-time names.
            constexpr int NA = AD + 1, U = (NA % 2 == 0) ? NA : 2 * NA;
            f16x8 bb[2][PM];
            loadB(bb[0]);
            // whole groups of U steps without a branch inside (the compiler counts the loads in flight exactly only on straight-line
            // code: with a conditional step in the group its s_waitcnt assumed the shortest path and drained the ring), then the
            // remainder as a NEST of conditions, so that step u + 1 is only reachable through step u
            auto step = [&](auto U_T, int f) {
                constexpr int u = decltype(U_T)::value;
                loadA(af[(u + AD) % NA], F + f + u + AD);
                loadB(bb[(u + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);       // the requests first, THEN the step's MFMAs (the scheduler sank the LDS reads to
                mma(af[u % NA], bb[u & 1]);              // their uses to save registers: four exposed LDS round trips per step)
                __builtin_amdgcn_sched_barrier(0);
            };
            int f = 0;
            for (; f + U <= nk; f += U) {
                step(std::integral_constant<int, 0>{}, f); step(std::integral_constant<int, 1>{}, f);
                step(std::integral_constant<int, 2>{}, f); step(std::integral_constant<int, 3>{}, f);
                if constexpr (U == 6) { step(std::integral_constant<int, 4>{}, f); step(std::integral_constant<int, 5>{}, f); }
            }
            const int rem = nk - f;
            if (rem >= 1) {
                step(std::integral_constant<int, 0>{}, f);
                if (rem >= 2) {
                    step(std::integral_constant<int, 1>{}, f);
                    if (rem >= 3) {
                        step(std::integral_constant<int, 2>{}, f);
                        if constexpr (U == 6) {
                            if (rem >= 4) {
                                step(std::integral_constant<int, 3>{}, f);
                                if (rem >= 5) step(std::integral_constant<int, 4>{}, f);
                            }
                        }
                    }
                }
            }
            const int rot = nk % NA;                     // workgroup-uniform
#pragma unroll
            for (int r = 1; r < NA; ++r) {
                if (rot == r) {
#pragma unroll
                    for (int m = 0; m < PN; ++m) {
                        f16x8 t[NA];
#pragma unroll
                        for (int i = 0; i < NA; ++i) t[i] = af[i][m];
#pragma unroll
                        for (int i = 0; i < NA; ++i) af[i][m] = t[(i + r) % NA];
                    }
                }
            }
            F += nk;
            return;
        }
        f16x8 b0[PM], b1[PM];
        loadB(b0);
        // (Round 6, measured and removed: weight fragments TWO k-steps ahead of their MFMAs -- three register buffers, loads stopped
        //  at the next stage's first two fragments, a select-rotation at the stage end; bit-identical.  +31 registers took the 128-channel
        //  forms from 3 to 2 waves per SIMD: Cityscapes alpha 2 step 4.63 -> 4.87 ms, alpha 1.25 3.61 -> 3.83, 128-image inference 4.13 ->
        //  4.42, EvalNet 2.02 -> 2.09 (profiles/r06_ab_adepth.txt).  One k-step of look-ahead with three resident waves hides the L2.)
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
    };
    for (int st = 0; st < n_stage; ++st) {
        const uint8_t *cur = (st & 1) ? s_tile1 : s_tile0;
        uint8_t *nxt = (st & 1) ? s_tile0 : s_tile1;
        if (st + 1 < n_stage) commit(st + 1, nxt);       // its loads were issued a whole stage ago
        issue(min(st + 2, n_stage - 1));                 // in flight during this stage's MFMAs
        run_stage(st, cur);
        if (st == 0) IMK_STAMP(2);
        __syncthreads();
        if (st == 0) IMK_STAMP(3);
    }
    IMK_STAMP(4);

    // ---- epilogue ------------------------------------------------------------------------------------------------------
    constexpr int OP = BN + 8;                           // halfs per pixel of the output tile in LDS (17 / 9 chunks: odd)
    f16 *s_out = reinterpret_cast<f16 *>(smem);
    const int cs_o = CH2 ? a.cs_out2 : a.cs_out;
    f16 *const out_t = CH2 ? a.out2 : a.out;
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
    if constexpr (CH2) {
        if (a.out) {                                     // training: the backward pass reads the 3x3's output -- one coalesced sweep
            constexpr int CPP1 = BN / 8, RPI1 = NT / CPP1, IT1 = G_BM / RPI1;
            const int cj1 = t % CPP1, prow1 = t / CPP1;
            if (cj1 * 8 < a.cs_out) {
                const unsigned pitch1 = (unsigned)a.cs_out * 2u;
                char *ob = const_cast<char *>(pix_base(a.out, tc.b, H, W, tc.ty0, tc.tx0, pitch1)) + cj1 * 16;
                const int my1 = H - 1 - tc.ty0, mx1 = W - 1 - tc.tx0;
#pragma unroll
                for (int k = 0; k < IT1; ++k) {
                    const int pixel = prow1 + k * RPI1, py = pixel >> 4, px = pixel & 15;
                    if (py <= my1 && px <= mx1)
                        *reinterpret_cast<f16x8 *>(ob + __umul24(__umul24(py, W) + px, pitch1)) = *reinterpret_cast<const f16x8 *>(s_out + pixel * OP + cj1 * 8);
                }
            }
        }
        f32x4 acc2[PN][PM];
#pragma unroll
        for (int m = 0; m < PN; ++m)
#pragma unroll
            for (int p = 0; p < PM; ++p) acc2[m][p] = f32x4{0, 0, 0, 0};
        for (int ks = 0; ks < gm.ns_2; ++ks) {
            int c8 = ks * gm.nc8p_2 + g;
            if (g >= gm.nc8p_2 || c8 >= gm.nc8_2) c8 = 0;          // zero weights in those k-slots: any finite chunk
            f16x8 af[PN], bf[PM];
#pragma unroll
            for (int m = 0; m < PN; ++m) {
                const int mt = wn * PN + m;
                af[m] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                if (mt < gm.mt_2) af[m] = *reinterpret_cast<const f16x8 *>(a.wpk2 + ((size_t)(mt * gm.ns_2 + ks) * 64 + lane) * 8);
            }
#pragma unroll
            for (int p = 0; p < PM; ++p) bf[p] = *reinterpret_cast<const f16x8 *>(s_out + ((wm * PM + p) * 16 + n) * OP + c8 * 8);
#pragma unroll
            for (int m = 0; m < PN; ++m)
#pragma unroll
                for (int p = 0; p < PM; ++p) acc2[m][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[m], bf[p], acc2[m][p], 0, 0, 0);
        }
        float bias2[PN][4];
#pragma unroll
        for (int m = 0; m < PN; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int co = (wn * PN + m) * 16 + 4 * g + r; bias2[m][r] = co < a.cout2 ? a.bias2[co] : 0.f; }
        __syncthreads();                                 // every wave has read the 3x3's tile
#pragma unroll
        for (int m = 0; m < PN; ++m)
#pragma unroll
            for (int p = 0; p < PM; ++p) {
                f16x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (f16)fmaxf(acc2[m][p][r] + bias2[m][r], 0.f);
                *reinterpret_cast<f16x4 *>(s_out + ((wm * PM + p) * 16 + n) * OP + (wn * PN + m) * 16 + 4 * g) = v;
            }
        __syncthreads();
    }
    IMK_STAMP(5);
    // The tile leaves in sweeps of 16-byte chunks.  BNV = channels a sweep covers = the tile's width, except for a chained 1x1
    // with at most 64 outputs behind a 128-wide 3x3 (CH2): its sweep is the one its own 64-wide launch would make, so that the
    // BatchNorm statistics -- per-thread sums over the sweep's pixels, then a fixed-order reduction -- come out bit for bit the same.
    auto finish = [&](auto BNV_T) {
    constexpr int BNV = decltype(BNV_T)::value;
    constexpr int CPP = BNV / 8, RPI = NT / CPP, IT = G_BM / RPI;   // chunks per pixel, pixels per sweep, sweeps
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
        char *ob = const_cast<char *>(pix_base(out_t, tc.b, H, W, tc.ty0, tc.tx0, pitch)) + cho;
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
    const bool dystat = !CH2 && (a.epi != EP_RELU) && a.dystat_z && a.stats_partial;
    const bool want_stats = ((a.epi == EP_RELU) && a.stats_partial) || dystat;      // CH2: the statistics of the 1x1's output
    if (a.epi == EP_RELU) { if (want_stats) sweep(I0{}, I1{}); else sweep(I0{}, I0{}); }
    else if (a.epi == EP_MASK) { if (dystat) sweep(I2{}, I2{}); else sweep(I2{}, I0{}); }
    else { if (dystat) sweep(I1{}, I2{}); else sweep(I1{}, I0{}); }
    IMK_STAMP(6);
    if (want_stats) {                                    // workgroup-uniform
        __syncthreads();                                 // everyone is done reading the output tile
        float *s_red = reinterpret_cast<float *>(smem);  // [2][RPI][BNV]
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s_red[(0 * RPI + prow) * BNV + cj * 8 + e] = s1[e];
            s_red[(1 * RPI + prow) * BNV + cj * 8 + e] = s2[e];
        }
        __syncthreads();
        for (int i = t; i < 2 * BNV; i += NT) {
            const int which = i / BNV, c = i - which * BNV;
            const int co = ct0 * 16 + c;
            if (co < cs_o) {
                float v = 0.f;
                for (int r = 0; r < RPI; ++r) v += s_red[(which * RPI + r) * BNV + c];
                a.stats_partial[(size_t)sp * 2 * cs_o + which * cs_o + co] = v;
            }
        }
    }
    };
    if (CH2 && BN == 128 && cs_o <= 64) finish(std::integral_constant<int, 64>{});
    else finish(std::integral_constant<int, BN>{});
    IMK_STAMP_END(7);
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
    static const int force_pn = []() { const char *e = getenv("IMK_GEMM_PN"); return e ? atoi(e) : 0; }();
    if (force_pn == 2 || (force_pn == 4 && mt_total > 4)) pn = force_pn;
    if (a.wpk2) {            // chained 1x1: one workgroup holds every output channel of its pixels
        pn = mt_total > 4 ? 4 : 2;
        const int nc8_2 = a.cs_out / 8;
        gm.nc8_2 = nc8_2; gm.nc8p_2 = imk_pass_chunks(nc8_2); gm.ns_2 = imk_cdiv_d(nc8_2, gm.nc8p_2);
        gm.mt_2 = (a.cout2 + 15) / 16;
    }
    const int bn = 16 * pn * G_WN;
    const int halo = ks3 ? 1 : 0;
    gm.tiles_x = imk_cdiv(a.W, TW); gm.tiles_y = imk_cdiv(a.H, G_TH);
    gm.n_sp = a.B * gm.tiles_x * gm.tiles_y;
    gm.gy = imk_cdiv_d(mt_total, bn / 16);
    gm.mt_total = mt_total;
    gm.nc8 = nc8; gm.nc8p = nc8p; gm.n_pass = n_pass; gm.nsp = nsp; gm.ns_total = n_pass * nsp;
    gm.spp = spp; gm.cps = cps; gm.n_stage = imk_cdiv_d(n_pass, spp); gm.ps = imk_lds_pitch(cps);
    const size_t tile_bytes = (size_t)(G_TH + 2 * halo) * (TW + 2 * halo) * gm.ps * 16;
    lds = 2 * tile_bytes + 4 * (size_t)a.x.cs_in * sizeof(float);
    const size_t out_bytes = (size_t)G_BM * (bn + 8) * sizeof(f16);
    const size_t red_bytes = 2 * (size_t)256 * 8 * sizeof(float);
    if (lds < out_bytes) lds = out_bytes;
    if (lds < red_bytes) lds = red_bytes;
    if (lds > 64 * 1024) return IMK_EUNSUPPORTED;
    // workgroups per compute unit: capped by padding the LDS request (160 KB / w), IMK_GEMM_W = 2 / 3 / 4 (0: what fits)
    static const int force_w = []() { const char *e = getenv("IMK_GEMM_W"); return e ? atoi(e) : 0; }();
    if (force_w >= 1 && force_w <= 3) {
        const size_t want = (size_t)160 * 1024 / (force_w + 1) + 1024;
        if (lds < want && want <= 64 * 1024) lds = want;
    }
    grid = imk_cdiv_d(gm.n_sp, 8) * 8 * gm.gy;
    return IMK_OK;
}

// launches of at most this many workgroups take the deep-look-ahead form (AD > 1): few workgroups per compute unit are resident,
// nothing else hides the weight fragments' L2 round trip (IMK_GEMM_AD3_WGS; 0 = never).  Measured on one box (profiles/r06_ab_ad3.txt),
// training step old -> 512 / 1 024 / 1 600 / 6 000: ISIC 0.958 -> 0.930 / 0.936 / 0.943 / 0.936 ms, SUIM 1.630 -> 1.593 / 1.593 / 1.600 /
// 1.596, HeLa 1.61 -> 1.588 / 1.584 / 1.593 / 1.592, EvalNet 2.04 -> 2.004 / 1.997 / 1.997 / 1.995, Cityscapes alpha 2 4.63 -> 4.618 /
// 4.616 / 4.608 / 4.613, alpha 1.25 3.61 -> 3.619 / 3.619 / 3.618 / 3.677: bit-identical everywhere
int gemm_ad3_max_wgs() {
    static const int v = []() { const char *e = getenv("IMK_GEMM_AD3_WGS"); return e ? atoi(e) : 1024; }();
    return v;
}

// ring depth of the deep-look-ahead form per load mode: what fits 256 registers without scratch (pool / upsample + add on load hold
// 4 / 2 raw chunks per staged item)
// (a 3x3 stage of 9 k-steps takes the ring of THREE: 9 % 3 == 0, the ring needs no rotation at the stage's end -- a rotation moves
// registers whose loads are still in flight, i.e. waits for them)
template <int LM, bool KS3, bool NC4> constexpr int gemm_deep_ad() { return (KS3 && NC4) || LM == LM_POOL || LM == LM_UPADD ? 2 : 3; }

template <int LM>
int launch_conv_gemm_chain(const ImkConvArgs &a, const GemmGeom &gm, int pn, size_t lds, int grid, hipStream_t stream) {
    if (grid <= gemm_ad3_max_wgs()) {
        if (gm.nc8p == 4) {
            constexpr int AD = gemm_deep_ad<LM, true, true>();
            if (pn == 4) imk_klaunch(conv_gemm_kernel<LM, true, 4, true, true, AD>, dim3(grid), dim3(256), lds, stream, a, gm);
            else imk_klaunch(conv_gemm_kernel<LM, true, 2, true, true, AD>, dim3(grid), dim3(256), lds, stream, a, gm);
        } else {
            constexpr int AD = gemm_deep_ad<LM, true, false>();
            if (pn == 4) imk_klaunch(conv_gemm_kernel<LM, true, 4, false, true, AD>, dim3(grid), dim3(256), lds, stream, a, gm);
            else imk_klaunch(conv_gemm_kernel<LM, true, 2, false, true, AD>, dim3(grid), dim3(256), lds, stream, a, gm);
        }
        return IMK_OK;
    }
    if (gm.nc8p == 4) {
        if (pn == 4) imk_klaunch(conv_gemm_kernel<LM, true, 4, true, true>, dim3(grid), dim3(256), lds, stream, a, gm);
        else imk_klaunch(conv_gemm_kernel<LM, true, 2, true, true>, dim3(grid), dim3(256), lds, stream, a, gm);
    } else {
        if (pn == 4) imk_klaunch(conv_gemm_kernel<LM, true, 4, false, true>, dim3(grid), dim3(256), lds, stream, a, gm);
        else imk_klaunch(conv_gemm_kernel<LM, true, 2, false, true>, dim3(grid), dim3(256), lds, stream, a, gm);
    }
    return IMK_OK;
}

template <int LM, bool KS3>
int launch_conv_gemm_k(const ImkConvArgs &a, const GemmGeom &gm, int pn, size_t lds, int grid, hipStream_t stream) {
    if (grid <= gemm_ad3_max_wgs()) {
        if (gm.nc8p == 4) {
            constexpr int AD = gemm_deep_ad<LM, KS3, true>();
            if (pn == 4) imk_klaunch(conv_gemm_kernel<LM, KS3, 4, true, false, AD>, dim3(grid), dim3(256), lds, stream, a, gm);
            else imk_klaunch(conv_gemm_kernel<LM, KS3, 2, true, false, AD>, dim3(grid), dim3(256), lds, stream, a, gm);
        } else {
            constexpr int AD = gemm_deep_ad<LM, KS3, false>();
            if (pn == 4) imk_klaunch(conv_gemm_kernel<LM, KS3, 4, false, false, AD>, dim3(grid), dim3(256), lds, stream, a, gm);
            else imk_klaunch(conv_gemm_kernel<LM, KS3, 2, false, false, AD>, dim3(grid), dim3(256), lds, stream, a, gm);
        }
        return IMK_OK;
    }
    if (gm.nc8p == 4) {      // 4 chunks per pass: scalar tile offsets in the k-loop
        if (pn == 4) imk_klaunch(conv_gemm_kernel<LM, KS3, 4, true>, dim3(grid), dim3(256), lds, stream, a, gm);
        else imk_klaunch(conv_gemm_kernel<LM, KS3, 2, true>, dim3(grid), dim3(256), lds, stream, a, gm);
    } else {
        if (pn == 4) imk_klaunch(conv_gemm_kernel<LM, KS3, 4, false>, dim3(grid), dim3(256), lds, stream, a, gm);
        else imk_klaunch(conv_gemm_kernel<LM, KS3, 2, false>, dim3(grid), dim3(256), lds, stream, a, gm);
    }
    return IMK_OK;
}

}  // namespace

// Which launches the GEMM-class kernel takes: more than 32 channels on a side (the per-tile kernel's layers), plain
// (unchained) convs whose input is a fp16 tensor.
bool imk_conv_gemm_ok(const ImkConvArgs &a) {
    if (!gemm_env_on()) return false;
    if (a.wpk2 || a.pre_wpk || a.wg_partial) return false;
    // (Round 5, measured and removed: the pooled-input 3x3 with 16 -> 32 channels -- the second encoder block at alpha 1, the third at
    //  alpha 0.5; on the per-tile kernel the largest single item of an alpha = 1 inference call, 1.0-1.2 TB/s -- on THIS kernel with
    //  half of its 64-channel tile padding: bit-identical, and slower: 584-image ISIC forward 1.80 -> 1.89 ms, 128-image SUIM 1.05 ->
    //  1.13, HeLa 1.01 -> 1.08, Cityscapes 1.68 -> 1.77; training steps +0.5-1 %.  profiles/r05_notes.md.)
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

// Conv3x3+ReLU -> Conv1x1+ReLU in one launch of the GEMM-class kernel: the 3x3 is one this kernel takes, both convs have at
// most 128 output channels, the 1x1's pack is one k-step per channel pass.  Inference: no stored intermediate, no statistics;
// training: the intermediate is stored (one more sweep of the tile) and the BatchNorm statistics are those of the 1x1's output,
// one row per workgroup over the same 128 pixels, summed in the order of the 1x1's own launch (see `finish`): bit-identical to
// two launches
bool imk_conv_gemm_chain_ok(const ImkConvArgs &a) {
    static const bool off = []() { const char *e = getenv("IMK_GEMM_CHAIN"); return e && e[0] == '0'; }();
    static const bool train_off = []() { const char *e = getenv("IMK_GEMM_CHAIN_TRAIN"); return e && e[0] == '0'; }();
    if (off || !a.wpk2 || !a.out2 || a.epi != EP_RELU || a.ksize != 3) return false;
    if ((a.out || a.stats_partial) && train_off) return false;      // training: the intermediate is stored, statistics on the 1x1's output
    if (a.x.lmode != LM_POOL && a.x.lmode != LM_AFFINE) return false;
    if (a.cout > 128 || a.cout2 > 128 || a.cs_out2 > (a.cout > 64 ? 128 : 64)) return false;
    ImkConvArgs plain = a;
    plain.wpk2 = nullptr; plain.out = a.out2;
    return imk_conv_gemm_ok(plain);
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
    if (a.wpk2) {
        if (!imk_conv_gemm_chain_ok(a) || gm.gy != 1) return IMK_EUNSUPPORTED;
        rc = a.x.lmode == LM_POOL ? launch_conv_gemm_chain<LM_POOL>(a, gm, pn, lds, grid, stream)
                                  : launch_conv_gemm_chain<LM_AFFINE>(a, gm, pn, lds, grid, stream);
        if (rc) return rc;
        IMK_LAUNCH_CHECK();
        if (a.stats_rows) *a.stats_rows = gm.n_sp;
        return IMK_OK;
    }
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

