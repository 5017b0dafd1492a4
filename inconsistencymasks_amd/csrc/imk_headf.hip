// Training step of a SOFTMAX output layer in one pass (multi-class heads: SUIM 9 outputs, Cityscapes 35): what
// head_loss_kernel (imk_elem.hip) + the head's dgrad launch + its weight-gradient launch did in three -- the fp32
// 1x1 conv on the last BatchNorm's output (unet.py:63, dtype float32), softmax, categorical cross-entropy
// (functions.py:303: CategoricalCrossentropy on the softmax activation's logits), d(loss * scale)/d(logits), the
// gradient w.r.t. that BatchNorm's output with its BatchNorm-backward statistics, and the output layer's weight / bias
// gradient partials -- reading the last activation ONCE and never storing the [pixels, K] logit gradient.
// At 35 classes the three launches were 15 % of a Cityscapes training step (fp32 dot products on the vector units, the
// logit gradient written once and read twice: 80 bytes per pixel each way).
//
// All products run on the matrix cores, laid out so that no value changes lanes between the stages:
//   logits^T [class][pixel] = W^T . x^T + b     v_mfma_f32_16x16x32_f16 with the fp32 weights split into fp16 hi + lo halves
//       (two MFMAs per 16-class tile, fp32 accumulation: the reference's float32 output layer to ~1e-7 relative):
//       A = W^T (rows = classes of a 16-class tile), B = x^T: lane (pixel p = lane & 15, g = lane >> 4) supplies the channels
//       8 g + j it loaded itself (one 16-byte load) and receives the classes 16 kt + 4 g + r of ITS pixel: softmax max / sum =
//       registers + two shuffles over the 4 lanes of a pixel;
//   dy^T [channel][pixel] = W . dlogit^T        v_mfma_f32_16x16x32_f16: the contraction runs over the classes = the ROW index
//       of the logits tile, so the fp16 logit gradients are the B operand as they stand (k-slot (g, j) <-> class
//       16 (2 s + (j >> 2)) + 4 g + (j & 3); the A fragments of W are built in that order), and the result lands on the
//       channels 16 ct + 4 g + r of the lane's pixel: exactly the z values it holds for the BatchNorm-backward statistics;
//   dW [channel][class] = x^T . dlogit          contraction over pixels (lanes): x and dlogit go through a wave-private LDS
//       image [tile][pixel][16] and come back with transposed reads (the weight-gradient kernels' pattern).
// Fixed reduction orders everywhere (per-wave accumulators -> per-workgroup partial rows): bit-reproducible.
#include "imk_stage.h"

IMK_STAMP_TABLE(headf)

namespace {

struct HeadCceArgs {
    const f16 *z;                 // [n_pix][CS] last decoder activation (pre-BatchNorm)
    const float *sc, *sh;         // its BatchNorm scale / shift (batch statistics) [CS]
    const float *w, *bias;        // output layer: kernel [cin][K] fp32, bias [K]
    int cin, cs, K;               // cs: channel stride of z / dy (8, 16, 24 or 32; the kernel's CS = 16 or 32 is its tile capacity)
    long long n_pix;
    const uint8_t *y;             // class ids [n_pix]
    const ImkCtl *ctl;
    float *stats;
    f16 *dy;                      // [n_pix][CS] gradient w.r.t. the BatchNorm output
    float *loss_partial;          // [grid]
    float *dystat_partial;        // [grid][2 * CS]: sum dy, sum dy * z per channel
    float *wg_partial;            // [grid][NCT * cot_n][2][256]: weight-gradient partials in wgf_stage1's layout
    int cot_n;                    // 16-class tiles of the weight-gradient layout = ceil(pad8(K) / 16)
};

template <int CS, int KT>
__global__ __launch_bounds__(256) void head_cce_fused_kernel(HeadCceArgs a) {
    IMK_STAMP_BEGIN(headf, 90001);
    constexpr int NCT = CS / 16;                 // 16-channel tiles
    constexpr int NS = (KT + 1) / 2;             // 32-class k-steps of the dgrad
    constexpr int KT2 = 2 * NS;                  // class tiles incl. the zero tile that completes the last k-step
    constexpr int H16 = WG_STRIDE_H;
    // wave-private LDS image of 64 pixels: x slices [NCT][64][16] and dlogit slices [KT][64][16] (fp16); reused at the end
    constexpr int IMG_H = (NCT + KT) * 64 * H16;                     // halfs per wave
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, p16 = lane & 15, g = lane >> 4, qq = p16 >> 2, pp = p16 & 3;
    f16 *s_x = reinterpret_cast<f16 *>(smem) + wave * IMG_H;
    f16 *s_g = s_x + NCT * 64 * H16;
    const int K = a.K, cin = a.cin;
    const float S = a.ctl->loss_scale;
    if (blockIdx.x == 0 && t == 0) { a.stats[1] = 0.f; a.stats[2] = S; a.stats[3] = (float)a.ctl->step; }   // as head_loss_kernel
    const float gs = S / (float)a.n_pix;

    // ---- operands that live in registers for the whole kernel ----------------------------------------------------------
    // logits: ONE k-step of v_mfma_f32_16x16x32_f16 covers all (<= 32) input channels: A[class row p16][k-slot (g, j)] =
    // W[c = 8 g + j][16 kt + p16], the fp32 weight split into two halves w = hi + lo (hi = fp16(w), lo = fp16(w - hi)): the
    // products x * hi, x * lo are exact in fp32 and their sum carries ~22 bits of w -- the reference's float32 output layer to
    // 1e-7 relative, at 1/8 of the matrix-core time of the fp32 MFMA chain (8 dependent 16x16x4 steps per class tile before).
    f16x8 whi[KT], wlo[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = 8 * g + j, k = 16 * kt + p16;
            const float w = (c < cin && k < K) ? a.w[(size_t)c * K + k] : 0.f;
            whi[kt][j] = (f16)w;
            wlo[kt][j] = (f16)(w - (float)whi[kt][j]);
        }
    float sc8[8], sh8[8];          // BatchNorm of the channels 8 g + j (the logits' B operand)
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int c = 8 * g + j; sc8[j] = c < a.cs ? a.sc[c] : 0.f; sh8[j] = c < a.cs ? a.sh[c] : 0.f; }
    // dgrad: A[channel row p16][k-slot (g, j)] of k-step s = fp16(W[16 ct + p16][class 16 (2 s + (j >> 2)) + 4 g + (j & 3)])
    f16x8 wd[NCT][NS];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = 16 * ct + p16, k = 16 * (2 * s + (j >> 2)) + 4 * g + (j & 3);
                wd[ct][s][j] = (f16)((c < cin && k < K) ? a.w[(size_t)c * K + k] : 0.f);
            }
    float bias_r[KT][4];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int k = 16 * kt + 4 * g + r; bias_r[kt][r] = k < K ? a.bias[k] : 0.f; }

    f32x4 accw[NCT][KT], accb = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) accw[ct][kt] = f32x4{0, 0, 0, 0};
    float s1[NCT][4], s2[NCT][4], loss = 0.f;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[ct][r] = s2[ct][r] = 0.f;

    // ---- 64 pixels (4 units of 16) per wave and iteration ----------------------------------------------------------------
    const long long n_grp = (a.n_pix + 63) / 64;
    for (long long grp = (long long)blockIdx.x * 4 + wave; grp < n_grp; grp += (long long)gridDim.x * 4) {
        f16x4 zr[4][NCT];
        f16x8 z8[4];
        int yv[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {                 // all loads of the group first
            const long long px = grp * 64 + u * 16 + p16;
            ok[u] = px < a.n_pix;
            const long long pc = ok[u] ? px : a.n_pix - 1;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                zr[u][ct] = f16x4{0, 0, 0, 0};
                if (16 * ct + 4 * g < a.cs) zr[u][ct] = *reinterpret_cast<const f16x4 *>(a.z + pc * a.cs + 16 * ct + 4 * g);
            }
            z8[u] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (8 * g < a.cs) z8[u] = *reinterpret_cast<const f16x8 *>(a.z + pc * a.cs + 8 * g);
            yv[u] = a.y[pc];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // x = fp16(z * sc + sh): the head's input, as head_kernel / head_loss_kernel form it
            f16x8 x8;                                  // channels 8 g + j of the lane's pixel
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const f16x2 r2 = imk_affine2(f16x2{z8[u][j], z8[u][j + 1]}, f32x2{sc8[j], sc8[j + 1]}, f32x2{sh8[j], sh8[j + 1]});
                x8[j] = r2[0]; x8[j + 1] = r2[1];
            }
            // logits of the lane's pixel: classes 16 kt + 4 g + r
            f32x4 lg[KT];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                lg[kt] = f32x4{bias_r[kt][0], bias_r[kt][1], bias_r[kt][2], bias_r[kt][3]};
                lg[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[kt], x8, lg[kt], 0, 0, 0);
                lg[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo[kt], x8, lg[kt], 0, 0, 0);
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (16 * kt + 4 * g + r >= K) lg[kt][r] = -INFINITY;     // padding classes
                    mx = fmaxf(mx, lg[kt][r]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { lg[kt][r] = __builtin_amdgcn_exp2f((lg[kt][r] - mx) * 1.44269504088896f); sum += lg[kt][r]; }   // v_exp_f32: 2 instructions instead of expf's ~10 (the softmax was VALU-bound)
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.0f / sum;
            f16x4 gh[KT2];
            float p_hit = -1.f;
#pragma unroll
            for (int kt = 0; kt < KT2; ++kt) gh[kt] = f16x4{0, 0, 0, 0};
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 16 * kt + 4 * g + r;
                    const float pk = lg[kt][r] * inv;
                    const bool hit = k == yv[u];
                    p_hit = hit ? pk : p_hit;
                    gh[kt][r] = (k < K && ok[u]) ? (f16)(gs * (pk - (hit ? 1.0f : 0.0f))) : (f16)0.f;
                }
            // one logarithm per lane (the lane of the pixel's four that holds the labelled class), not one per class
            if (p_hit >= 0.f && ok[u]) loss += -logf(fmaxf(p_hit, 1.17549435e-38f));
            // dy^T = W . dlogit^T (fp16 operands, fp32 accumulation), rounded to fp16 like the dgrad launch it replaces
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                f32x4 d = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    f16x8 bf;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { bf[j] = gh[2 * s][j]; bf[4 + j] = gh[2 * s + 1][j]; }
                    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(wd[ct][s], bf, d, 0, 0, 0);
                }
                f16x4 dh;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dh[r] = (f16)d[r];
                    const float f = ok[u] ? (float)dh[r] : 0.f;
                    s1[ct][r] += f;
                    s2[ct][r] += f * (float)zr[u][ct][r];
                }
                if (ok[u] && 16 * ct + 4 * g < a.cs) *reinterpret_cast<f16x4 *>(a.dy + (grp * 64 + u * 16 + p16) * a.cs + 16 * ct + 4 * g) = dh;
            }
            // LDS image for the weight gradient: pixel u * 16 + p16, this lane's 4 channels / 4 classes of every tile
            if ((g >> 1) < NCT) {                      // slice (8 g) / 16, halfs 8 (g & 1) .. + 7 of the pixel's row
                f16x8 xs = x8;
                if (!ok[u]) xs = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                *reinterpret_cast<f16x8 *>(s_x + ((g >> 1) * 64 + u * 16 + p16) * H16 + 8 * (g & 1)) = xs;
            }
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) *reinterpret_cast<f16x4 *>(s_g + (kt * 64 + u * 16 + p16) * H16 + 4 * g) = gh[kt];
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        // dW[channel][class] += x^T . dlogit over the 64 pixels: two k-steps of 32 (units 2 kk, 2 kk + 1); k-slot <-> pixel map
        // of wgrad_mfma_kernel (a 32-lane half reads 8 consecutive pixels of one unit = one 256-byte bank row)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int row = 2 * kk + (g >> 1), xx = 4 * (g & 1) + qq;
            f16x8 af[NCT], bf[KT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const f16 *pa = s_x + (ct * 64 + row * 16 + xx) * H16 + 4 * pp;
                const h4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(LDS_PTR(h4, pa));
                const h4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(LDS_PTR(h4, pa + 8 * H16));
#pragma unroll
                for (int e = 0; e < 4; ++e) { af[ct][e] = (f16)a0[e]; af[ct][4 + e] = (f16)a1[e]; }
            }
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const f16 *pb = s_g + (kt * 64 + row * 16 + xx) * H16 + 4 * pp;
                const h4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(LDS_PTR(h4, pb));
                const h4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(LDS_PTR(h4, pb + 8 * H16));
#pragma unroll
                for (int e = 0; e < 4; ++e) { bf[kt][e] = (f16)b0[e]; bf[kt][4 + e] = (f16)b1[e]; }
            }
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) accw[ct][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ct], bf[kt], accw[ct][kt], 0, 0, 0);
                f16x8 e;                              // column sums -> bias gradient: A = ones in row kt
#pragma unroll
                for (int j = 0; j < 8; ++j) e[j] = (f16)(p16 == kt ? 1.0f : 0.0f);
                accb = __builtin_amdgcn_mfma_f32_16x16x32_f16(e, bf[kt], accb, 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    // ---- per-workgroup partials: loss, BatchNorm-backward statistics, weight gradient ------------------------------------
    __syncthreads();                                  // the LDS images are free
    float *s_red = reinterpret_cast<float *>(smem);   // [4 waves][(NCT * KT + 1) * 256] accumulators, then [4][2 * CS + 1]
    constexpr int NA = NCT * KT + 1;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) s_red[((wave * NA) + ct * KT + kt) * 256 + r * 64 + lane] = accw[ct][kt][r];
#pragma unroll
    for (int r = 0; r < 4; ++r) s_red[((wave * NA) + NCT * KT) * 256 + r * 64 + lane] = accb[r];
    float *s_st = s_red + 4 * NA * 256;
    loss = wave_sum<64>(loss);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v1 = wave_sum<16>(s1[ct][r]), v2 = wave_sum<16>(s2[ct][r]);
            if (p16 == 0) {
                s_st[wave * (2 * CS + 1) + 16 * ct + 4 * g + r] = v1;
                s_st[wave * (2 * CS + 1) + CS + 16 * ct + 4 * g + r] = v2;
            }
        }
    if (lane == 0) s_st[wave * (2 * CS + 1) + 2 * CS] = loss;
    __syncthreads();
    if (t < 2 * a.cs) {           // rows of the statistics are [2][cs]
        const int which = t >= a.cs, q = which * CS + (t - which * a.cs);
        a.dystat_partial[(size_t)blockIdx.x * 2 * a.cs + t] =
            (s_st[q] + s_st[(2 * CS + 1) + q]) + (s_st[2 * (2 * CS + 1) + q] + s_st[3 * (2 * CS + 1) + q]);
    }
    if (t == 0) a.loss_partial[blockIdx.x] = (s_st[2 * CS] + s_st[(2 * CS + 1) + 2 * CS]) + (s_st[2 * (2 * CS + 1) + 2 * CS] + s_st[3 * (2 * CS + 1) + 2 * CS]);
    // weight-gradient partial rows of this workgroup: [pair = ct * cot_n + kt][tap 0 | bias row][256]
    float *wp = a.wg_partial + (size_t)blockIdx.x * NCT * a.cot_n * 2 * 256;
    for (int i = t; i < NCT * a.cot_n * 2 * 256; i += 256) {
        const int e = i & 255, blk = i >> 8, tap = blk & 1, pair = blk >> 1;
        const int ct = pair / a.cot_n, kt = pair - ct * a.cot_n;
        float v = 0.f;
        if (kt < KT) {
            if (tap == 0) {
                const int idx = (ct * KT + kt) * 256 + e;
                v = (s_red[idx] + s_red[NA * 256 + idx]) + (s_red[2 * NA * 256 + idx] + s_red[3 * NA * 256 + idx]);
            } else if (ct == 0 && e < 16) {           // bias gradient of class 16 kt + e: row kt of accb = register kt of lanes 0-15
                const int idx = NCT * KT * 256 + kt * 64 + e;
                v = (s_red[idx] + s_red[NA * 256 + idx]) + (s_red[2 * NA * 256 + idx] + s_red[3 * NA * 256 + idx]);
            }
        }
        wp[i] = v;
    }
    IMK_STAMP_END(1);
}

// =====================================================================================================
// SIGMOID heads (ISIC: 1 map, HeLa: 3; functions.py:303 'mse'), round 4: the same one-pass scheme.  head_loss_kernel wrote the
// logit gradient as an 8-channel-padded fp16 tensor (16 bytes per pixel for 1-3 numbers) and the head's dgrad launch
// (conv_pipe_kernel<LM_RAW, ..., WG = 2>) read it back together with the last activation: 5 full-resolution tensor passes
// and two launches (19.6 + 19.7 us at ISIC) for what is, with K <= 4 outputs, an outer product per pixel.  Here one thread
// owns a pixel: BatchNorm on load, the fp32 output layer + sigmoid, mse and d(loss * scale)/d(logit) rounded to fp16 (the
// value the dgrad consumed before), dy = W . dlogit (fp16 weights like the packed dgrad operand, fp32 sums, fp16 result),
// its BatchNorm-backward statistics, and the output layer's weight / bias gradient -- z read once, dy written once.
// Per-thread fp32 accumulators, reduced per workgroup in a fixed order into the partial-row layouts the other producers use
// (bn_bwd_coef_kernel, wgf_stage1_kernel): bit-reproducible.
template <int CS, int K>
__global__ __launch_bounds__(256) void head_mse_fused_kernel(HeadCceArgs a) {
    IMK_STAMP_BEGIN(headf, 90000);
    constexpr int NA = 2 * CS + CS * K + K + 1;      // sum dy | sum dy z | dW[c][k] | db[k] | loss
    __shared__ float s_w[K * CS], s_wd[K * CS], s_b[K], s_sc[CS], s_sh[CS];
    __shared__ float s_red[4][NA];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < K * CS; i += 256) {
        const int k = i / CS, c = i - k * CS;
        const float w = (c < a.cin) ? a.w[(size_t)c * K + k] : 0.f;
        s_w[i] = w;                                  // forward: the fp32 output layer (unet.py:63)
        s_wd[i] = (float)(f16)w;                     // dgrad: the fp16 operand the packed weights hold
    }
    if (t < K) s_b[t] = a.bias[t];
    if (t < CS) { s_sc[t] = a.sc[t]; s_sh[t] = a.sh[t]; }
    const float S = a.ctl->loss_scale;
    if (blockIdx.x == 0 && t == 0) { a.stats[1] = 0.f; a.stats[2] = S; a.stats[3] = (float)a.ctl->step; }   // as head_loss_kernel
    __syncthreads();
    const float inv_n = 1.0f / ((float)a.n_pix * (float)K);
    float acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = 0.f;
    const long long stride = (long long)gridDim.x * 256;
    auto pixel = [&](const f16x8 (&zv)[CS / 8], const uint8_t (&yv)[K], long long p) {
        float zf[CS], xin[CS];
#pragma unroll
        for (int c = 0; c < CS; ++c) {
            zf[c] = (float)zv[c >> 3][c & 7];
            xin[c] = (float)imk_affine1(zv[c >> 3][c & 7], s_sc[c], s_sh[c]);
        }
        float gk[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float lg = s_b[k];
#pragma unroll
            for (int c = 0; c < CS; ++c) lg += xin[c] * s_w[k * CS + c];
            const float pk = 1.0f / (1.0f + expf(-lg)), e = pk - (float)yv[k];
            acc[2 * CS + CS * K + K] += e * e;
            gk[k] = (float)(f16)(S * 2.0f * e * inv_n * pk * (1.0f - pk));
            acc[2 * CS + CS * K + k] += gk[k];
        }
        f16x8 dv[CS / 8];
#pragma unroll
        for (int c = 0; c < CS; ++c) {
            float d = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                d += gk[k] * s_wd[k * CS + c];
                acc[2 * CS + c * K + k] += xin[c] * gk[k];
            }
            const f16 d16 = (f16)d;
            dv[c >> 3][c & 7] = d16;
            acc[c] += (float)d16;
            acc[CS + c] += (float)d16 * zf[c];
        }
#pragma unroll
        for (int q = 0; q < CS / 8; ++q) *reinterpret_cast<f16x8 *>(a.dy + p * CS + q * 8) = dv[q];
    };
    // NP pixels per iteration: all their loads are issued before the arithmetic of the first (round 5: 2 -> 4 -- with 4 waves per SIMD
    // that is 64 KB of loads in flight per CU instead of 32; the kernel moved its 68 MB at 2.9 TB/s)
    constexpr int NP = 4;
    long long p = (long long)blockIdx.x * 256 + t;
    for (; p + (NP - 1) * stride < a.n_pix; p += NP * stride) {
        f16x8 zz[NP][CS / 8];
        uint8_t yy[NP][K];
#pragma unroll
        for (int u = 0; u < NP; ++u)
#pragma unroll
            for (int q = 0; q < CS / 8; ++q) zz[u][q] = *reinterpret_cast<const f16x8 *>(a.z + (p + u * stride) * CS + q * 8);
#pragma unroll
        for (int u = 0; u < NP; ++u)
#pragma unroll
            for (int k = 0; k < K; ++k) yy[u][k] = a.y[(p + u * stride) * K + k];
#pragma unroll
        for (int u = 0; u < NP; ++u) pixel(zz[u], yy[u], p + u * stride);
    }
    for (; p < a.n_pix; p += stride) {
        f16x8 z0[CS / 8];
        uint8_t y0[K];
#pragma unroll
        for (int q = 0; q < CS / 8; ++q) z0[q] = *reinterpret_cast<const f16x8 *>(a.z + p * CS + q * 8);
#pragma unroll
        for (int k = 0; k < K; ++k) y0[k] = a.y[p * K + k];
        pixel(z0, y0, p);
    }
    // ---- per-workgroup partial rows, fixed order: lanes (butterfly), then waves 0..3 ----------------------------------------
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const float v = wave_sum<64>(acc[i]);
        if (lane == 0) s_red[wave][i] = v;
    }
    __syncthreads();
    auto tot = [&](int i) { return (s_red[0][i] + s_red[1][i]) + (s_red[2][i] + s_red[3][i]); };
    if (t < 2 * CS) a.dystat_partial[(size_t)blockIdx.x * 2 * CS + t] = tot(t);
    if (t == 0) a.loss_partial[blockIdx.x] = tot(2 * CS + CS * K + K);
    // weight-gradient partial row [tap 0 | bias][256] in wgf_stage1's layout: element e = r * 64 + l holds (ci = 4 (l >> 4) + r,
    // co = l & 15); the bias row keeps co in elements 0..15
    float *wp = a.wg_partial + (size_t)blockIdx.x * 2 * 256;
    {
        const int r = t >> 6, l = t & 63, ci = 4 * (l >> 4) + r, co = l & 15;
        wp[t] = (ci < CS && co < K) ? tot(2 * CS + ci * K + co) : 0.f;
        wp[256 + t] = (t < K) ? tot(2 * CS + CS * K + t) : 0.f;
    }
    IMK_STAMP_END(1);
}

}  // namespace

// Sigmoid heads with one map on 8 (padded) channels: ISIC at alpha 0.5, the headline shape.  Measured per training step
// (tests/gpu_probe/ab_env.sh, off / on): ISIC alpha 0.5 1.007 / 0.983 ms; the wider forms hold their 2 CS + CS K + K + 1
// accumulators in 215-256 registers (two waves per SIMD) and lose -- ISIC alpha 1 (16 channels, 1 map) 1.674 / 1.684, HeLa
// (16 channels, 3 maps) 1.673 / 1.745 -- so they keep head_loss_kernel + the pipelined dgrad.  rows_cap as below.
bool imk_head_mse_fused_ok(int cs, int K, long long n_pix, int rows_cap) {
    static const bool off = []() { const char *e = getenv("IMK_HEAD_MSE_FUSE"); return e && e[0] == '0'; }();
    if (off || cs != 8 || K != 1) return false;
    return imk_loss_blocks(n_pix) <= rows_cap;
}

int imk_launch_head_mse_fused(const f16 *z, const float *sc, const float *sh, const float *w, const float *bias, int cin, int cs,
                              int K, long long n_pix, const uint8_t *y, const ImkCtl *ctl, float *stats, f16 *dy,
                              float *loss_partial, float *dystat_partial, float *wg_partial, hipStream_t stream) {
    HeadCceArgs a{z, sc, sh, w, bias, cin, cs, K, n_pix, y, ctl, stats, dy, loss_partial, dystat_partial, wg_partial, 1};
    const int grid = imk_loss_blocks(n_pix);
    ImkProfScope prof(PF_HEAD_LOSS, (double)n_pix * (cs * 2 + K + cs * 2), stream, 6.0 * n_pix * cin * K);
    if (cs == 8 && K == 1) imk_klaunch(head_mse_fused_kernel<8, 1>, dim3(grid), dim3(256), 0, stream, a);
    else return IMK_EUNSUPPORTED;
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}

// Softmax heads with 8 ... 32 (padded) input channels and up to 64 classes.  `rows_cap`: capacity (rows) of dystat_partial.
bool imk_head_cce_fused_ok(int cs, int K, long long n_pix, int rows_cap) {
    static const bool off = []() { const char *e = getenv("IMK_HEAD_FUSE"); return e && e[0] == '0'; }();
    if (off || (cs != 8 && cs != 16 && cs != 24 && cs != 32) || K < 2 || K > 64) return false;
    return imk_loss_blocks(n_pix) <= rows_cap;
}

int imk_head_cce_fused_rows(long long n_pix) { return imk_loss_blocks(n_pix); }

int imk_launch_head_cce_fused(const f16 *z, const float *sc, const float *sh, const float *w, const float *bias, int cin, int cs,
                              int K, long long n_pix, const uint8_t *y, const ImkCtl *ctl, float *stats, f16 *dy,
                              float *loss_partial, float *dystat_partial, float *wg_partial, hipStream_t stream) {
    HeadCceArgs a{z, sc, sh, w, bias, cin, cs, K, n_pix, y, ctl, stats, dy, loss_partial, dystat_partial, wg_partial,
                  (imk_pad8(K) + 15) / 16};
    const int kt = (K + 15) / 16, csv = cs <= 16 ? 16 : 32, nct = csv / 16;
    const int grid = imk_loss_blocks(n_pix);
    const size_t img = (size_t)4 * (nct + kt) * 64 * WG_STRIDE_H * sizeof(f16);
    const size_t red = ((size_t)4 * (nct * kt + 1) * 256 + 4 * (2 * csv + 1)) * sizeof(float);
    const size_t lds = img > red ? img : red;
    ImkProfScope prof(PF_HEAD_LOSS, (double)n_pix * (cs * 2 + 1 + cs * 2), stream, 6.0 * n_pix * cin * K);
#define IMK_HF(CSV, KTV) imk_klaunch(head_cce_fused_kernel<CSV, KTV>, dim3(grid), dim3(256), lds, stream, a)
#define IMK_HF_KT(CSV) do { if (kt == 1) IMK_HF(CSV, 1); else if (kt == 2) IMK_HF(CSV, 2); else if (kt == 3) IMK_HF(CSV, 3); else IMK_HF(CSV, 4); } while (0)
    if (csv == 16) IMK_HF_KT(16); else IMK_HF_KT(32);
#undef IMK_HF_KT
#undef IMK_HF
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}
