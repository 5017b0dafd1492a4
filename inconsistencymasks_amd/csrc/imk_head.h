// The U-Net's output layer (unet.py:63: Conv2D(num_outputmasks, (1,1), activation=actifuout, dtype='float32') on the last
// decoder block's BatchNorm output), per pixel, in fp32 -- shared by head_kernel (probabilities -> HBM) and head_im_kernel
// (probabilities -> votes -> inconsistency mask, nothing stored), so that both produce the SAME fp32 probabilities bit for
// bit: the IM chain thresholds them (`>` / `>=` 0.5, functions.py:3157 / 3187) and takes their arg-max (functions.py:3225).
#pragma once
#include "imk_common.h"

// LDS image of one model's head: w^T [K][CS] (input channels beyond cin are zero), bias [K], BN scale [CS], shift [CS]
template <int CS>
__device__ __forceinline__ int head_lds_floats(int K) { return K * CS + K + 2 * CS; }

template <int CS>
__device__ __forceinline__ void head_stage(const float *__restrict__ w /*[cin][K]*/, const float *__restrict__ bias,
                                           const float *__restrict__ sc, const float *__restrict__ sh, int cin, int K,
                                           float *s_w) {
    float *s_b = s_w + K * CS, *s_sc = s_b + K, *s_sh = s_sc + CS;
    for (int i = threadIdx.x; i < K * CS; i += 256) {
        const int k = i / CS, c = i - k * CS;
        s_w[i] = (c < cin) ? w[(size_t)c * K + k] : 0.f;
    }
    for (int i = threadIdx.x; i < K; i += 256) s_b[i] = bias[i];
    for (int i = threadIdx.x; i < CS; i += 256) { s_sc[i] = sc[i]; s_sh[i] = sh[i]; }
}

// the pixel's input: fp16(z * scale + shift) per channel (the BatchNorm of the last decoder block, applied on load)
template <int CS>
__device__ __forceinline__ void head_input(const f16 *__restrict__ z, long long p, const float *s_w, int K, float (&xin)[CS]) {
    const float *s_sc = s_w + K * CS + K, *s_sh = s_sc + CS;
#pragma unroll
    for (int q = 0; q < CS / 8; ++q) {
        const f16x8 v = *reinterpret_cast<const f16x8 *>(z + p * CS + q * 8);
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const f16x2 r2 = imk_affine2(f16x2{v[j], v[j + 1]}, f32x2{s_sc[q * 8 + j], s_sc[q * 8 + j + 1]}, f32x2{s_sh[q * 8 + j], s_sh[q * 8 + j + 1]});
            xin[q * 8 + j] = (float)r2[0]; xin[q * 8 + j + 1] = (float)r2[1];
        }
    }
}

template <int CS>
__device__ __forceinline__ float head_logit(const float (&xin)[CS], const float *s_w, int K, int k) {
    float acc = s_w[K * CS + k];
#pragma unroll
    for (int c = 0; c < CS; ++c) acc += xin[c] * s_w[k * CS + c];
    return acc;
}

__device__ __forceinline__ float head_sigmoid(float logit) { return 1.0f / (1.0f + expf(-logit)); }

// softmax over the K classes into row[0..K) (an LDS row of the caller)
template <int CS>
__device__ __forceinline__ void head_softmax_row(const float (&xin)[CS], const float *s_w, int K, float *row) {
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) {
        const float acc = head_logit<CS>(xin, s_w, K, k);
        row[k] = acc;
        mx = fmaxf(mx, acc);
    }
    float sum = 0.f;
    for (int k = 0; k < K; ++k) { const float e = expf(row[k] - mx); row[k] = e; sum += e; }
    const float inv = 1.0f / sum;
    for (int k = 0; k < K; ++k) row[k] *= inv;
}

// ---- softmax heads on the matrix cores (head_softmax_kernel: probabilities -> HBM; head_im_softmax_kernel: -> votes) ------
// logits^T [class][pixel] = W^T . x^T + b on v_mfma_f32_16x16x32_f16, ONE k-step for all (<= 32) input channels, with the fp32
// weights of the reference's float32 output layer split into two fp16 halves w = hi + lo (hi = fp16(w), lo = fp16(w - hi)): the
// products with the fp16 activations are exact in fp32 and the two MFMAs carry ~22 bits of w (1e-7 relative).  A wave takes 16
// pixels per unit, lane (p16 = lane & 15, g = lane >> 4) supplies the channels 8 g + j of pixel p16 (the 16 bytes it loads
// itself) and receives the classes 16 kt + 4 g + r of the same pixel, so the softmax is registers + two shuffles over the 4
// lanes of a pixel.  History: one thread per pixel with a broadcast LDS read per product took 2.0 ms per 128 Cityscapes images
// and 2 models at alpha 2; fp32 MFMAs (16x16x4, 8 dependent steps per class tile) 1.0 ms, 40 % of it matrix-core time.
// Both kernels call probs() below and nothing else, so the stored and the voted probabilities are the same bits.
template <int KT /* 16-class tiles */>
struct HeadMfma {
    f16x8 whi[KT], wlo[KT];
    float bias_r[KT][4], sc8[8], sh8[8];

    __device__ __forceinline__ void load(const float *__restrict__ w /*[cin][K]*/, const float *__restrict__ bias,
                                         const float *__restrict__ sc, const float *__restrict__ sh, int cin, int cs, int K) {
        const int lane = threadIdx.x & 63, p16 = lane & 15, g = lane >> 4;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = 8 * g + j, k = 16 * kt + p16;
                const float v = (c < cin && k < K) ? w[(size_t)c * K + k] : 0.f;
                whi[kt][j] = (f16)v;
                wlo[kt][j] = (f16)(v - (float)whi[kt][j]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int k = 16 * kt + 4 * g + r; bias_r[kt][r] = k < K ? bias[k] : 0.f; }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int c = 8 * g + j; sc8[j] = c < cs ? sc[c] : 0.f; sh8[j] = c < cs ? sh[c] : 0.f; }
    }

    // this lane's 8 channels of pixel p (cs is a multiple of 8: they are all inside or all outside the tensor)
    __device__ __forceinline__ f16x8 load_z(const f16 *__restrict__ z, long long p, int cs) const {
        const int g = (threadIdx.x & 63) >> 4;
        f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (8 * g < cs) v = *reinterpret_cast<const f16x8 *>(z + p * cs + 8 * g);
        return v;
    }

    // softmax probabilities of the lane's pixel: pr[kt][r] = class 16 kt + 4 g + r (0 for classes >= K)
    __device__ __forceinline__ void probs(const f16x8 &z8, int K, f32x4 (&pr)[KT]) const {
        const int g = (threadIdx.x & 63) >> 4;
        f16x8 x8;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {                                                  // as head_input
            const f16x2 r2 = imk_affine2(f16x2{z8[j], z8[j + 1]}, f32x2{sc8[j], sc8[j + 1]}, f32x2{sh8[j], sh8[j + 1]});
            x8[j] = r2[0]; x8[j + 1] = r2[1];
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            pr[kt] = f32x4{bias_r[kt][0], bias_r[kt][1], bias_r[kt][2], bias_r[kt][3]};
            pr[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[kt], x8, pr[kt], 0, 0, 0);
            pr[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo[kt], x8, pr[kt], 0, 0, 0);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (16 * kt + 4 * g + r >= K) pr[kt][r] = -INFINITY;     // padding classes: exp -> 0
                mx = fmaxf(mx, pr[kt][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { pr[kt][r] = __builtin_amdgcn_exp2f((pr[kt][r] - mx) * 1.44269504088896f); sum += pr[kt][r]; }   // v_exp_f32 (1 ulp of 2^x)
        sum += __shfl_xor(sum, 16, 64);      // (s_g + s_g^1) + (s_g^2 + s_g^3): the same bits on the 4 lanes of a pixel
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) pr[kt][r] *= inv;
    }

    // arg-max over the K classes of the lane's pixel, the lowest index winning ties (numpy's argmax, functions.py:3225);
    // the same value on the 4 lanes of the pixel
    __device__ __forceinline__ int argmax(const f32x4 (&pr)[KT], int K) const {
        const int g = (threadIdx.x & 63) >> 4;
        float bv = -1.f;
        int bk = 0x7fffffff;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 16 * kt + 4 * g + r;
                if (k < K && pr[kt][r] > bv) { bv = pr[kt][r]; bk = k; }     // ascending k within a lane: strict > keeps the first
            }
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int ok = __shfl_xor(bk, o, 64);
            if (ov > bv || (ov == bv && ok < bk)) { bv = ov; bk = ok; }
        }
        return bk == 0x7fffffff ? 0 : bk;
    }
};

// ---- fused head + inconsistency mask (imk_im.hip) -------------------------------------------------------------------------
#define IMK_HEAD_IM_MAX_MODELS 8
struct ImkHeadImArgs {
    const f16 *z[IMK_HEAD_IM_MAX_MODELS];        // last decoder block's conv output [B,H,W,cs] of every model
    const float *sc[IMK_HEAD_IM_MAX_MODELS], *sh[IMK_HEAD_IM_MAX_MODELS];   // its folded BatchNorm
    const float *w[IMK_HEAD_IM_MAX_MODELS], *bias[IMK_HEAD_IM_MAX_MODELS];  // the head's fp32 kernel [cin][K] / bias [K]
    int n_models, cin, cs, K, softmax;
    int batch, hw;
    float thr;
    int cmp_ge;
    const uint8_t *img;
    int c, block_in, block_out;
    uint8_t *img_out, *masks_out, *im_out;
    int64_t *im_size, *pred_size;
    uint8_t *presence;
};
// sigmoid heads: K <= 4 (ISIC 1, HeLa 3); softmax heads: K <= 64.  Returns IMK_EUNSUPPORTED otherwise (the caller then
// runs head_kernel per model + imk_im_binary / imk_im_multiclass).
bool imk_head_im_supported(const ImkHeadImArgs &a);
int imk_launch_head_im(const ImkHeadImArgs &a, hipStream_t stream);
