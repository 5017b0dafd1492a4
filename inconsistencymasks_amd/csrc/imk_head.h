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
        for (int j = 0; j < 8; ++j) xin[q * 8 + j] = (float)(f16)((float)v[j] * s_sc[q * 8 + j] + s_sh[q * 8 + j]);
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
