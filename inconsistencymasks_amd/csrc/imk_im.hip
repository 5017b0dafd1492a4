// Fused Inconsistency-Mask kernels for gfx950 (MI355X).
//
// One launch per batch turns the N-model probability stack into pseudo-label, inconsistency mask,
// blocked image and per-image sizes -- the chain the reference runs per image in numpy:
//   threshold/argmax  functions.py:3157, 3187-3189, 3225
//   agreement         functions.py:3104-3120 (binary), 3123-3137 (multiclass), 3195-3200 (HeLa combine)
//   blocking          functions.py:2867-2874
// The kernels are pure streaming (about 3 integer ops per byte): the roofline is HBM bandwidth.
// Layout choices for that: every global access is a 16-byte vector per lane (float4 probability
// loads, uint4 mask / image stores); results are staged in LDS so that the byte-granular outputs
// (3-channel images, 1-byte masks) still leave the CU as full 16-byte stores; one workgroup never
// straddles two images so the per-image sizes need one integer atomic per workgroup and channel.
#include "imk_common.h"
#include "imk_head.h"

namespace {

constexpr int BIN_CHUNK = 1024;  // pixels per workgroup, binary kernel (4 per thread)
constexpr int MC_CHUNK = 256;    // pixels per workgroup, multiclass kernel (1 per thread)

// ---- shared phase 2: LDS results -> global, 16-byte stores when the shape allows -------------------
template <int CHUNK>
__device__ __forceinline__ void store_bytes(uint8_t *__restrict__ dst, const uint8_t *s, int n, bool vec) {
    const int t = threadIdx.x;
    if (vec) {
        for (int i = t; i < n / 16; i += 256)
            reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(s)[i];
    } else {
        for (int i = t; i < n; i += 256) dst[i] = s[i];
    }
}

__device__ __forceinline__ void block_image(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                            const uint8_t *s_im, int n_px, int c, int block_in, bool vec) {
    const int t = threadIdx.x;
    const int nbytes = n_px * c;
    if (vec) {
        for (int i = t; i < nbytes / 16; i += 256) {
            uint4 v = reinterpret_cast<const uint4 *>(src)[i];
            if (block_in) {
                uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint32_t keep = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int idx = i * 16 + q * 4 + k;
                        const int px = (c == 3) ? idx / 3 : (c == 1 ? idx : idx / c);
                        keep |= (s_im[px] ? 0u : 0xffu) << (8 * k);
                    }
                    w[q] &= keep;
                }
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
            reinterpret_cast<uint4 *>(dst)[i] = v;
        }
    } else {
        for (int i = t; i < nbytes; i += 256) {
            const int px = i / c;
            dst[i] = (block_in && s_im[px]) ? 0 : src[i];
        }
    }
}

// ---- binary / HeLa -----------------------------------------------------------------------------
// grid (ceil(HW/1024), B); requires HW % 4 == 0 and 16-byte aligned preds (host checks), else the
// generic kernel below runs.
template <int KB>
__global__ __launch_bounds__(256) void im_binary_vec(
    const float *__restrict__ preds, int n_models, int batch, int hw, float thr, int cmp_ge,
    const uint8_t *__restrict__ img, int c, int block_in, int block_out,
    uint8_t *__restrict__ img_out, uint8_t *__restrict__ masks_out, uint8_t *__restrict__ im_out,
    unsigned long long *__restrict__ im_size, unsigned long long *__restrict__ pred_size, int vec_out, int vec_img) {
    __shared__ __attribute__((aligned(16))) uint8_t s_final[KB][BIN_CHUNK];
    __shared__ __attribute__((aligned(16))) uint8_t s_im[BIN_CHUNK];
    __shared__ int s_cnt[2 * KB];
    const int b = blockIdx.y;
    const int p_base = blockIdx.x * BIN_CHUNK;
    const int t = threadIdx.x;
    const int p0 = p_base + 4 * t;
    if (t < 2 * KB) s_cnt[t] = 0;
    int cnt_fg[KB], cnt_mx[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) cnt_fg[k] = cnt_mx[k] = 0;

    if (p0 < hw) {
        int s[4][KB];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < KB; ++k) s[j][k] = 0;
        for (int n = 0; n < n_models; ++n) {
            const float4 *src = reinterpret_cast<const float4 *>(preds + ((size_t)(n * batch + b) * hw + p0) * KB);
            float v[4 * KB];
#pragma unroll
            for (int q = 0; q < KB; ++q) {
                const float4 f = src[q];
                v[4 * q] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w;
            }
#pragma unroll
            for (int i = 0; i < 4 * KB; ++i) {
                const bool hit = cmp_ge ? (v[i] >= thr) : (v[i] > thr);  // NaN compares false
                s[i / KB][i % KB] += hit ? 1 : 0;
            }
        }
        uint32_t fin[KB];
#pragma unroll
        for (int k = 0; k < KB; ++k) fin[k] = 0;
        uint32_t im4 = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bool any = false;
            bool fg[KB];
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                fg[k] = (s[j][k] == n_models);
                const bool mx = (s[j][k] != 0) && !fg[k];
                cnt_fg[k] += fg[k];
                cnt_mx[k] += mx;
                any |= mx;
            }
            if (any) im4 |= 0xffu << (8 * j);
#pragma unroll
            for (int k = 0; k < KB; ++k)
                if (fg[k] && !(block_out && any)) fin[k] |= 0xffu << (8 * j);
        }
#pragma unroll
        for (int k = 0; k < KB; ++k) *reinterpret_cast<uint32_t *>(&s_final[k][4 * t]) = fin[k];
        *reinterpret_cast<uint32_t *>(&s_im[4 * t]) = im4;
    }
    __syncthreads();
    // per-image sizes: wave reduction, one LDS atomic per wave, one global atomic per workgroup
#pragma unroll
    for (int k = 0; k < KB; ++k) {
        int a = cnt_fg[k], m = cnt_mx[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); m += __shfl_xor(m, o, 64); }
        if ((t & 63) == 0) { atomicAdd(&s_cnt[2 * k], a); atomicAdd(&s_cnt[2 * k + 1], m); }
    }
    __syncthreads();
    if (t < KB) {
        if (s_cnt[2 * t]) atomicAdd(&pred_size[(size_t)b * KB + t], (unsigned long long)s_cnt[2 * t]);
        if (s_cnt[2 * t + 1]) atomicAdd(&im_size[(size_t)b * KB + t], (unsigned long long)s_cnt[2 * t + 1]);
    }
    const int n_px = min(BIN_CHUNK, hw - p_base);
#pragma unroll
    for (int k = 0; k < KB; ++k)
        store_bytes<BIN_CHUNK>(masks_out + ((size_t)b * KB + k) * hw + p_base, s_final[k], n_px, vec_out);
    store_bytes<BIN_CHUNK>(im_out + (size_t)b * hw + p_base, s_im, n_px, vec_out);
    if (img) {
        const size_t off = ((size_t)b * hw + p_base) * c;
        block_image(img + off, img_out + off, s_im, n_px, c, block_in, vec_img);
    }
}

// Any shape / alignment: one pixel per thread.  Used for odd sizes only.
__global__ __launch_bounds__(256) void im_binary_generic(
    const float *__restrict__ preds, int n_models, int batch, int hw, int kb, float thr, int cmp_ge,
    const uint8_t *__restrict__ img, int c, int block_in, int block_out,
    uint8_t *__restrict__ img_out, uint8_t *__restrict__ masks_out, uint8_t *__restrict__ im_out,
    unsigned long long *__restrict__ im_size, unsigned long long *__restrict__ pred_size) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= hw) return;
    bool any = false;
    for (int k = 0; k < kb; ++k) {
        int s = 0;
        for (int n = 0; n < n_models; ++n) {
            const float v = preds[((size_t)(n * batch + b) * hw + p) * kb + k];
            s += cmp_ge ? (v >= thr) : (v > thr);
        }
        const bool mx = (s != 0) && (s != n_models);
        any |= mx;
        if (s == n_models) atomicAdd(&pred_size[(size_t)b * kb + k], 1ull);
        if (mx) atomicAdd(&im_size[(size_t)b * kb + k], 1ull);
        masks_out[((size_t)b * kb + k) * hw + p] = (s == n_models) ? 255 : 0;
    }
    im_out[(size_t)b * hw + p] = any ? 255 : 0;
    if (block_out && any)
        for (int k = 0; k < kb; ++k) masks_out[((size_t)b * kb + k) * hw + p] = 0;
    if (img)
        for (int ch = 0; ch < c; ++ch) {
            const size_t i = ((size_t)b * hw + p) * c + ch;
            img_out[i] = (block_in && any) ? 0 : img[i];
        }
}

// ---- multiclass ----------------------------------------------------------------------------------
// grid (ceil(HW/256), B), dynamic LDS = 256 * (K|1) floats.  Each model's [256 px][K] slab is read
// with coalesced 16-byte loads into LDS (odd row stride: conflict-free column walks), then one thread
// per pixel takes the arg-max (strict >, so the lowest index wins ties like numpy).
__global__ __launch_bounds__(256) void im_multi_kernel(
    const float *__restrict__ probs, int n_models, int batch, int hw, int k_classes, uint32_t magic_k,
    const uint8_t *__restrict__ img, int c, int block_in, int block_out,
    uint8_t *__restrict__ img_out, uint8_t *__restrict__ final_out, uint8_t *__restrict__ im_out,
    unsigned long long *__restrict__ im_size, uint8_t *__restrict__ presence, int vec_in, int vec_out, int vec_img) {
    extern __shared__ __attribute__((aligned(16))) float s_p[];
    __shared__ __attribute__((aligned(16))) uint8_t s_final[MC_CHUNK];
    __shared__ __attribute__((aligned(16))) uint8_t s_im[MC_CHUNK];
    __shared__ uint32_t s_pres[64];
    __shared__ int s_cnt;
    const int ks = k_classes | 1;
    const int b = blockIdx.y;
    const int p_base = blockIdx.x * MC_CHUNK;
    const int n_px = min(MC_CHUNK, hw - p_base);
    const int t = threadIdx.x;
    if (t == 0) s_cnt = 0;
    int label0 = 0;
    bool agree = true;
    for (int n = 0; n < n_models; ++n) {
        if (t < 64) s_pres[t] = 0;
        const float *src = probs + ((size_t)(n * batch + b) * hw + p_base) * k_classes;
        const int total = n_px * k_classes;
        if (vec_in && (total & 3) == 0) {
            for (int i = t; i < total / 4; i += 256) {
                const float4 v = reinterpret_cast<const float4 *>(src)[i];
                const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t idx = 4u * i + q;
                    const uint32_t px = __umulhi(idx, magic_k);  // idx / K, exact for idx*K < 2^32
                    s_p[px * ks + (idx - px * k_classes)] = e[q];
                }
            }
        } else {
            for (int i = t; i < total; i += 256) {
                const int px = i / k_classes;
                s_p[px * ks + (i - px * k_classes)] = src[i];
            }
        }
        __syncthreads();
        if (t < n_px) {
            const float *row = s_p + t * ks;
            int best = 0;
            float bv = row[0];
            for (int k = 1; k < k_classes; ++k) {
                const float v = row[k];
                if (v > bv) { bv = v; best = k; }
            }
            s_pres[best] = 1;
            if (n == 0) label0 = best; else agree = agree && (best == label0);
        }
        __syncthreads();
        if (presence && t < k_classes && s_pres[t]) presence[((size_t)n * batch + b) * k_classes + t] = 1;
    }
    int dis = 0;
    if (t < n_px) {
        const uint8_t im = agree ? 0 : 255;
        s_im[t] = im;
        s_final[t] = (agree && !(block_out && im)) ? (uint8_t)label0 : 0;
        dis = agree ? 0 : 1;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dis += __shfl_xor(dis, o, 64);
    if ((t & 63) == 0 && dis) atomicAdd(&s_cnt, dis);
    __syncthreads();
    if (t == 0 && s_cnt) atomicAdd(&im_size[b], (unsigned long long)s_cnt);
    store_bytes<MC_CHUNK>(final_out + (size_t)b * hw + p_base, s_final, n_px, vec_out);
    store_bytes<MC_CHUNK>(im_out + (size_t)b * hw + p_base, s_im, n_px, vec_out);
    if (img) {
        const size_t off = ((size_t)b * hw + p_base) * c;
        block_image(img + off, img_out + off, s_im, n_px, c, block_in, vec_img);
    }
}

// ---- head + IM in one pass (ensemble inference: imk_unet_forward_im) ------------------------------------------------
// The probability stack [N,B,H,W,K] fp32 never exists: every model's output layer (BatchNorm on load, fp32 1x1 conv,
// sigmoid / softmax -- the arithmetic of head_kernel, imk_head.h) is evaluated per pixel in registers / LDS and goes
// straight into the vote.  Per pixel it reads N x cs fp16 + the image and writes image + label map(s) + IM:
// ISIC 2 x 16 + 3 B in, 5 B out (SURVEY 8d's fused figure) instead of 8 + 3 in / 5 out behind 2 x (16 in, 4 out).
// A workgroup never straddles two images.  Sigmoid heads: 4 consecutive pixels per thread (1024 per workgroup), all
// N x 4 activation loads of a thread issued before the first use; softmax heads: the matrix-core arithmetic of imk_head.h
// (HeadMfma), four lanes per pixel.
template <int NF>
__device__ __forceinline__ void head_im_finish(const ImkHeadImArgs &a, const uint8_t (*s_final)[BIN_CHUNK], const uint8_t *s_im,
                                               int *s_cnt, const int *cnt_a, const int *cnt_m, bool count_fg, int b, int p_base,
                                               int n_px, int vec_out, int vec_img) {
    const int t = threadIdx.x, hw = a.hw;
    // per-image sizes: wave reduction, one LDS atomic per wave, one global atomic per workgroup
#pragma unroll
    for (int k = 0; k < NF; ++k) {
        int fa = cnt_a[k], fm = cnt_m[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { fa += __shfl_xor(fa, o, 64); fm += __shfl_xor(fm, o, 64); }
        if ((t & 63) == 0) { if (fa) atomicAdd(&s_cnt[2 * k], fa); if (fm) atomicAdd(&s_cnt[2 * k + 1], fm); }
    }
    __syncthreads();
    if (t < NF) {
        auto *ims = reinterpret_cast<unsigned long long *>(a.im_size), *pss = reinterpret_cast<unsigned long long *>(a.pred_size);
        if (count_fg && s_cnt[2 * t]) atomicAdd(&pss[(size_t)b * NF + t], (unsigned long long)s_cnt[2 * t]);
        if (s_cnt[2 * t + 1]) atomicAdd(&ims[(size_t)b * NF + t], (unsigned long long)s_cnt[2 * t + 1]);
    }
#pragma unroll
    for (int k = 0; k < NF; ++k)
        store_bytes<BIN_CHUNK>(a.masks_out + ((size_t)b * NF + k) * hw + p_base, s_final[k], n_px, vec_out);
    store_bytes<BIN_CHUNK>(a.im_out + (size_t)b * hw + p_base, s_im, n_px, vec_out);
    if (a.img) {
        const size_t off = ((size_t)b * hw + p_base) * a.c;
        block_image(a.img + off, a.img_out + off, s_im, n_px, a.c, a.block_in, vec_img);
    }
}

template <int CS, int KB, int NM /* models, compile-time (0: run-time) */>
__global__ __launch_bounds__(256) void head_im_sigmoid_kernel(ImkHeadImArgs a, int vec_out, int vec_img) {
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];   // N x head image (imk_head.h)
    __shared__ __attribute__((aligned(16))) uint8_t s_final[KB][BIN_CHUNK];
    __shared__ __attribute__((aligned(16))) uint8_t s_im[BIN_CHUNK];
    __shared__ int s_cnt[2 * KB];
    const int K = a.K, hw = a.hw;
    const int n_models = NM > 0 ? NM : a.n_models;
    const int per_model = head_lds_floats<CS>(K);
    const int b = blockIdx.y, t = threadIdx.x;
    const int p_base = blockIdx.x * BIN_CHUNK;
    const int n_px = min(BIN_CHUNK, hw - p_base);
    for (int n = 0; n < n_models; ++n) head_stage<CS>(a.w[n], a.bias[n], a.sc[n], a.sh[n], a.cin, K, s_dyn + n * per_model);
    if (t < 2 * KB) s_cnt[t] = 0;
    int cnt_a[KB], cnt_m[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) cnt_a[k] = cnt_m[k] = 0;
    const int p0 = 4 * t;                                   // hw % 16 == 0: a thread's 4 pixels are all in or all out
    const long long p = (long long)b * hw + p_base + p0;
    constexpr int NMR = NM > 0 ? NM : 1;
    f16x8 raw[NMR][4][CS / 8];
    if (NM > 0 && p0 < n_px) {                               // every activation load of this thread goes out first
#pragma unroll
        for (int n = 0; n < NMR; ++n)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < CS / 8; ++q) raw[n][j][q] = *reinterpret_cast<const f16x8 *>(a.z[n] + (p + j) * CS + q * 8);
    }
    __syncthreads();                                         // head images staged
    if (p0 < n_px) {
        int votes[4][KB];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < KB; ++k) votes[j][k] = 0;
        auto vote = [&](const float *hw_n, const f16x8 (&r)[CS / 8], int j) {
            const float *s_sc = hw_n + K * CS + K, *s_sh = s_sc + CS;
            float xin[CS];
#pragma unroll
            for (int q = 0; q < CS / 8; ++q)
#pragma unroll
                for (int e = 0; e < 8; e += 2) {                                                                                  // = head_input
                    const f16x2 r2 = imk_affine2(f16x2{r[q][e], r[q][e + 1]}, f32x2{s_sc[q * 8 + e], s_sc[q * 8 + e + 1]}, f32x2{s_sh[q * 8 + e], s_sh[q * 8 + e + 1]});
                    xin[q * 8 + e] = (float)r2[0]; xin[q * 8 + e + 1] = (float)r2[1];
                }
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                const float v = head_sigmoid(head_logit<CS>(xin, hw_n, K, k));
                votes[j][k] += (a.cmp_ge ? (v >= a.thr) : (v > a.thr)) ? 1 : 0;      // NaN compares false
            }
        };
        if constexpr (NM > 0) {
#pragma unroll
            for (int n = 0; n < NMR; ++n)
#pragma unroll
                for (int j = 0; j < 4; ++j) vote(s_dyn + n * per_model, raw[n][j], j);
        } else {
            for (int n = 0; n < n_models; ++n) {
                f16x8 r[4][CS / 8];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < CS / 8; ++q) r[j][q] = *reinterpret_cast<const f16x8 *>(a.z[n] + (p + j) * CS + q * 8);
#pragma unroll
                for (int j = 0; j < 4; ++j) vote(s_dyn + n * per_model, r[j], j);
            }
        }
        uint32_t fin[KB], im4 = 0;
#pragma unroll
        for (int k = 0; k < KB; ++k) fin[k] = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bool any = false, fg[KB];
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                fg[k] = votes[j][k] == n_models;
                const bool mx = votes[j][k] != 0 && !fg[k];
                cnt_a[k] += fg[k]; cnt_m[k] += mx;
                any |= mx;
            }
            if (any) im4 |= 0xffu << (8 * j);
#pragma unroll
            for (int k = 0; k < KB; ++k)
                if (fg[k] && !(a.block_out && any)) fin[k] |= 0xffu << (8 * j);
        }
#pragma unroll
        for (int k = 0; k < KB; ++k) *reinterpret_cast<uint32_t *>(&s_final[k][p0]) = fin[k];
        *reinterpret_cast<uint32_t *>(&s_im[p0]) = im4;
    }
    __syncthreads();
    head_im_finish<KB>(a, s_final, s_im, s_cnt, cnt_a, cnt_m, true, b, p_base, n_px, vec_out, vec_img);
}

// Softmax heads: HeadMfma (imk_head.h) gives every group of 4 lanes the K probabilities of one pixel; a wave takes 64 of
// the workgroup's 256 pixels (4 units of 16), model after model, and keeps label / agreement per unit in registers.
template <int KT>
__global__ __launch_bounds__(256) void head_im_softmax_kernel(ImkHeadImArgs a, int vec_out, int vec_img) {
    __shared__ __attribute__((aligned(16))) uint8_t s_final[1][BIN_CHUNK];   // MC_CHUNK bytes used
    __shared__ __attribute__((aligned(16))) uint8_t s_im[BIN_CHUNK];
    __shared__ uint32_t s_pres[IMK_HEAD_IM_MAX_MODELS][64];
    __shared__ int s_cnt[2];
    const int K = a.K, hw = a.hw, cs = a.cs;
    const int b = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6, p16 = lane & 15, g = lane >> 4;
    const int p_base = blockIdx.x * MC_CHUNK;
    const int n_px = min(MC_CHUNK, hw - p_base);
    for (int i = t; i < IMK_HEAD_IM_MAX_MODELS * 64; i += 256) (&s_pres[0][0])[i] = 0;
    if (t < 2) s_cnt[t] = 0;
    __syncthreads();
    int cnt_a[1] = {0}, cnt_m[1] = {0};
    int label0[4] = {0, 0, 0, 0};
    bool agree[4] = {true, true, true, true};
    for (int n = 0; n < a.n_models; ++n) {
        HeadMfma<KT> h;
        h.load(a.w[n], a.bias[n], a.sc[n], a.sh[n], a.cin, cs, K);
        f16x8 zr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = 64 * wave + 16 * u + p16;
            zr[u] = h.load_z(a.z[n], (long long)b * hw + p_base + (q < n_px ? q : 0), cs);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f32x4 pr[KT];
            h.probs(zr[u], K, pr);
            const int best = h.argmax(pr, K);
            if (g == 0 && 64 * wave + 16 * u + p16 < n_px) s_pres[n][best] = 1;
            if (n == 0) label0[u] = best; else agree[u] = agree[u] && (best == label0[u]);
        }
    }
    if (g == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = 64 * wave + 16 * u + p16;
            if (q < n_px) {
                const uint8_t im = agree[u] ? 0 : 255;
                s_im[q] = im;
                s_final[0][q] = (agree[u] && !(a.block_out && im)) ? (uint8_t)label0[u] : 0;
                cnt_m[0] += agree[u] ? 0 : 1;
            }
        }
    }
    __syncthreads();
    if (a.presence)
        for (int i = t; i < a.n_models * 64; i += 256) {
            const int n = i >> 6, k = i & 63;
            if (k < K && s_pres[n][k]) a.presence[((size_t)n * a.batch + b) * K + k] = 1;
        }
    head_im_finish<1>(a, s_final, s_im, s_cnt, cnt_a, cnt_m, false, b, p_base, n_px, vec_out, vec_img);
}

// ---- morphology + late blocking (cold path: EK = DK = 0 in every shipped config) ------------------
__global__ __launch_bounds__(256) void morph_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                    int h, int w, int ksize, int op) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const int a = ksize / 2;
    const uint8_t *s = src + (size_t)b * h * w;
    int acc = op ? 0 : 255;
    for (int dy = -a; dy < ksize - a; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= h) continue;
        for (int dx = -a; dx < ksize - a; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= w) continue;
            const int v = s[(size_t)yy * w + xx];
            acc = op ? max(acc, v) : min(acc, v);
        }
    }
    dst[(size_t)b * h * w + (size_t)y * w + x] = (uint8_t)acc;
}

__global__ __launch_bounds__(256) void block_apply_kernel(const uint8_t *__restrict__ im, uint8_t *__restrict__ img, int c,
                                                          uint8_t *__restrict__ masks, int n_masks, int hw) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= hw) return;
    if (im[(size_t)b * hw + p] == 0) return;
    if (img)
        for (int ch = 0; ch < c; ++ch) img[((size_t)b * hw + p) * c + ch] = 0;
    if (masks)
        for (int m = 0; m < n_masks; ++m) masks[((size_t)b * n_masks + m) * hw + p] = 0;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

static size_t head_im_lds(const ImkHeadImArgs &a) {
    if (a.softmax) return 0;      // weights live in registers (HeadMfma)
    return (size_t)a.n_models * (a.K * a.cs + a.K + 2 * a.cs) * sizeof(float);
}

bool imk_head_im_supported(const ImkHeadImArgs &a) {
    if (a.n_models < 1 || a.n_models > IMK_HEAD_IM_MAX_MODELS) return false;
    if (a.softmax ? a.K > 64 : a.K > 4) return false;
    if (a.cs != 8 && a.cs != 16 && a.cs != 24 && a.cs != 32) return false;
    if (a.hw % 16 != 0 || !aligned16(a.masks_out) || !aligned16(a.im_out)) return false;   // odd sizes: unfused path
    return head_im_lds(a) <= 150 * 1024;
}

int imk_launch_head_im(const ImkHeadImArgs &a, hipStream_t stream) {
    IMK_CHECK_ARG(a.n_models > 0 && a.batch > 0 && a.hw > 0 && a.K > 0);
    IMK_CHECK_ARG(a.masks_out && a.im_out && a.im_size && (a.softmax || a.pred_size));
    IMK_CHECK_ARG(!a.img || (a.img_out && a.c > 0));
    if (!imk_head_im_supported(a)) return IMK_EUNSUPPORTED;
    const int nf = a.softmax ? 1 : a.K;
    IMK_HIP(hipMemsetAsync(a.im_size, 0, sizeof(int64_t) * a.batch * nf, stream));
    if (a.pred_size) IMK_HIP(hipMemsetAsync(a.pred_size, 0, sizeof(int64_t) * a.batch * nf, stream));
    if (a.softmax && a.presence) IMK_HIP(hipMemsetAsync(a.presence, 0, (size_t)a.n_models * a.batch * a.K, stream));
    const size_t lds = head_im_lds(a);
    const int chunk = a.softmax ? MC_CHUNK : BIN_CHUNK;
    const int vec_img2 = a.img && ((int64_t)a.hw * a.c % 16 == 0) && (chunk * a.c % 16 == 0) && aligned16(a.img) && aligned16(a.img_out);
    const dim3 grid(imk_cdiv(a.hw, chunk), a.batch);
    // algorithmic bytes: N last activations + image read, image + label map(s) + IM written
    ImkProfScope prof(PF_IM, (double)a.batch * a.hw * ((double)a.n_models * a.cs * 2 + (a.img ? 2.0 * a.c : 0.0) + nf + 1), stream);
#define IMK_HIM_LAUNCH(KERN)                                                                                                \
    do {                                                                                                                    \
        if (lds > 64 * 1024)                                                                                                \
            IMK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(KERN), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        imk_klaunch(KERN, dim3(grid), dim3(256), lds, stream, a, 1, vec_img2);                                                                   \
    } while (0)
    if (a.softmax) {
        const int kt = (a.K + 15) / 16;
        switch (kt) {
            case 1: IMK_HIM_LAUNCH(head_im_softmax_kernel<1>); break;
            case 2: IMK_HIM_LAUNCH(head_im_softmax_kernel<2>); break;
            case 3: IMK_HIM_LAUNCH(head_im_softmax_kernel<3>); break;
            default: IMK_HIM_LAUNCH(head_im_softmax_kernel<4>); break;
        }
    } else {
        // the reference's ensembles have 2-4 members: 2 and 3 get all their loads hoisted (compile-time N)
#define IMK_HIM_SIG(CSV, KBV)                                                                                               \
        do {                                                                                                                \
            if (a.n_models == 2 && CSV <= 16) IMK_HIM_LAUNCH((head_im_sigmoid_kernel<CSV, KBV, 2>));                        \
            else if (a.n_models == 3 && CSV <= 8) IMK_HIM_LAUNCH((head_im_sigmoid_kernel<CSV, KBV, 3>));                    \
            else IMK_HIM_LAUNCH((head_im_sigmoid_kernel<CSV, KBV, 0>));                                                     \
        } while (0)
#define IMK_HIM_KB(CSV)                                                                                                     \
        switch (a.K) {                                                                                                      \
            case 1: IMK_HIM_SIG(CSV, 1); break;                                                                             \
            case 2: IMK_HIM_SIG(CSV, 2); break;                                                                             \
            case 3: IMK_HIM_SIG(CSV, 3); break;                                                                             \
            default: IMK_HIM_SIG(CSV, 4); break;                                                                            \
        }
        switch (a.cs) {
            case 8: IMK_HIM_KB(8); break;
            case 16: IMK_HIM_KB(16); break;
            case 24: IMK_HIM_KB(24); break;
            default: IMK_HIM_KB(32); break;
        }
#undef IMK_HIM_KB
#undef IMK_HIM_SIG
    }
#undef IMK_HIM_LAUNCH
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}

extern "C" int imk_im_binary(const float *preds, int n_models, int batch, int h, int w, int kb,
                             float thr, int cmp_ge, const uint8_t *img, int c, int block_in, int block_out,
                             uint8_t *img_out, uint8_t *masks_out, uint8_t *im_out,
                             int64_t *im_size, int64_t *pred_size, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    IMK_CHECK_ARG(preds && masks_out && im_out && im_size && pred_size);
    IMK_CHECK_ARG(n_models > 0 && batch > 0 && h > 0 && w > 0 && kb > 0);
    IMK_CHECK_ARG(!img || (img_out && c > 0));
    const int64_t hw64 = (int64_t)h * w;
    IMK_CHECK_ARG(hw64 < (1ll << 30));
    const int hw = (int)hw64;
    IMK_HIP(hipMemsetAsync(im_size, 0, sizeof(int64_t) * batch * kb, stream));
    IMK_HIP(hipMemsetAsync(pred_size, 0, sizeof(int64_t) * batch * kb, stream));
    auto *ims = reinterpret_cast<unsigned long long *>(im_size);
    auto *pss = reinterpret_cast<unsigned long long *>(pred_size);
    const bool vec_ok = (hw % 4 == 0) && aligned16(preds) && (kb == 1 || kb == 3);
    // probability stack + image read, image + masks + IM written
    ImkProfScope prof(PF_IM, (double)batch * hw * ((double)n_models * kb * 4 + (img ? 2.0 * c : 0.0) + kb + 1), stream);
    if (vec_ok) {
        const int vec_out = (hw % 16 == 0) && aligned16(masks_out) && aligned16(im_out);
        const int vec_img = img && ((int64_t)hw * c % 16 == 0) && aligned16(img) && aligned16(img_out);
        dim3 grid(imk_cdiv(hw, BIN_CHUNK), batch);
        if (kb == 1)
            imk_klaunch(im_binary_vec<1>, dim3(grid), dim3(256), 0, stream, preds, n_models, batch, hw, thr, cmp_ge, img, c, block_in, block_out,
                                                       img_out, masks_out, im_out, ims, pss, vec_out, vec_img);
        else
            imk_klaunch(im_binary_vec<3>, dim3(grid), dim3(256), 0, stream, preds, n_models, batch, hw, thr, cmp_ge, img, c, block_in, block_out,
                                                       img_out, masks_out, im_out, ims, pss, vec_out, vec_img);
    } else {
        dim3 grid(imk_cdiv(hw, 256), batch);
        imk_klaunch(im_binary_generic, dim3(grid), dim3(256), 0, stream, preds, n_models, batch, hw, kb, thr, cmp_ge, img, c, block_in, block_out,
                                                    img_out, masks_out, im_out, ims, pss);
    }
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}

extern "C" int imk_im_multiclass(const float *probs, int n_models, int batch, int h, int w, int k,
                                 const uint8_t *img, int c, int block_in, int block_out,
                                 uint8_t *img_out, uint8_t *final_out, uint8_t *im_out,
                                 int64_t *im_size, uint8_t *presence, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    IMK_CHECK_ARG(probs && final_out && im_out && im_size);
    IMK_CHECK_ARG(n_models > 0 && batch > 0 && h > 0 && w > 0 && k > 0);
    IMK_CHECK_ARG(!img || (img_out && c > 0));
    if (k > 64) return IMK_EUNSUPPORTED;
    const int64_t hw64 = (int64_t)h * w;
    IMK_CHECK_ARG(hw64 < (1ll << 30));
    const int hw = (int)hw64;
    IMK_HIP(hipMemsetAsync(im_size, 0, sizeof(int64_t) * batch, stream));
    if (presence) IMK_HIP(hipMemsetAsync(presence, 0, (size_t)n_models * batch * k, stream));
    const int vec_in = aligned16(probs) && (((int64_t)hw * k) % 4 == 0) && ((MC_CHUNK * k) % 4 == 0);
    const int vec_out = (hw % 16 == 0) && aligned16(final_out) && aligned16(im_out);
    const int vec_img = img && ((int64_t)hw * c % 16 == 0) && aligned16(img) && aligned16(img_out);
    const uint32_t magic = (uint32_t)((1ull << 32) / (uint32_t)k) + 1u;
    const size_t lds = (size_t)MC_CHUNK * (k | 1) * sizeof(float);
    dim3 grid(imk_cdiv(hw, MC_CHUNK), batch);
    ImkProfScope prof(PF_IM, (double)batch * hw * ((double)n_models * k * 4 + (img ? 2.0 * c : 0.0) + 2), stream);
    imk_klaunch(im_multi_kernel, dim3(grid), dim3(256), lds, stream, probs, n_models, batch, hw, k, magic, img, c, block_in, block_out,
                                                img_out, final_out, im_out, reinterpret_cast<unsigned long long *>(im_size),
                                                presence, vec_in, vec_out, vec_img);
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}

extern "C" int imk_morph(const uint8_t *src, uint8_t *dst, int batch, int h, int w, int ksize, int op, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    IMK_CHECK_ARG(src && dst && src != dst && batch > 0 && h > 0 && w > 0 && ksize > 0 && ksize <= 31 && (op == 0 || op == 1));
    dim3 grid(imk_cdiv(w, 64), imk_cdiv(h, 4), batch);
    imk_klaunch(morph_kernel, dim3(grid), dim3(256), 0, stream, src, dst, h, w, ksize, op);
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}

extern "C" int imk_block_apply(const uint8_t *im, uint8_t *img, int c, uint8_t *masks, int n_masks,
                               int batch, int h, int w, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    IMK_CHECK_ARG(im && batch > 0 && h > 0 && w > 0);
    IMK_CHECK_ARG(!img || c > 0);
    IMK_CHECK_ARG(!masks || n_masks > 0);
    const int hw = h * w;
    dim3 grid(imk_cdiv(hw, 256), batch);
    imk_klaunch(block_apply_kernel, dim3(grid), dim3(256), 0, stream, im, img, c, masks, n_masks, hw);
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}

extern "C" int imk_version(void) { return IMK_VERSION; }

extern "C" const char *imk_error_string(int code) {
    switch (code) {
        case IMK_OK: return "ok";
        case IMK_EINVAL: return "invalid argument";
        case IMK_EUNSUPPORTED: return "unsupported shape";
        case IMK_EWORKSPACE: return "workspace too small";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown imk error";
    }
}
