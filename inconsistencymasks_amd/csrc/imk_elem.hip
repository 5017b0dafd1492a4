// Streaming kernels around the convolutions: BatchNorm statistics / backward, output head, loss gradient,
// fused AdamW.  All are HBM-bound; every tensor access is a 16-byte (8 x fp16) vector per lane.
//
// Reference ops replaced (third-party TF/Keras/TFA, call sites): BatchNormalization unet.py:7,16,27,35,41;
// head Conv2D(dtype float32)+sigmoid/softmax unet.py:63; losses 'mse' / CategoricalCrossentropy
// (ISIC_2018/09_ISIC_2018_IM.py:110, SUIM/10_SUIM_IM.py:114); tfa AdamW functions.py:215;
// Keras LossScaleOptimizer (dynamic loss scaling under mixed_float16).
#include "imk_elem.h"
#include "imk_head.h"

IMK_STAMP_TABLE(elem)

namespace {

constexpr float BN_EPS = 1e-3f;       // Keras BatchNormalization default epsilon

// Column sums of the two halves of partial[n_part][2 * cs] for channel ch in double precision, by one 256-thread block:
// every thread's rows are requested before any is used (one memory latency), lanes combine with shuffles, the 4 waves through
// LDS (one barrier).  Fixed order: deterministic.  The result is valid in every thread.
__device__ __forceinline__ void block_sum_rows(const float *__restrict__ partial, int n_part, int cs, int ch, int t,
                                               double &o1, double &o2) {
    double s1 = 0.0, s2 = 0.0;
    for (int i0 = t; i0 < n_part; i0 += 8 * 256) {     // 16 independent loads in flight per thread
        float v1[8], v2[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = min(i0 + u * 256, n_part - 1);           // unconditional loads (no branch between them) ...
            v1[u] = partial[(size_t)i * 2 * cs + ch];
            v2[u] = partial[(size_t)i * 2 * cs + cs + ch];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double live = (i0 + u * 256 < n_part) ? 1.0 : 0.0;   // ... rows beyond the end count as zero
            s1 += live * (double)v1[u];
            s2 += live * (double)v2[u];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    __shared__ double r1[4], r2[4];
    if ((t & 63) == 0) { r1[t >> 6] = s1; r2[t >> 6] = s2; }
    __syncthreads();
    o1 = (r1[0] + r1[1]) + (r1[2] + r1[3]);
    o2 = (r2[0] + r2[1]) + (r2[2] + r2[3]);
}

// ---- BatchNorm forward statistics ---------------------------------------------------------------
// partial [n_part][2*cs] (sum | sumsq) -> scale/shift for the consumers, saved mean/invstd for backward,
// moving statistics update.  One block per channel, fixed summation order (deterministic).
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float *__restrict__ partial, int n_part, int c, int cs,
                                                          double count, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, float *__restrict__ mov_mean,
                                                          float *__restrict__ mov_var, float *__restrict__ scale,
                                                          float *__restrict__ shift, float *__restrict__ save_mean,
                                                          float *__restrict__ save_invstd, float momentum) {
    const int ch = blockIdx.x;
    const int t = threadIdx.x;
    if (ch >= c) {  // padded channels: identity-zero
        if (t == 0) { scale[ch] = 0.f; shift[ch] = 0.f; save_mean[ch] = 0.f; save_invstd[ch] = 0.f; }
        return;
    }
    IMK_STAMP_BEGIN(elem, 1);
    // the channel's parameters: requested first, so their latency runs under the partial rows' (they are only needed at the end)
    const float gam = gamma[ch], bet = beta[ch], mm = mov_mean[ch], mv = mov_var[ch];
    double s1, s2;
    block_sum_rows(partial, n_part, cs, ch, t, s1, s2);
    IMK_STAMP(1);
    if (t == 0) {
        const double mean = s1 / count;
        double var = s2 / count - mean * mean;
        if (var < 0) var = 0;
        const float invstd = (float)(1.0 / sqrt(var + (double)BN_EPS));
        const float sc = gam * invstd;
        scale[ch] = sc;
        shift[ch] = bet - (float)mean * sc;
        save_mean[ch] = (float)mean;
        save_invstd[ch] = invstd;
        mov_mean[ch] = mm * momentum + (float)mean * (1.f - momentum);
        // Keras' fused BatchNormalization feeds the moving average the Bessel-corrected batch variance (n / (n - 1); the
        // normalisation itself uses the biased one)
        const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
        mov_var[ch] = mv * momentum + (float)unbiased * (1.f - momentum);
    }
    IMK_STAMP_END(2);
}

// inference: fold the moving statistics of all BatchNorm layers (one job each) into scale | shift
__global__ void bn_fold_batched_kernel(ImkFoldJobs jobs) {
    const ImkFoldJob &jb = jobs.j[blockIdx.y];
    for (int ch = blockIdx.x * blockDim.x + threadIdx.x; ch < jb.cs; ch += gridDim.x * blockDim.x) {
        if (ch >= jb.c) { jb.scale[ch] = 0.f; jb.scale[jb.cs + ch] = 0.f; continue; }
        const float sc = jb.gamma[ch] / sqrtf(jb.var[ch] + BN_EPS);
        jb.scale[ch] = sc;
        jb.scale[jb.cs + ch] = jb.beta[ch] - jb.mean[ch] * sc;
    }
}

// ---- BatchNorm backward, pass 1: assemble dy (if it has more than one source) and reduce --------------
// Each thread owns one 8-channel chunk index for its whole grid-stride walk (grid size is a multiple of
// nc8), so the per-channel sums live in registers until one LDS reduction per block.
struct BnPrepArgs {
    int mode;             // 0 DIRECT, 1 POOL, 2 SUM2
    const f16 *g_direct;  // DIRECT: dy [B,H,W,cs]; POOL: skip-branch gradient at full res (may be null)
    const f16 *g_other;   // POOL: dP [B,H/2,W/2,cs]; SUM2: dU [B,2H,2W,cs]
    const f16 *z;         // [B,H,W,cs] the BN's input (post-ReLU conv output)
    const float *sc, *sh; // POOL: forward affine, to recompute which window element was the max
    f16 *dy_out;          // POOL / SUM2: assembled dy
    float *partial;       // [gridDim.x][2*cs]
    int B, H, W, cs;
    int go_cs;            // POOL: channel stride of g_other (>= cs: a channel slice of a wider tensor, EvalNet's concat)
};

// block reduction over the pixel slots (fixed tree => deterministic), one partial row per block
__device__ __forceinline__ void prep_block_reduce(const float (&s1)[8], const float (&s2)[8], int slot, int slots, int sh,
                                                  int c8, int nc8, int cs, float *partial) {
    __shared__ float s_r[256][17];   // [slot][chunk lane][16]
#pragma unroll
    for (int j = 0; j < 8; ++j) { s_r[threadIdx.x][j] = s1[j]; s_r[threadIdx.x][8 + j] = s2[j]; }
    __syncthreads();
    for (int o = slots >> 1; o > 0; o >>= 1) {
        if (slot < o)
#pragma unroll
            for (int j = 0; j < 16; ++j) s_r[threadIdx.x][j] += s_r[threadIdx.x + (o << sh)][j];
        __syncthreads();
    }
    if (slot == 0 && c8 < nc8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            partial[(size_t)blockIdx.x * 2 * cs + c8 * 8 + j] = s_r[threadIdx.x][j];
            partial[(size_t)blockIdx.x * 2 * cs + cs + c8 * 8 + j] = s_r[threadIdx.x][8 + j];
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void bn_bwd_prep_kernel(BnPrepArgs a) {
    IMK_STAMP_BEGIN(elem, 10 + MODE);
    // thread -> (pixel slot, chunk): 256 threads = (256 / nc8p) pixel slots x nc8p chunk lanes, nc8p = nc8 rounded up
    // to a power of two <= 256 (lanes with chunk >= nc8 idle).  No division in the hot loop.
    const int nc8 = a.cs / 8;
    int sh = 0;
    while ((1 << sh) < nc8) ++sh;
    const int nc8p = 1 << sh;
    const int c8 = threadIdx.x & (nc8p - 1);
    const int slot = threadIdx.x >> sh;
    const int slots = 256 >> sh;
    const unsigned n_pix = (unsigned)a.B * a.H * a.W;
    const unsigned pix_stride = (unsigned)gridDim.x * slots;
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
    if (c8 < nc8) {
        float sc[8], shf[8];
        if (MODE == 1)
#pragma unroll
            for (int j = 0; j < 8; ++j) { sc[j] = a.sc[c8 * 8 + j]; shf[j] = a.sh[c8 * 8 + j]; }
        constexpr int UB = 2;   // pixels in flight per thread: all loads of a batch are issued before any use
        for (unsigned pix0 = blockIdx.x * slots + slot; pix0 < n_pix; pix0 += UB * pix_stride) {
            f16x8 zz[UB], d0[UB], d1[UB], w1[UB], w2[UB], w3[UB];
            unsigned xs[UB], ys[UB];
            bool ok[UB], inwin[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const unsigned pix = pix0 + u * pix_stride;
                ok[u] = pix < n_pix;
                if (!ok[u]) continue;
                const size_t it = (size_t)pix * nc8 + c8;
                zz[u] = *reinterpret_cast<const f16x8 *>(a.z + it * 8);
                if (MODE == 0) {
                    d0[u] = *reinterpret_cast<const f16x8 *>(a.g_direct + it * 8);
                } else {
                    const unsigned x = pix % (unsigned)a.W;
                    const unsigned r = pix / (unsigned)a.W;
                    const unsigned y = r % (unsigned)a.H;
                    const unsigned b = r / (unsigned)a.H;
                    xs[u] = x; ys[u] = y;
                    if (MODE == 1) {
                        const int Hh = a.H / 2, Wh = a.W / 2;
                        // an odd last row / column belongs to no window (MaxPooling2D 'valid'): its loads are clamped
                        // into the last window and its pooled gradient dropped below (inwin)
                        const unsigned yh = min(y >> 1, (unsigned)Hh - 1), xh = min(x >> 1, (unsigned)Wh - 1);
                        inwin[u] = (y >> 1) < (unsigned)Hh && (x >> 1) < (unsigned)Wh;
                        d1[u] = *reinterpret_cast<const f16x8 *>(a.g_other + ((size_t)(b * Hh + yh) * Wh + xh) * a.go_cs + c8 * 8);
                        d0[u] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                        if (a.g_direct) d0[u] = *reinterpret_cast<const f16x8 *>(a.g_direct + it * 8);
                        // the other three elements of the 2x2 pooling window (row-major order kept below)
                        const size_t w00 = (((size_t)(b * a.H + 2 * yh) * a.W + 2 * xh) * nc8 + c8) * 8;
                        const int me = (y & 1) * 2 + (x & 1);
                        const size_t o1 = (size_t)a.cs, o2 = (size_t)a.W * a.cs;
                        const int e0 = 0 < me ? 0 : 1, e1 = 1 < me ? 1 : 2, e2 = 2 < me ? 2 : 3;   // window indices != me
                        w1[u] = *reinterpret_cast<const f16x8 *>(a.z + w00 + (e0 >> 1) * o2 + (e0 & 1) * o1);
                        w2[u] = *reinterpret_cast<const f16x8 *>(a.z + w00 + (e1 >> 1) * o2 + (e1 & 1) * o1);
                        w3[u] = *reinterpret_cast<const f16x8 *>(a.z + w00 + (e2 >> 1) * o2 + (e2 & 1) * o1);
                    } else {
                        const int W2 = a.W * 2;
                        const size_t u00 = (((size_t)(b * a.H * 2 + 2 * y) * W2 + 2 * x) * nc8 + c8) * 8;
                        d0[u] = *reinterpret_cast<const f16x8 *>(a.g_other + u00);
                        w1[u] = *reinterpret_cast<const f16x8 *>(a.g_other + u00 + (size_t)a.cs);
                        w2[u] = *reinterpret_cast<const f16x8 *>(a.g_other + u00 + (size_t)W2 * a.cs);
                        w3[u] = *reinterpret_cast<const f16x8 *>(a.g_other + u00 + (size_t)W2 * a.cs + a.cs);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                if (!ok[u]) continue;
                const unsigned pix = pix0 + u * pix_stride;
                const size_t it = (size_t)pix * nc8 + c8;
                f16x8 dy;
                if (MODE == 0) {
                    dy = d0[u];
                } else if (MODE == 1) {
                    // this pixel takes the pooled gradient iff it is the FIRST maximum of its window in row-major
                    // order: strictly greater than the elements before it, greater or equal to those after it
                    const int me = (ys[u] & 1) * 2 + (xs[u] & 1);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const f16 vm = imk_affine1(zz[u][j], sc[j], shf[j]);      // the forward's pooled values (imk_common.h)
                        const f16 o[3] = {imk_affine1(w1[u][j], sc[j], shf[j]), imk_affine1(w2[u][j], sc[j], shf[j]),
                                          imk_affine1(w3[u][j], sc[j], shf[j])};
                        bool win = true;
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const int e = k < me ? k : k + 1;           // window index of the k-th "other" element
                            win = win && (e < me ? (vm > o[k]) : (vm >= o[k]));
                        }
                        dy[j] = (f16)((float)d0[u][j] + ((win && inwin[u]) ? (float)d1[u][j] : 0.f));
                    }
                    *reinterpret_cast<f16x8 *>(a.dy_out + it * 8) = dy;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        dy[j] = (f16)(((float)d0[u][j] + (float)w1[u][j]) + ((float)w2[u][j] + (float)w3[u][j]));
                    *reinterpret_cast<f16x8 *>(a.dy_out + it * 8) = dy;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = (float)dy[j]; s1[j] += d; s2[j] += d * (float)zz[u][j]; }
            }
        }
    }
    prep_block_reduce(s1, s2, slot, slots, sh, c8, nc8, a.cs, a.partial);
    IMK_STAMP_END(1);
}

// MODE 1 (encoder outputs: skip gradient + max-pool scatter), one 2x2 pooling window per thread and iteration: the four
// z and four skip-gradient chunks of the window and its pooled gradient are 9 loads for 4 pixels (the per-pixel
// form above needs 6 per pixel: itself, the pooled gradient, the skip gradient and the three other window elements),
// the first-maximum decision is taken once per window, and the index arithmetic runs per window instead of per pixel.
// DIRECT = false: no skip gradient (EvalNet's blocks: max-pool scatter only, 5 loads per window; the pooled gradient may
// be a channel slice of a wider tensor, go_cs).
template <bool DIRECT>
__global__ __launch_bounds__(256) void bn_bwd_prep_pool_kernel(BnPrepArgs a) {
    IMK_STAMP_BEGIN(elem, 20);
    const int nc8 = a.cs / 8;
    int sh = 0;
    while ((1 << sh) < nc8) ++sh;
    const int nc8p = 1 << sh;
    const int c8 = threadIdx.x & (nc8p - 1);
    const int slot = threadIdx.x >> sh;
    const int slots = 256 >> sh;
    const unsigned Hh = a.H / 2, Wh = a.W / 2;
    const unsigned n_win = (unsigned)a.B * Hh * Wh;
    const unsigned stride = (unsigned)gridDim.x * slots;
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
    if (c8 < nc8) {
        float sc[8], shf[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { sc[j] = a.sc[c8 * 8 + j]; shf[j] = a.sh[c8 * 8 + j]; }
        const size_t o1 = (size_t)a.cs, o2 = (size_t)a.W * a.cs;
        // Software-pipelined over the thread's windows (round 5): the 9 loads of window w + stride are issued before window w is
        // computed and stored, so a wave keeps two windows of loads in flight -- alone on the chip the kernel moved its 3.25 MB per
        // image at 2.6 TB/s with 2 waves per SIMD and one window each.  The loads are unconditional (a thread past its last window
        // re-reads that window: a cache hit), so the wait counts are compile-time constants.
        struct Win { f16x8 z[4], g[4], dp; size_t w00; };
        auto load = [&](unsigned w, Win &q) {
            const unsigned xh = w % Wh, r = w / Wh, yh = r % Hh, b = r / Hh;
            q.w00 = (((size_t)(b * a.H + 2 * yh) * a.W + 2 * xh) * nc8 + c8) * 8;
            const size_t off[4] = {q.w00, q.w00 + o1, q.w00 + o2, q.w00 + o2 + o1};   // row-major window order
#pragma unroll
            for (int e = 0; e < 4; ++e) q.z[e] = *reinterpret_cast<const f16x8 *>(a.z + off[e]);
            q.dp = *reinterpret_cast<const f16x8 *>(a.g_other + (size_t)w * a.go_cs + c8 * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) q.g[e] = DIRECT ? *reinterpret_cast<const f16x8 *>(a.g_direct + off[e]) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        };
        auto compute = [&](const Win &q) {
            const size_t off[4] = {q.w00, q.w00 + o1, q.w00 + o2, q.w00 + o2 + o1};
            f16x8 dy[4];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // the pooled gradient goes to the FIRST maximum of the window in row-major order
                f16 v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = imk_affine1(q.z[e][j], sc[j], shf[j]);
                int win = 0;
#pragma unroll
                for (int e = 1; e < 4; ++e) if (v[e] > v[win]) win = e;
#pragma unroll
                for (int e = 0; e < 4; ++e) dy[e][j] = (f16)((float)q.g[e][j] + (e == win ? (float)q.dp[j] : 0.f));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                *reinterpret_cast<f16x8 *>(a.dy_out + off[e]) = dy[e];
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = (float)dy[e][j]; s1[j] += d; s2[j] += d * (float)q.z[e][j]; }
            }
        };
        unsigned w = blockIdx.x * slots + slot;
        if (w < n_win) {
            Win cur, nxt;
            load(w, cur);
            while (true) {
                const unsigned wn = w + stride;
                const bool more = wn < n_win;
                load(more ? wn : w, nxt);
                compute(cur);
                if (!more) break;
                cur = nxt;
                w = wn;
            }
        }
    }
    prep_block_reduce(s1, s2, slot, slots, sh, c8, nc8, a.cs, a.partial);
    IMK_STAMP_END(1);
}

// pass 2: per-channel coefficients  dz = A*dy + Bc*z + Cc  and the gamma/beta gradients
__global__ __launch_bounds__(256) void bn_bwd_coef_kernel(const float *__restrict__ partial, int n_part, int c, int cs,
                                                          double count, const float *__restrict__ gamma,
                                                          const float *__restrict__ save_mean,
                                                          const float *__restrict__ save_invstd,
                                                          const float *__restrict__ inv_scale_ptr,
                                                          float *__restrict__ coef /*[3][cs]*/, float *__restrict__ dgamma,
                                                          float *__restrict__ dbeta, float *__restrict__ found_inf) {
    const int ch = blockIdx.x, t = threadIdx.x;
    if (ch >= c) {
        if (t == 0) { coef[ch] = 0.f; coef[cs + ch] = 0.f; coef[2 * cs + ch] = 0.f; }
        return;
    }
    IMK_STAMP_BEGIN(elem, 2);
    // requested first: only needed after the reduction
    const double mean = save_mean[ch], invstd = save_invstd[ch], gam = gamma[ch];
    const float inv = *inv_scale_ptr;
    double S1, S2;
    block_sum_rows(partial, n_part, cs, ch, t, S1, S2);
    IMK_STAMP(1);
    if (t == 0) {
        const double dg = (S2 - mean * S1) * invstd;   // sum dy * zhat
        const double A = gam * invstd;
        coef[ch] = (float)A;
        coef[cs + ch] = (float)(-A * invstd * dg / count);
        coef[2 * cs + ch] = (float)(-A * S1 / count + A * mean * invstd * dg / count);
        const float g1 = (float)dg * inv, g2 = (float)S1 * inv;
        if (!isfinite(g1) || !isfinite(g2)) *found_inf = 1.0f;
        dgamma[ch] = g1;
        dbeta[ch] = g2;
    }
    IMK_STAMP_END(2);
}

// ---- output head: BN on load -> 1x1 conv in fp32 -> sigmoid / softmax --------------------------------
// One thread per pixel computes its K outputs into an LDS row (odd pitch: conflict-free), weights broadcast from LDS;
// the block's 256 x K probabilities are contiguous in the [pixel][K] output, so they leave as one coalesced stream
// (a thread writing its own K floats, 4*K bytes apart from its neighbour's, ran at 1/50 of that for K = 35).  K <= 64.
template <int CS>
__global__ __launch_bounds__(256) void head_kernel(const f16 *__restrict__ z, const float *__restrict__ sc,
                                                   const float *__restrict__ sh, const float *__restrict__ w /*[cin][K]*/,
                                                   const float *__restrict__ bias, int cin, int K, int softmax,
                                                   long long n_pix, float *__restrict__ probs) {
    extern __shared__ float s_w[];  // [K][CS] transposed, then bias[K], sc[CS], sh[CS] (imk_head.h), then out[256][K | 1]
    float *s_out = s_w + head_lds_floats<CS>(K);
    const int pitch = K | 1;
    head_stage<CS>(w, bias, sc, sh, cin, K, s_w);
    __syncthreads();
    const long long p0 = (long long)blockIdx.x * 256;
    const long long p = p0 + threadIdx.x;
    float *row = s_out + threadIdx.x * pitch;
    if (p < n_pix) {
        float xin[CS];
        head_input<CS>(z, p, s_w, K, xin);
        if (!softmax) {
            for (int k = 0; k < K; ++k) row[k] = head_sigmoid(head_logit<CS>(xin, s_w, K, k));
        } else {
            head_softmax_row<CS>(xin, s_w, K, row);
        }
    }
    __syncthreads();
    const long long n_here = (n_pix - p0 < 256 ? n_pix - p0 : 256) * K;   // floats this block owns, contiguous
    float *dst = probs + p0 * K;
    for (long long i = threadIdx.x; i < n_here; i += 256) {
        const int px = (int)(i / K), k = (int)(i - (long long)px * K);
        dst[i] = s_out[px * pitch + k];
    }
}

// Same arithmetic as head_kernel followed by loss_grad_kernel (same expressions in the same order, so the values are
// bit-identical), without the [n_pix, K] fp32 probability tensor in between: the logits are recomputed per pass
// (K * CS FMAs) instead of being parked in HBM.
template <int CS>
__global__ __launch_bounds__(256) void head_loss_kernel(const f16 *__restrict__ z, const float *__restrict__ sc,
                                                        const float *__restrict__ sh, const float *__restrict__ w,
                                                        const float *__restrict__ bias, int cin, int K, int softmax,
                                                        long long n_pix, const uint8_t *__restrict__ y,
                                                        const ImkCtl *__restrict__ ctl, float *__restrict__ stats,
                                                        int cs_out, f16 *__restrict__ dlogit,
                                                        float *__restrict__ loss_partial) {
    extern __shared__ float s_w[];  // [K][CS] transposed, then bias[K], sc[CS], sh[CS], then (softmax) logits[256][K | 1]
    float *s_b = s_w + K * CS, *s_sc = s_b + K, *s_sh = s_sc + CS, *s_rows = s_sh + CS;
    for (int i = threadIdx.x; i < K * CS; i += 256) {
        const int k = i / CS, c = i - k * CS;
        s_w[i] = (c < cin) ? w[(size_t)c * K + k] : 0.f;
    }
    for (int i = threadIdx.x; i < K; i += 256) s_b[i] = bias[i];
    for (int i = threadIdx.x; i < CS; i += 256) { s_sc[i] = sc[i]; s_sh[i] = sh[i]; }
    __syncthreads();
    const float S = ctl->loss_scale;
    // start of the step's gradient part: clear the overflow flag stats[1] (every kernel that can set it runs after this
    // one, the optimizer step that consumed the previous value ran before it) and note the scale / step in use
    if (blockIdx.x == 0 && threadIdx.x == 0) { stats[1] = 0.f; stats[2] = S; stats[3] = (float)ctl->step; }
    float l = 0.f;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < n_pix; p += (long long)gridDim.x * 256) {
        float xin[CS];
#pragma unroll
        for (int q = 0; q < CS / 8; ++q) {
            const f16x8 v = *reinterpret_cast<const f16x8 *>(z + p * CS + q * 8);
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const f16x2 r2 = imk_affine2(f16x2{v[j], v[j + 1]}, f32x2{s_sc[q * 8 + j], s_sc[q * 8 + j + 1]}, f32x2{s_sh[q * 8 + j], s_sh[q * 8 + j + 1]});
                xin[q * 8 + j] = (float)r2[0]; xin[q * 8 + j + 1] = (float)r2[1];
            }
        }
        auto logit = [&](int k) {
            float acc = s_b[k];
#pragma unroll
            for (int c = 0; c < CS; ++c) acc += xin[c] * s_w[k * CS + c];
            return acc;
        };
        f16 *d = dlogit + p * cs_out;
        if (!softmax) {
            const float inv_n = 1.0f / ((float)n_pix * (float)K);
            for (int k0 = 0; k0 < cs_out; k0 += 8) {
                f16x8 g8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = k0 + j;
                    if (k < K) {
                        const float pk = 1.0f / (1.0f + expf(-logit(k))), t = (float)y[p * K + k];
                        const float e = pk - t;
                        l += e * e;
                        g8[j] = (f16)(S * 2.0f * e * inv_n * pk * (1.0f - pk));
                    }
                }
                *reinterpret_cast<f16x8 *>(d + k0) = g8;
            }
        } else {
            // the K logits of this pixel are computed once and parked in an LDS row (odd pitch: conflict-free)
            float *row = s_rows + threadIdx.x * (K | 1);
            float mx = -INFINITY;
            for (int k = 0; k < K; ++k) { const float lg = logit(k); row[k] = lg; mx = fmaxf(mx, lg); }
            float sum = 0.f;
            for (int k = 0; k < K; ++k) { const float e = expf(row[k] - mx); row[k] = e; sum += e; }
            const float inv = 1.0f / sum;
            const int t = y[p];
            const float inv_n = 1.0f / (float)n_pix;
            for (int k0 = 0; k0 < cs_out; k0 += 8) {
                f16x8 g8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = k0 + j;
                    if (k < K) {
                        const float pk = row[k] * inv;
                        if (k == t) l += -logf(fmaxf(pk, 1.17549435e-38f));
                        g8[j] = (f16)(S * (pk - (k == t ? 1.0f : 0.0f)) * inv_n);
                    }
                }
                *reinterpret_cast<f16x8 *>(d + k0) = g8;
            }
        }
    }
    __shared__ float s_l[4];
    l = wave_sum<64>(l);
    if ((threadIdx.x & 63) == 0) s_l[threadIdx.x >> 6] = l;
    __syncthreads();
    if (threadIdx.x == 0) loss_partial[blockIdx.x] = (s_l[0] + s_l[1]) + (s_l[2] + s_l[3]);
}

// ---- loss (head_loss_kernel above) ------------------------------------------------------------------------------
// mse:  L = mean_{all elements}(p - t)^2, dlogit = S * 2 (p - t) / Ntot * p (1 - p)      (sigmoid head)
// cce:  L = mean_{pixels} -log(p_t), dlogit_k = S * (p_k - [k == t]) / Npix   (softmax head; Keras takes the
//       softmax ACTIVATION's cached logits, so there is no probability clipping in loss or gradient; p_t is only
//       floored at FLT_MIN so that an fp32 underflow reports 87.3 instead of inf)
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float *__restrict__ partial, int n, double denom,
                                                            float *__restrict__ stats) {
    __shared__ double r[256];
    double s = 0;
    for (int i0 = threadIdx.x; i0 < n; i0 += 8 * 256) {     // 8 independent loads in flight per thread; same summation order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (i0 + u * 256 < n) ? partial[i0 + u * 256] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    r[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) r[threadIdx.x] += r[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) stats[0] = (float)(r[0] / denom);   // stats[1..3]: head_loss_kernel and the gradient kernels
}

// ---- optimizer --------------------------------------------------------------------------------------
__global__ void ctl_init_kernel(ImkCtl *ctl) {
    ctl->loss_scale = 32768.0f;  // Keras LossScaleOptimizer initial_scale = 2^15
    ctl->inv_loss_scale = 1.0f / 32768.0f;
    ctl->good_steps = 0;
    ctl->step = 0;
    ctl->found_inf = 0.f;
}

// tfa AdamW: var -= wd*var; m,v update; var -= lr_t * m / (sqrt(v) + eps), lr_t = lr*sqrt(1-b2^t)/(1-b1^t)
__global__ __launch_bounds__(256) void adamw_kernel(float *__restrict__ p, float *__restrict__ m, float *__restrict__ v,
                                                    const float *__restrict__ g, long long n, const ImkCtl *__restrict__ ctl,
                                                    const float *__restrict__ stats, float grad_scale, float lr, float wd,
                                                    float b1, float b2, float eps) {
    if (stats[1] != 0.f) return;  // non-finite gradients somewhere: skip the step (dynamic loss scaling)
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float t = (float)(ctl->step + 1);
    const float lr_t = lr * sqrtf(1.0f - powf(b2, t)) / (1.0f - powf(b1, t));
    const float gi = g[i] * grad_scale;
    float pi = p[i];
    pi -= wd * pi;
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    pi -= lr_t * mi / (sqrtf(vi) + eps);
    p[i] = pi; m[i] = mi; v[i] = vi;
}

// ---- EvalNet (evalnet.py:24-47) ---------------------------------------------------------------------------------
// concatenate([towerA, towerB]) of the two towers' BatchNorm + MaxPooling2D outputs (evalnet.py:17-19, 36): one thread
// per (pooled pixel, 8-channel chunk of the concatenated tensor)
__global__ __launch_bounds__(256) void concat_pool_kernel(const f16 *__restrict__ za, const float *__restrict__ sca,
                                                          const float *__restrict__ sha, int csa,
                                                          const f16 *__restrict__ zb, const float *__restrict__ scb,
                                                          const float *__restrict__ shb, int csb, int B, int Hh, int Wh,
                                                          f16 *__restrict__ cat) {
    const int nc8 = (csa + csb) / 8;
    const long long n = (long long)B * Hh * Wh * nc8;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c8 = (int)(i % nc8);
    const long long pix = i / nc8;
    const int x = (int)(pix % Wh);
    const long long r = pix / Wh;
    const int y = (int)(r % Hh), b = (int)(r / Hh);
    const bool second = c8 * 8 >= csa;
    const f16 *z = second ? zb : za;
    const int cs = second ? csb : csa, c0 = second ? c8 * 8 - csa : c8 * 8;
    const float *sc = (second ? scb : sca) + c0, *sh = (second ? shb : sha) + c0;
    const int W = 2 * Wh;
    const f16 *p = z + ((size_t)(b * 2 * Hh + 2 * y) * W + 2 * x) * cs + c0;
    const f16x8 v0 = *reinterpret_cast<const f16x8 *>(p), v1 = *reinterpret_cast<const f16x8 *>(p + cs);
    const f16x8 v2 = *reinterpret_cast<const f16x8 *>(p + (size_t)W * cs), v3 = *reinterpret_cast<const f16x8 *>(p + (size_t)W * cs + cs);
    const f16x8 o = imk_affine_pool8(v0, v1, v2, v3, sc, sh);   // the LM_POOL load of the conv kernels
    *reinterpret_cast<f16x8 *>(cat + (size_t)pix * (csa + csb) + c8 * 8) = o;
}

// class ids [n_pix] u8 -> fp16 one-hot [n_pix][cs] (what the reference's generator builds on the host, functions.py:4978)
__global__ __launch_bounds__(256) void onehot_kernel(const uint8_t *__restrict__ cls, long long n_pix, int cs,
                                                     f16 *__restrict__ out) {
    const int nc8 = cs / 8;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pix * nc8) return;
    const long long pix = i / nc8;
    const int c0 = (int)(i - pix * nc8) * 8, k = cls[pix];
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (f16)(c0 + j == k ? 1.0f : 0.0f);
    *reinterpret_cast<f16x8 *>(out + pix * cs + c0) = o;
}

// The tail of EvalNet for one sample per workgroup: BatchNorm + MaxPooling2D of the last block (on load), GlobalAvgPool2D,
// the Dense head(s) with sigmoid (evalnet.py:43-45, 69-71), and in training the losses (head 0: mean squared error, head 1:
// binary cross-entropy -- functions.py:4708), d(loss * scale)/d(pooled tensor) and the per-sample Dense gradients.
struct EvalHeadArgs {
    const f16 *z;            // [B,H,W,cs] output of the last 1x1 conv (pre-BN)
    const float *sc, *sh;    // [cs] BatchNorm scale / shift
    const float *w[2], *bias[2];   // Dense kernels [C,K] and biases [K]
    int n_heads, K, C, cs, B, H, W;
    float *out;              // [B, n_heads*K] sigmoid outputs
    const float *y;          // training: [B, n_heads*K] targets, else null
    const ImkCtl *ctl;
    float *stats;
    f16 *dP;                 // [B,H/2,W/2,cs]
    float *partial;          // [B][n_heads*K*(C+1) + 2]: per sample dW (o-major), db, loss terms of the two heads
};

__global__ __launch_bounds__(256) void evalnet_head_kernel(EvalHeadArgs a) {
    extern __shared__ float s_f[];            // feat[cs], then logits / dlogits [n_heads*K]
    float *s_o = s_f + a.cs;
    const int b = blockIdx.x, t = threadIdx.x;
    const int Hh = a.H / 2, Wh = a.W / 2, NO = a.n_heads * a.K;
    const bool train = a.y != nullptr;
    if (train && b == 0 && t == 0) { a.stats[1] = 0.f; a.stats[2] = a.ctl->loss_scale; a.stats[3] = (float)a.ctl->step; }
    // features: mean over the pooled map of max(2x2 window of fp16(z*sc + sh))
    for (int c = t; c < a.cs; c += 256) {
        float acc = 0.f;
        if (c < a.C) {
            const float sc = a.sc[c], sh = a.sh[c];
            for (int y = 0; y < Hh; ++y)
                for (int x = 0; x < Wh; ++x) {
                    const f16 *p = a.z + ((size_t)(b * a.H + 2 * y) * a.W + 2 * x) * a.cs + c;
                    const float v0 = (float)imk_affine1(p[0], sc, sh), v1 = (float)imk_affine1(p[a.cs], sc, sh);
                    const float v2 = (float)imk_affine1(p[(size_t)a.W * a.cs], sc, sh), v3 = (float)imk_affine1(p[(size_t)a.W * a.cs + a.cs], sc, sh);
                    acc += fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
                }
            acc /= (float)(Hh * Wh);
        }
        s_f[c] = acc;
    }
    __syncthreads();
    // Dense: one wave per output unit
    for (int o = t >> 6; o < NO; o += 4) {
        const int h = o / a.K, k = o - h * a.K;
        float acc = 0.f;
        for (int c = t & 63; c < a.C; c += 64) acc += s_f[c] * a.w[h][(size_t)c * a.K + k];
        acc = wave_sum<64>(acc);
        if ((t & 63) == 0) s_o[o] = acc + a.bias[h][k];
    }
    __syncthreads();
    float loss_h[2] = {0.f, 0.f};
    if (t < NO) {
        const float lg = s_o[t], p = 1.0f / (1.0f + expf(-lg));
        a.out[(size_t)b * NO + t] = p;
        if (train) {
            const int h = t / a.K;
            const float yv = a.y[(size_t)b * NO + t], S = a.ctl->loss_scale, inv_n = 1.0f / ((float)a.B * (float)a.K);
            float g;
            if (h == 0) {   // mean squared error
                const float e = p - yv;
                loss_h[0] = e * e;
                g = 2.0f * e * p * (1.0f - p);
            } else {        // binary cross-entropy on the sigmoid's logit (Keras uses the activation's cached logits)
                loss_h[1] = fmaxf(lg, 0.f) - lg * yv + log1pf(expf(-fabsf(lg)));
                g = p - yv;
            }
            s_o[t] = S * g * inv_n;
        }
    }
    if (!train) return;
    __syncthreads();
    float *part = a.partial + (size_t)b * ((size_t)NO * (a.C + 1) + 2);
    __shared__ float s_l[2][2];   // loss terms of this sample: NO <= 128 outputs sit in the first two waves
    if (t < 128) {
        const float l0 = wave_sum<64>(loss_h[0]), l1 = wave_sum<64>(loss_h[1]);
        if ((t & 63) == 0) { s_l[t >> 6][0] = l0; s_l[t >> 6][1] = l1; }
    }
    __syncthreads();
    if (t == 0) { part[(size_t)NO * (a.C + 1)] = s_l[0][0] + s_l[1][0]; part[(size_t)NO * (a.C + 1) + 1] = s_l[0][1] + s_l[1][1]; }
    // per-sample Dense gradients dW[o][c] = dlogit[o] * feat[c], db[o] = dlogit[o]
    for (int i = t; i < NO * a.C; i += 256) { const int o = i / a.C, c = i - o * a.C; part[i] = s_o[o] * s_f[c]; }
    if (t < NO) part[(size_t)NO * a.C + t] = s_o[t];
    // gradient w.r.t. the pooled map: every window gets dfeat / (Hh*Wh)
    for (int c = t; c < a.cs; c += 256) {
        float g = 0.f;
        if (c < a.C)
            for (int o = 0; o < NO; ++o) { const int h = o / a.K, k = o - h * a.K; g += s_o[o] * a.w[h][(size_t)c * a.K + k]; }
        const f16 gv = (f16)(g / (float)(Hh * Wh));
        for (int q = 0; q < Hh * Wh; ++q) a.dP[((size_t)b * Hh * Wh + q) * a.cs + c] = gv;
    }
}

// sum the per-sample Dense gradients over the batch (fixed order), unscale, flag non-finite values; block 0 also
// reduces the loss terms: stats[0] = total, stats[4] = head 0 (mse), stats[5] = head 1 (bce)
__global__ __launch_bounds__(256) void evalnet_head_reduce_kernel(const float *__restrict__ partial, int B, int n_heads, int K,
                                                                 int C, const float *__restrict__ inv_scale_ptr,
                                                                 float *__restrict__ dw0, float *__restrict__ db0,
                                                                 float *__restrict__ dw1, float *__restrict__ db1,
                                                                 float *__restrict__ found_inf, float *__restrict__ stats) {
    const int NO = n_heads * K, per = NO * (C + 1) + 2;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < NO * (C + 1)) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += partial[(size_t)b * per + i];
        s *= *inv_scale_ptr;
        if (!isfinite(s)) *found_inf = 1.0f;
        if (i < NO * C) {
            const int o = i / C, c = i - o * C, h = o / K, k = o - h * K;
            (h ? dw1 : dw0)[(size_t)c * K + k] = s;
        } else {
            const int o = i - NO * C, h = o / K, k = o - h * K;
            (h ? db1 : db0)[k] = s;
        }
    }
    if (i == 0) {
        double l0 = 0, l1 = 0;
        for (int b = 0; b < B; ++b) { l0 += partial[(size_t)b * per + per - 2]; l1 += partial[(size_t)b * per + per - 1]; }
        const float m0 = (float)(l0 / ((double)B * K)), m1 = (float)(l1 / ((double)B * K));
        stats[0] = m0 + m1; stats[4] = m0; stats[5] = m1;
    }
}

}  // namespace

// -----------------------------------------------------------------------------------------------------
int imk_launch_bn_finalize(const float *partial, int n_part, int c, int cs, double count, const float *gamma,
                           const float *beta, float *mov_mean, float *mov_var, float *scale, float *shift,
                           float *save_mean, float *save_invstd, hipStream_t stream, float momentum) {
    ImkProfScope prof(PF_BN_FINALIZE, (double)n_part * 2 * cs * 4 + 8.0 * cs * 4, stream);
    imk_klaunch(bn_finalize_kernel, dim3(cs), dim3(256), 0, stream, partial, n_part, c, cs, count, gamma, beta, mov_mean, mov_var, scale, shift,
                                               save_mean, save_invstd, momentum);
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}

int imk_launch_bn_fold_jobs(const ImkFoldJobs &jobs, hipStream_t stream) {
    if (jobs.n <= 0) return IMK_OK;
    ImkProfScope prof(PF_STEP_TAIL, 0.0, stream);
    imk_klaunch(bn_fold_batched_kernel, dim3(dim3(2, jobs.n)), dim3(256), 0, stream, jobs);
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}

int imk_bn_prep_blocks(int B, int H, int W, int cs) {
    // Small tensors are latency-bound: one batch (2 pixels) per thread and as many blocks as that gives.
    // Large ones: at most 4096 blocks (= 2 rounds of full occupancy), more pixels per thread.
    const int nc8 = cs / 8;
    int nc8p = 1;
    while (nc8p < nc8) nc8p <<= 1;
    const int slots = 256 / nc8p;                      // pixels per block per sweep
    const long long pix = (long long)B * H * W;
    long long nb = (pix + (long long)slots * 2 - 1) / ((long long)slots * 2);
    // (cap measured on the training step, ISIC / SUIM / HeLa / Cityscapes shapes: 4096 blocks -- one window or pixel pair per
    //  thread, every block resident and in the same phase at the same time -- 1.191 / 2.260 / 2.237 / 3.382 ms; 1024: 1.170;
    //  512: 1.167 / 2.230 / 2.204 / 3.344; 256: 1.168)
    static const int cap = []() { const char *e = getenv("IMK_PREP_MAX_BLOCKS"); return e ? atoi(e) : 512; }();
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return (int)nb;
}

int imk_launch_bn_bwd_prep(int mode, const f16 *g_direct, const f16 *g_other, const f16 *z, const float *sc,
                           const float *sh, f16 *dy_out, float *partial, int B, int H, int W, int cs, hipStream_t stream,
                           int g_other_cs) {
    BnPrepArgs a{mode, g_direct, g_other, z, sc, sh, dy_out, partial, B, H, W, cs, g_other_cs > 0 ? g_other_cs : cs};
    if (a.go_cs != cs && (mode != 1 || g_direct)) return IMK_EUNSUPPORTED;
    const int nb = imk_bn_prep_blocks(B, H, W, cs);
    // mode 0: dy and z read;  mode 1: skip gradient (optional), pooled gradient at quarter size, z read, dy written;
    // mode 2: the 2H x 2W upsampled-branch gradient and z read, dy written;  + the partial rows
    const double px = (double)B * H * W, t = px * cs * 2;
    const double prep_bytes = (mode == 0 ? 2 * t : mode == 1 ? (g_direct ? 3 * t : 2 * t) + 0.25 * px * a.go_cs * 2 : 6 * t)
                            + (double)nb * 2 * cs * 4;
    ImkProfScope prof(PF_BN_PREP, prep_bytes, stream);
    if (mode == 0) imk_klaunch(bn_bwd_prep_kernel<0>, dim3(nb), dim3(256), 0, stream, a);
    else if (mode == 1 && H % 2 == 0 && W % 2 == 0 && g_direct) imk_klaunch(bn_bwd_prep_pool_kernel<true>, dim3(nb), dim3(256), 0, stream, a);
    else if (mode == 1 && H % 2 == 0 && W % 2 == 0) imk_klaunch(bn_bwd_prep_pool_kernel<false>, dim3(nb), dim3(256), 0, stream, a);
    else if (mode == 1) imk_klaunch(bn_bwd_prep_kernel<1>, dim3(nb), dim3(256), 0, stream, a);
    else imk_klaunch(bn_bwd_prep_kernel<2>, dim3(nb), dim3(256), 0, stream, a);
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}

int imk_launch_bn_bwd_coef(const float *partial, int n_part, int c, int cs, double count, const float *gamma,
                           const float *save_mean, const float *save_invstd, const float *inv_scale_ptr, float *coef,
                           float *dgamma, float *dbeta, float *found_inf, hipStream_t stream) {
    ImkProfScope prof(PF_BN_COEF, (double)n_part * 2 * cs * 4 + 8.0 * cs * 4, stream);
    imk_klaunch(bn_bwd_coef_kernel, dim3(cs), dim3(256), 0, stream, partial, n_part, c, cs, count, gamma, save_mean, save_invstd, inv_scale_ptr,
                                               coef, dgamma, dbeta, found_inf);
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}


// Softmax heads: the shared matrix-core arithmetic of imk_head.h (HeadMfma).  Persistent over groups of 64 pixels per wave
// (4 units of 16, all activation loads of a group issued first); a wave's 64 x K probabilities are contiguous in the
// [pixel][K] output, so they go through a wave-private LDS slab and leave as 16-byte stores.
template <int KT>
__global__ __launch_bounds__(256) void head_softmax_kernel(const f16 *__restrict__ z, const float *__restrict__ sc,
                                                           const float *__restrict__ sh, const float *__restrict__ w,
                                                           const float *__restrict__ bias, int cin, int cs, int K,
                                                           long long n_pix, float *__restrict__ probs, int vec) {
    extern __shared__ __attribute__((aligned(16))) float s_slab[];     // [4 waves][64][K]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p16 = lane & 15, g = lane >> 4;
    float *so = s_slab + (size_t)wave * 64 * K;
    HeadMfma<KT> h;
    h.load(w, bias, sc, sh, cin, cs, K);
    const long long n_grp = (n_pix + 63) / 64;
    for (long long grp = (long long)blockIdx.x * 4 + wave; grp < n_grp; grp += (long long)gridDim.x * 4) {
        f16x8 zr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long px = grp * 64 + u * 16 + p16;
            zr[u] = h.load_z(z, px < n_pix ? px : n_pix - 1, cs);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f32x4 pr[KT];
            h.probs(zr[u], K, pr);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 16 * kt + 4 * g + r;
                    if (k < K) so[(u * 16 + p16) * K + k] = pr[kt][r];
                }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        const long long left = n_pix - grp * 64;
        const int n_here = (int)(left < 64 ? left : 64) * K;           // floats this wave owns, contiguous
        float *dst = probs + grp * 64 * K;
        if (vec && (n_here & 3) == 0) {
            for (int i = lane; i < n_here / 4; i += 64) reinterpret_cast<float4 *>(dst)[i] = reinterpret_cast<const float4 *>(so)[i];
        } else {
            for (int i = lane; i < n_here; i += 64) dst[i] = so[i];
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

static int launch_head_softmax(const f16 *z, const float *sc, const float *sh, const float *w, const float *bias, int cin, int cs,
                               int K, long long n_pix, float *probs, hipStream_t stream) {
    static const int cap = []() { const char *e = getenv("IMK_HEAD_BLOCKS"); return e ? atoi(e) : 2048; }();
    const long long want = (n_pix + 255) / 256;
    const int nb = (int)(want < cap ? want : cap);
    const size_t lds = (size_t)4 * 64 * K * sizeof(float);
    const int vec = (reinterpret_cast<uintptr_t>(probs) & 15) == 0;
    const int kt = (K + 15) / 16;
#define IMK_HS(KT) imk_klaunch(head_softmax_kernel<KT>, dim3(nb), dim3(256), lds, stream, z, sc, sh, w, bias, cin, cs, K, n_pix, probs, vec)
    switch (kt) {
        case 1: IMK_HS(1); break;
        case 2: IMK_HS(2); break;
        case 3: IMK_HS(3); break;
        default: IMK_HS(4); break;
    }
#undef IMK_HS
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}

int imk_launch_head(const f16 *z, const float *sc, const float *sh, const float *w, const float *bias, int cin, int cs,
                    int K, int softmax, long long n_pix, float *probs, hipStream_t stream) {
    if (K > 64) return IMK_EUNSUPPORTED;
    if (softmax && (cs == 8 || cs == 16 || cs == 24 || cs == 32)) {
        ImkProfScope prof(PF_HEAD, (double)n_pix * (cs * 2 + K * 4), stream);
        return launch_head_softmax(z, sc, sh, w, bias, cin, cs, K, n_pix, probs, stream);
    }
    const int nb = (int)((n_pix + 255) / 256);
    const size_t lds = ((size_t)K * cs + K + 2 * cs + 256 * (size_t)(K | 1)) * sizeof(float);
    ImkProfScope prof(PF_HEAD, (double)n_pix * (cs * 2 + K * 4), stream);
#define IMK_HEAD(CS)                                                                                                     \
    do {                                                                                                                 \
        if (lds > 64 * 1024)                                                                                             \
            IMK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(head_kernel<CS>),                                  \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                          \
        imk_klaunch(head_kernel<CS>, dim3(nb), dim3(256), lds, stream, z, sc, sh, w, bias, cin, K, softmax, n_pix, probs);                    \
    } while (0)
    switch (cs) {
        case 8: IMK_HEAD(8); break;
        case 16: IMK_HEAD(16); break;
        case 24: IMK_HEAD(24); break;
        case 32: IMK_HEAD(32); break;
        default: return IMK_EUNSUPPORTED;
    }
#undef IMK_HEAD
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}

// blocks of head_loss_kernel = rows of its loss partials: grid-stride over the pixels beyond IMK_HEAD_LOSS_BLOCKS blocks
int imk_loss_blocks(long long n_pix) {
    static const int cap = []() { const char *e = getenv("IMK_HEAD_LOSS_BLOCKS"); return e ? atoi(e) : 1024; }();
    const long long nb = (n_pix + 255) / 256;
    return (int)(nb > cap ? cap : nb);
}

int imk_launch_head_loss(const f16 *z, const float *sc, const float *sh, const float *w, const float *bias, int cin, int cs,
                         int K, int softmax, long long n_pix, const uint8_t *y, const ImkCtl *ctl, float *stats,
                         f16 *dlogit, float *loss_partial, hipStream_t stream) {
    if (K > 64) return IMK_EUNSUPPORTED;
    const int nb = imk_loss_blocks(n_pix);
    const size_t lds = ((size_t)K * cs + K + 2 * cs + (softmax ? 256 * (size_t)(K | 1) : 0)) * sizeof(float);
    const int cs_out = imk_pad8(K);
    ImkProfScope prof(PF_HEAD_LOSS, (double)n_pix * (cs * 2 + (softmax ? 1 : K) + cs_out * 2), stream);
#define IMK_HL(CS) if (lds > 64 * 1024) IMK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(head_loss_kernel<CS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    imk_klaunch(head_loss_kernel<CS>, dim3(nb), dim3(256), lds, stream, z, sc, sh, w, bias, cin, K, softmax, n_pix, y, ctl, stats, cs_out, dlogit, loss_partial)
    switch (cs) {
        case 8: IMK_HL(8); break;
        case 16: IMK_HL(16); break;
        case 24: IMK_HL(24); break;
        case 32: IMK_HL(32); break;
        default: return IMK_EUNSUPPORTED;
    }
#undef IMK_HL
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}


int imk_launch_loss_finalize(const float *loss_partial, long long n_pix, int K, int kind, float *stats, hipStream_t stream) {
    const double denom = kind == 0 ? (double)n_pix * K : (double)n_pix;
    ImkProfScope prof(PF_STEP_TAIL, (double)imk_loss_blocks(n_pix) * 4, stream);
    imk_klaunch(loss_finalize_kernel, dim3(1), dim3(256), 0, stream, loss_partial, imk_loss_blocks(n_pix), denom, stats);
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}

int imk_launch_ctl_init(ImkCtl *ctl, hipStream_t stream) {
    imk_klaunch(ctl_init_kernel, dim3(1), dim3(1), 0, stream, ctl);
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}

int imk_launch_adamw(float *p, float *m, float *v, const float *g, long long n, ImkCtl *ctl, const float *stats,
                     float grad_scale, float lr, float wd, float b1, float b2, float eps, hipStream_t stream) {
    ImkProfScope prof(PF_STEP_TAIL, (double)n * 28, stream);   // p, m, v read + written, g read
    imk_klaunch(adamw_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, stream, p, m, v, g, n, ctl, stats, grad_scale, lr, wd, b1, b2, eps);
    IMK_LAUNCH_CHECK();
    return IMK_OK;   // the step counter / loss scale update rides on the re-packing launch that follows (imk_ctl_end_step)
}

int imk_launch_concat_pool(const f16 *za, const float *sca, const float *sha, int csa, const f16 *zb, const float *scb,
                           const float *shb, int csb, int B, int Hh, int Wh, f16 *cat, hipStream_t stream) {
    const long long n = (long long)B * Hh * Wh * ((csa + csb) / 8);
    imk_klaunch(concat_pool_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, stream, za, sca, sha, csa, zb, scb, shb, csb, B, Hh, Wh, cat);
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}

int imk_launch_onehot(const uint8_t *cls, long long n_pix, int cs, f16 *out, hipStream_t stream) {
    const long long n = n_pix * (cs / 8);
    imk_klaunch(onehot_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, stream, cls, n_pix, cs, out);
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}

size_t imk_evalnet_head_partial_floats(int B, int n_heads, int K, int C) { return (size_t)B * ((size_t)n_heads * K * (C + 1) + 2); }

int imk_launch_evalnet_head(const f16 *z, const float *sc, const float *sh, const float *const *w, const float *const *bias,
                            int n_heads, int K, int C, int cs, int B, int H, int W, float *out, const float *y,
                            const ImkCtl *ctl, float *stats, f16 *dP, float *partial, hipStream_t stream) {
    if (n_heads < 1 || n_heads > 2 || n_heads * K > 128 || H < 2 || W < 2) return IMK_EUNSUPPORTED;
    EvalHeadArgs a{z, sc, sh, {w[0], n_heads > 1 ? w[1] : nullptr}, {bias[0], n_heads > 1 ? bias[1] : nullptr},
                   n_heads, K, C, cs, B, H, W, out, y, ctl, stats, dP, partial};
    imk_klaunch(evalnet_head_kernel, dim3(B), dim3(256), (size_t)(cs + 128) * sizeof(float), stream, a);
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}

int imk_launch_evalnet_head_reduce(const float *partial, int B, int n_heads, int K, int C, const float *inv_scale_ptr,
                                   float *dw0, float *db0, float *dw1, float *db1, float *found_inf, float *stats,
                                   hipStream_t stream) {
    const int n = n_heads * K * (C + 1);
    imk_klaunch(evalnet_head_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, partial, B, n_heads, K, C, inv_scale_ptr, dw0, db0, dw1, db1,
                                                                   found_inf, stats);
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}
