// Evaluation reductions of the benchmark_* functions on the GPU.
// The reference thresholds / argmaxes a batch of predictions and then loops over images in numpy
// (functions.py:1078-1151 ISIC, :1265-1339 multiclass; metrics :1767-1861).  Here one streaming kernel per batch
// produces the prediction masks and, per image, the INTEGER pixel counts every one of those metrics is a ratio of;
// the host forms the ratios with the reference's own float expressions, so results are bit-identical.
// HBM-bound: probabilities read once, mask written once, counts via LDS histograms + one atomic per bin per block.
#include "imk_common.h"

namespace {

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// counts[b][0..4] = #(gt!=0 & pred), #(gt!=0 | pred), #(gt>=128), #pred, #(gt>=128 & pred)
__global__ __launch_bounds__(256) void eval_binary_kernel(const float *__restrict__ probs, float thr, int cmp_ge,
                                                          const uint8_t *__restrict__ gt, int hw,
                                                          uint8_t *__restrict__ pred_out, unsigned long long *__restrict__ counts) {
    const int b = blockIdx.y;
    const size_t base = (size_t)b * hw;
    int c[5] = {0, 0, 0, 0, 0};
    for (int p = (blockIdx.x * 256 + threadIdx.x) * 4; p < hw; p += gridDim.x * 1024) {
        float v[4];
        uint8_t g[4];
        if (p + 3 < hw && ((base + p) & 3) == 0) {
            const float4 f = *reinterpret_cast<const float4 *>(probs + base + p);
            v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
            const uint32_t gg = *reinterpret_cast<const uint32_t *>(gt + base + p);
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = (gg >> (8 * j)) & 0xff;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = (p + j < hw) ? probs[base + p + j] : -1.f; g[j] = (p + j < hw) ? gt[base + p + j] : 0; }
        }
        uint32_t packed = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool in = p + j < hw;
            const bool pr = in && (cmp_ge ? v[j] >= thr : v[j] > thr);
            const bool gn = g[j] != 0, gh = g[j] >= 128;
            c[0] += gn && pr; c[1] += gn || pr; c[2] += gh; c[3] += pr; c[4] += gh && pr;
            packed |= (pr ? 255u : 0u) << (8 * j);
        }
        if (pred_out) {
            if (p + 3 < hw && ((base + p) & 3) == 0) *reinterpret_cast<uint32_t *>(pred_out + base + p) = packed;
            else
                for (int j = 0; j < 4 && p + j < hw; ++j) pred_out[base + p + j] = (packed >> (8 * j)) & 0xff;
        }
    }
    __shared__ int s_red[4][5];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int s = wave_sum_i(c[k]);
        if (lane == 0) s_red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        const int s = s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x];
        if (s) atomicAdd(counts + (size_t)b * 5 + threadIdx.x, (unsigned long long)s);
    }
}

// counts[b][0][v] = #(gt==v), [1][v] = #(pred==v), [2][v] = #(gt==v & pred==v), v in 0..255; counts[b][3][0] = #(pred==gt)
__global__ __launch_bounds__(256) void eval_multi_kernel(const float *__restrict__ probs, const uint8_t *__restrict__ gt,
                                                         int hw, int K, uint8_t *__restrict__ pred_out,
                                                         unsigned long long *__restrict__ counts) {
    // 256 pixels x K probabilities are one contiguous slab of the [pixel][K] tensor: it comes in as a coalesced
    // stream (a thread reading its own K floats, 4*K bytes from its neighbour's, wastes most of every line) and each
    // thread then scans its row in LDS (odd pitch: conflict-free).
    extern __shared__ float s_p[];                 // [256][K | 1]
    __shared__ unsigned int hist[3 * 256 + 1];
    for (int i = threadIdx.x; i < 3 * 256 + 1; i += 256) hist[i] = 0;
    const int pitch = K | 1;
    const int b = blockIdx.y;
    const size_t base = (size_t)b * hw;
    for (int p0 = blockIdx.x * 256; p0 < hw; p0 += gridDim.x * 256) {
        __syncthreads();                           // histogram zeroed / previous slab consumed
        const int n_here = min(256, hw - p0);
        const float *src = probs + (base + p0) * K;
        for (int i = threadIdx.x; i < n_here * K; i += 256) {
            const int px = i / K;
            s_p[px * pitch + (i - px * K)] = src[i];
        }
        __syncthreads();
        const int p = p0 + threadIdx.x;
        if (p < hw) {
            const float *q = s_p + threadIdx.x * pitch;
            float best = q[0];
            int arg = 0;
            for (int k = 1; k < K; ++k) {        // first maximum wins (np.argmax); inputs are finite
                const float v = q[k];
                if (v > best) { best = v; arg = k; }
            }
            const int g = gt[base + p];
            if (pred_out) pred_out[base + p] = (uint8_t)arg;
            atomicAdd(&hist[g], 1u);
            atomicAdd(&hist[256 + arg], 1u);
            if (g == arg) { atomicAdd(&hist[512 + g], 1u); atomicAdd(&hist[768], 1u); }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * 256 + 1; i += 256)
        if (hist[i]) atomicAdd(counts + (size_t)b * 1024 + i, (unsigned long long)hist[i]);
}

// ---- soft reductions of the validation monitors ---------------------------------------------------------------------
// mode 0 (MeanIoU.update_state, functions.py:75-86): per class k  {sum_p [gt == k] * p_k, sum_p [gt == k], sum_p p_k}
// mode 1 (Keras' val_loss for 'mse'): sum over all elements of (p - y)^2
// Deterministic: a thread owns one class (mode 0) and a fixed subset of the pixels, partials are summed in a fixed order.
constexpr int SOFT_BLOCKS = 256;

__global__ __launch_bounds__(256) void eval_soft_kernel(const float *__restrict__ probs, const uint8_t *__restrict__ gt,
                                                        long long n_pix, int K, int mode, double *__restrict__ out) {
    __shared__ float s_acc[256][3];
    const int t = threadIdx.x;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    if (mode == 0) {
        const int G = 256 / K;                       // pixel groups per block (K <= 64: at least 4)
        const int g = t / K, k = t - g * K;
        if (g < G)
            for (long long p = (long long)blockIdx.x * G + g; p < n_pix; p += (long long)gridDim.x * G) {
                const float v = probs[p * K + k];
                const bool hit = gt[p] == k;
                a0 += hit ? v : 0.f; a1 += hit ? 1.f : 0.f; a2 += v;
            }
    } else {
        const long long n = n_pix * K;
        for (long long i = (long long)blockIdx.x * 256 + t; i < n; i += (long long)gridDim.x * 256) {
            const float e = probs[i] - (float)gt[i];
            a0 += e * e;
        }
    }
    s_acc[t][0] = a0; s_acc[t][1] = a1; s_acc[t][2] = a2;
    __syncthreads();
    double *row = out + (size_t)(1 + blockIdx.x) * 3 * K;
    if (mode == 0) {
        if (t < K) {
            const int G = 256 / K;
            double r0 = 0, r1 = 0, r2 = 0;
            for (int g = 0; g < G; ++g) { r0 += s_acc[g * K + t][0]; r1 += s_acc[g * K + t][1]; r2 += s_acc[g * K + t][2]; }
            row[t] = r0; row[K + t] = r1; row[2 * K + t] = r2;
        }
    } else if (t == 0) {
        double r = 0;
        for (int i = 0; i < 256; ++i) r += s_acc[i][0];
        row[0] = r;
    }
}

__global__ __launch_bounds__(256) void eval_soft_finalize_kernel(double *__restrict__ out, int n_cols, int n_rows) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n_cols) return;
    double r = 0;
    for (int b = 0; b < n_rows; ++b) r += out[(size_t)(1 + b) * n_cols + c];
    out[c] = r;
}

}  // namespace

extern "C" int64_t imk_eval_soft_out_doubles(int k) { return k > 0 && k <= 64 ? (int64_t)(SOFT_BLOCKS + 1) * 3 * k : IMK_EINVAL; }

extern "C" int imk_eval_soft_sums(const float *probs, const uint8_t *gt, int64_t n_pix, int k, int mode, double *out,
                                  void *stream_) {
    IMK_CHECK_ARG(probs && gt && out && n_pix > 0 && k > 0 && (mode == 0 || mode == 1));
    if (k > 64) return IMK_EUNSUPPORTED;
    hipStream_t stream = (hipStream_t)stream_;
    IMK_HIP(hipMemsetAsync(out, 0, (size_t)(SOFT_BLOCKS + 1) * 3 * k * sizeof(double), stream));
    imk_klaunch(eval_soft_kernel, dim3(SOFT_BLOCKS), dim3(256), 0, stream, probs, gt, (long long)n_pix, k, mode, out);
    IMK_LAUNCH_CHECK();
    imk_klaunch(eval_soft_finalize_kernel, dim3(imk_cdiv(3 * k, 256)), dim3(256), 0, stream, out, 3 * k, SOFT_BLOCKS);
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}

extern "C" int imk_eval_binary(const float *probs, float thr, int cmp_ge, const uint8_t *gt, int batch, int h, int w,
                               uint8_t *pred_out, int64_t *counts, void *stream_) {
    IMK_CHECK_ARG(probs && gt && counts && batch > 0 && h > 0 && w > 0);
    hipStream_t stream = (hipStream_t)stream_;
    IMK_HIP(hipMemsetAsync(counts, 0, (size_t)batch * 5 * sizeof(int64_t), stream));
    const int hw = h * w;
    int bx = (int)imk_cdiv(hw, 1024);
    if (bx > 64) bx = 64;
    imk_klaunch(eval_binary_kernel, dim3(dim3(bx, batch)), dim3(256), 0, stream, probs, thr, cmp_ge, gt, hw, pred_out, (unsigned long long *)counts);
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}

extern "C" int imk_eval_multiclass(const float *probs, const uint8_t *gt, int batch, int h, int w, int k, uint8_t *pred_out,
                                   int64_t *counts, void *stream_) {
    IMK_CHECK_ARG(probs && gt && counts && batch > 0 && h > 0 && w > 0 && k > 0 && k <= 256);
    hipStream_t stream = (hipStream_t)stream_;
    IMK_HIP(hipMemsetAsync(counts, 0, (size_t)batch * 1024 * sizeof(int64_t), stream));
    const int hw = h * w;
    int bx = (int)imk_cdiv(hw, 256 * 8);
    if (bx > 64) bx = 64;
    const size_t lds = (size_t)256 * (k | 1) * sizeof(float);   // <= 65 KB: one opt-in above the 64 KB default
    if (lds > 64 * 1024)
        IMK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(eval_multi_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    imk_klaunch(eval_multi_kernel, dim3(dim3(bx, batch)), dim3(256), lds, stream, probs, gt, hw, k, pred_out, (unsigned long long *)counts);
    IMK_LAUNCH_CHECK();
    return IMK_OK;
}
