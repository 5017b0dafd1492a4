// Shared helpers for libimk.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "../../include/imk.h"

#include <hip/hip_ext.h>
#include <tuple>
#include <utility>

// ---- kernel launches that can carry a completion event ---------------------------------------------------------------------------
// Every kernel of the library is launched through imk_klaunch.  While a thread has a stop-event ring installed for a stream (the
// training step's backward pass: ImkStopRingScope in imk_net.h), a launch on THAT stream binds the ring's next event to the kernel
// itself (hipExtLaunchKernel's stopEvent: the event completes with the kernel's own completion signal) and remembers it as `last`.
// A fork to the side stream then waits for `last` instead of recording an event of its own -- a recorded event is a marker packet
// on the main stream, and the chain's next kernel started 3.6 us later behind it (0.9 us with the kernel-bound event:
// tests/gpu_probe/fork_probe.hip; ~10 forks per training step).  `last` is only meaningful because EVERY launch goes through here.
struct ImkStopRing {
    hipEvent_t *ev;
    int n, cur;
    hipStream_t stream;          // the main stream of the pass
    hipEvent_t last;             // event bound to its most recent kernel
    hipStream_t side = nullptr;  // (optional) the pass's one side stream: its last kernel's event serves the join at the end
    hipEvent_t side_last = nullptr;
};
inline thread_local ImkStopRing *imk_tls_stop_ring = nullptr;
// Measurement (imk_prof_totals_enable on the context bound to this thread): every launch is counted under the kernel's own name as
// rocprofv3 prints it (template arguments included), with the algorithmic bytes / flops the enclosing ImkProfScope announced
// (imk_conv.hip: imk_prof_note_launch).  Off: one thread-local load per launch.
inline thread_local bool imk_tls_totals_on = false;
void imk_prof_note_launch(const void *kern, hipStream_t stream);

template <typename... KA, typename... A, size_t... I>
inline hipError_t imk_klaunch_ext(void (*kern)(KA...), dim3 grid, dim3 block, size_t lds, hipStream_t stream, hipEvent_t stop,
                                  std::index_sequence<I...>, A &&...args) {
    std::tuple<KA...> held{static_cast<KA>(args)...};            // the kernel's own parameter types, as <<<>>> would convert them
    void *argv[] = {static_cast<void *>(&std::get<I>(held))..., nullptr};
    return hipExtLaunchKernel(reinterpret_cast<const void *>(kern), grid, block, argv, lds, stream, nullptr, stop, 0);
}

template <typename... KA, typename... A>
inline void imk_klaunch(void (*kern)(KA...), dim3 grid, dim3 block, size_t lds, hipStream_t stream, A &&...args) {
    static_assert(sizeof...(KA) == sizeof...(A), "kernel argument count");
    if (imk_tls_totals_on) imk_prof_note_launch(reinterpret_cast<const void *>(kern), stream);
    ImkStopRing *r = imk_tls_stop_ring;
    if (r && (r->stream == stream || (r->side && r->side == stream))) {
        hipEvent_t e = r->ev[r->cur];
        r->cur = (r->cur + 1) % r->n;
        const bool ok = imk_klaunch_ext(kern, grid, block, lds, stream, e, std::index_sequence_for<KA...>{}, std::forward<A>(args)...) == hipSuccess;
        // (on failure the error stays pending for IMK_LAUNCH_CHECK and forks / joins fall back to recorded events)
        (r->stream == stream ? r->last : r->side_last) = ok ? e : nullptr;
        return;
    }
    kern<<<grid, block, lds, stream>>>(static_cast<KA>(args)...);
}

#define IMK_CHECK_ARG(cond) do { if (!(cond)) return IMK_EINVAL; } while (0)
#define IMK_LAUNCH_CHECK() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)
#define IMK_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return (int)e__; } while (0)

static inline int imk_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int imk_pad8(int c) { return (c + 7) & ~7; }

typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;

typedef __attribute__((ext_vector_type(2))) float f32x2;
// The library's ONE definition of "a BatchNorm applied to a stored fp16 activation": fp16(z * sc + sh), the fp32 fused multiply-add
// rounded to fp16 ONCE -- v_fma_mixlo_f16 / v_fma_mixhi_f16 (fp16 source, fp32 scale and shift, fp16 result), one instruction per
// value.  Written as a plain expression the compiler picks between that and v_cvt_f32_f16 + v_pk_fma_f32 + v_cvt_pk_f16_f32 (the sum
// rounded to fp32 first, then to fp16) site by site -- and, inside one kernel, channel by channel: 1e-4 of the values then differ by
// an fp16 ulp between a fused launch and its two-launch form (round 2-3's "border pixel" discrepancy of the decoder's first-stage
// launch).  Round 4 first pinned every site to the two-rounding form (4 instructions per pair, a third of the
// vector instructions of the inference kernels); this is the same pin on the cheaper and more accurate instruction.
__device__ __forceinline__ f16x2 imk_affine2(f16x2 z, f32x2 sc, f32x2 sh) {
    f16x2 d;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, %4, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(d) : "v"(z), "v"(sc[0]), "v"(sh[0]), "v"(sc[1]), "v"(sh[1]));
    return d;
}
__device__ __forceinline__ f16 imk_affine1(f16 z, float sc, float sh) {
    f16x2 d, zz = {z, z};
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=&v"(d) : "v"(zz), "v"(sc), "v"(sh));
    return d[0];
}
// v_pk_max_f16 as written: through __builtin_elementwise_max the compiler first "canonicalises" operands it cannot see through
// (the asm results above) with a v_pk_max_f16 x, x each -- 7 instructions per pair for a 4-way maximum instead of 3
__device__ __forceinline__ f16x2 imk_pk_max(f16x2 a, f16x2 b) {
    f16x2 d;
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// MaxPooling2D of a BatchNorm output, applied on load: the window's four values by the definition above, then their maximum
__device__ __forceinline__ f16x8 imk_affine_pool8(f16x8 z0, f16x8 z1, f16x8 z2, f16x8 z3, const float *sc, const float *sh) {
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const f32x2 s2 = {sc[j], sc[j + 1]}, h2 = {sh[j], sh[j + 1]};
        const f16x2 a = imk_affine2(f16x2{z0[j], z0[j + 1]}, s2, h2), b = imk_affine2(f16x2{z1[j], z1[j + 1]}, s2, h2);
        const f16x2 c = imk_affine2(f16x2{z2[j], z2[j + 1]}, s2, h2), d = imk_affine2(f16x2{z3[j], z3[j + 1]}, s2, h2);
        const f16x2 m = imk_pk_max(imk_pk_max(a, b), imk_pk_max(c, d));
        o[j] = m[0]; o[j + 1] = m[1];
    }
    return o;
}
// relu(acc + bias) rounded to fp16, four channels at a time: v_pk_add_f32, v_cvt_pk_f16_f32, v_pk_max_f16 -- 1.5 instructions per
// value instead of 3 (add, max, convert).  Rounding is monotonic and keeps the sign, so max-after-rounding gives the value
// max-before-rounding gives (a sum that rounds to -0 yields +0 either way: v_pk_max_f16 orders -0 below +0).
__device__ __forceinline__ f16x2 imk_bias_relu2(f32x2 acc, f32x2 bias) {
    return __builtin_elementwise_max(__builtin_convertvector(acc + bias, f16x2), f16x2{0, 0});
}
__device__ __forceinline__ f16x4 imk_bias_relu4(f32x4 acc, const float *bias) {
    const f16x2 l = imk_bias_relu2(f32x2{acc[0], acc[1]}, f32x2{bias[0], bias[1]});
    const f16x2 h = imk_bias_relu2(f32x2{acc[2], acc[3]}, f32x2{bias[2], bias[3]});
    return f16x4{l[0], l[1], h[0], h[1]};
}
__device__ __forceinline__ f16x8 imk_affine8(f16x8 z, const float *sc, const float *sh) {
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const f16x2 r = imk_affine2(f16x2{z[j], z[j + 1]}, f32x2{sc[j], sc[j + 1]}, f32x2{sh[j], sh[j + 1]});
        o[j] = r[0]; o[j + 1] = r[1];
    }
    return o;
}

// 64-lane butterfly sum inside groups of `width` consecutive lanes (width = 16 or 64)
template <int WIDTH>
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = WIDTH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- measurement hook (imk_prof_enable / imk_prof_collect, include/imk.h) --------------------------------------------
// Kernel families whose launches can be bracketed with HIP events on the stream they are launched on.
enum ImkProfFamily {
    PF_CONV_MFMA = 0,    // 0..5: conv_mfma_kernel<TH,MT>: 3 * (TH == 8) + log2(MT)
    PF_CONV_PIPE = 6,    // conv_pipe_kernel (every template variant)
    PF_WGRAD = 7,        // wgrad_mfma_kernel
    PF_BN_PREP = 8,      // bn_bwd_prep_kernel / bn_bwd_prep_pool_kernel (assemble dy + reduce)
    PF_BN_COEF = 9,      // bn_bwd_coef_kernel
    PF_BN_FINALIZE = 10, // bn_finalize_kernel
    PF_WGF = 11,         // wgf_stage1_kernel + wgf_stage2_kernel (one bracket around the pair)
    PF_HEAD = 12,        // head_kernel (inference)
    PF_HEAD_LOSS = 13,   // head_loss_kernel
    PF_STEP_TAIL = 14,   // loss_finalize_kernel, adamw_kernel, pack_conv_batched_kernel, bn_fold_batched_kernel
    PF_IM = 15,          // im_binary_vec / im_binary_generic / im_multi_kernel
    PF_CONV_GEMM = 16,   // conv_gemm_kernel (imk_gemm.hip: forward / dgrad of the wide layers)
    PF_WGRAD_GEMM = 17,  // wgrad_gemm_kernel
    PF_COUNT = 18
};
// Returns a slot >= 0 when this launch is sampled (an event was recorded on `stream`), else -1.
// With totals enabled (imk_prof_totals_enable) the bound context sums launches / algorithmic bytes / flops per KERNEL NAME as
// rocprofv3 prints it (template arguments included; resolved from the launched function itself in imk_klaunch): a scope announces
// its work, the next launch on this thread takes it -- what profiles/summarize.py divides by rocprofv3's own per-kernel durations
// and prices against max(bytes / 8 TB/s, flops / 2.5 PFLOP/s).  A scope with several launches re-announces per launch
// (imk_prof_work); launches outside any scope are counted with no work ("unpriced": KB-sized).  `variant` is unused since round 6.
void imk_prof_work(double algorithmic_bytes, double flops = 0.0);
int imk_prof_begin(int family, double algorithmic_bytes, hipStream_t stream, double flops = 0.0, const char *variant = nullptr);
void imk_prof_end(int slot, hipStream_t stream);
struct ImkProfScope {
    int slot;
    hipStream_t stream;
    ImkProfScope(int family, double bytes, hipStream_t s, double flops = 0.0, const char *variant = nullptr)
        : slot(imk_prof_begin(family, bytes, s, flops, variant)), stream(s) {}
    ~ImkProfScope() { if (slot >= 0) imk_prof_end(slot, stream); }
    ImkProfScope(const ImkProfScope &) = delete;
    ImkProfScope &operator=(const ImkProfScope &) = delete;
};

// ---- in-kernel phase stamps (probe builds only: -DIMK_STAMPS; tests/gpu_probe/stamps.py) ---------------------------------
// One thread of the first workgroup of a stamped kernel writes the shader clock at a few points of its path into a
// per-translation-unit table: where a latency-bound launch spends its microseconds, without a profiler attached.
#ifdef IMK_STAMPS
#define IMK_STAMP_ROWS 4096
#define IMK_STAMP_COLS 16
#define IMK_STAMP_TABLE(tu)                                                                                          \
    __device__ unsigned long long g_stamps_##tu[IMK_STAMP_ROWS * IMK_STAMP_COLS];                                     \
    __device__ unsigned g_stamps_n_##tu;                                                                              \
    extern "C" __attribute__((visibility("default"))) int imk_debug_stamps_##tu(unsigned long long *out, int reset) { \
        unsigned n = 0;                                                                                               \
        if (hipDeviceSynchronize() != hipSuccess) return -1;                                                          \
        if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_stamps_n_##tu), sizeof n) != hipSuccess) return -1;                  \
        if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_##tu), sizeof(unsigned long long) * IMK_STAMP_ROWS * IMK_STAMP_COLS) != hipSuccess) return -1; \
        if (reset) { unsigned z = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps_n_##tu), &z, sizeof z) != hipSuccess) return -1; } \
        return (int)(n < IMK_STAMP_ROWS ? n : IMK_STAMP_ROWS);                                                        \
    }
// col 0: kernel id, col 1: gridDim.x * gridDim.y, col 2: s_memrealtime at entry (100 MHz, orders the rows), col 3..: s_memtime
#define IMK_STAMP_BEGIN(tu, kid)                                                                                      \
    unsigned long long *stamp_row_ = nullptr;                                                                         \
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {                                                     \
        const unsigned r_ = atomicAdd(&g_stamps_n_##tu, 1u);                                                          \
        if (r_ < IMK_STAMP_ROWS) {                                                                                    \
            stamp_row_ = g_stamps_##tu + (size_t)r_ * IMK_STAMP_COLS;                                                 \
            stamp_row_[0] = (unsigned long long)(kid);                                                                \
            stamp_row_[1] = (unsigned long long)gridDim.x * gridDim.y;                                                \
            stamp_row_[2] = __builtin_amdgcn_s_memrealtime();                                                         \
            for (int z_ = 4; z_ < IMK_STAMP_COLS; ++z_) stamp_row_[z_] = 0;   /* the row's previous user may have had more stamps */ \
            stamp_row_[3] = __builtin_amdgcn_s_memtime();                                                             \
        }                                                                                                             \
    }
// Per-WORKGROUP entry / exit times (100 MHz counter) of the launches whose kernel id equals the selection (host: imk_debug_wgsel_<tu>):
// how evenly a persistent launch's workgroups finish.  Later launches of the same id overwrite earlier ones.
#define IMK_WGSTAMP_ROWS 4096
#define IMK_WGSTAMP_TABLE(tu)                                                                                        \
    __device__ unsigned long long g_wgstamps_##tu[IMK_WGSTAMP_ROWS * 2];                                              \
    __device__ unsigned g_wgsel_##tu;                                                                                 \
    extern "C" __attribute__((visibility("default"))) int imk_debug_wgsel_##tu(unsigned kid) {                        \
        if (hipDeviceSynchronize() != hipSuccess) return -1;                                                          \
        return hipMemcpyToSymbol(HIP_SYMBOL(g_wgsel_##tu), &kid, sizeof kid) == hipSuccess ? 0 : -1;                  \
    }                                                                                                                 \
    extern "C" __attribute__((visibility("default"))) int imk_debug_wgstamps_##tu(unsigned long long *out) {          \
        if (hipDeviceSynchronize() != hipSuccess) return -1;                                                          \
        return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wgstamps_##tu), sizeof(unsigned long long) * IMK_WGSTAMP_ROWS * 2) == hipSuccess ? 0 : -1; \
    }
#define IMK_WGSTAMP_BEGIN(tu, kid)                                                                                    \
    const bool wgstamp_on_ = (g_wgsel_##tu == (unsigned)(kid)) && blockIdx.x < IMK_WGSTAMP_ROWS && blockIdx.y == 0;   \
    unsigned long long *const wgstamp_p_ = g_wgstamps_##tu + 2 * blockIdx.x;                                          \
    if (wgstamp_on_ && threadIdx.x == 0) wgstamp_p_[0] = __builtin_amdgcn_s_memrealtime();
#define IMK_WGSTAMP_END() do { if (wgstamp_on_ && threadIdx.x == 0) wgstamp_p_[1] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define IMK_STAMP(i) do { if (stamp_row_) stamp_row_[3 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
// last stamp of a kernel: the shader clock and (col 15) the 100 MHz counter again, which calibrates the former per launch
#define IMK_STAMP_END(i) do { if (stamp_row_) { stamp_row_[3 + (i)] = __builtin_amdgcn_s_memtime(); stamp_row_[15] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define IMK_WGSTAMP_TABLE(tu)
#define IMK_WGSTAMP_BEGIN(tu, kid)
#define IMK_WGSTAMP_END() do { } while (0)
#define IMK_STAMP_TABLE(tu)
#define IMK_STAMP_BEGIN(tu, kid)
#define IMK_STAMP(i) do { } while (0)
#define IMK_STAMP_END(i) do { } while (0)
#endif
