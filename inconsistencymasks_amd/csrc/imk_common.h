// Shared helpers for libimk.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "../../include/imk.h"

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

// 64-lane butterfly sum inside groups of `width` consecutive lanes (width = 16 or 64)
template <int WIDTH>
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = WIDTH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
