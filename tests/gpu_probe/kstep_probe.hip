// What can ONE wave per SIMD do with the GEMM-class kernel's k-step (16 x v_mfma_f32_16x16x32_f16 on 4 x 4 fragments)?
//   mode 0: MFMAs only, operands in registers                       mode 1: + 4 ds_read_b128 per step (next step's pixel operand)
//   mode 2: + 4 buffer_load_dwordx4 per step, AD steps ahead (L2-resident 64 KB weight slab)     mode 3: the same from a 64 MB slab (misses)
// 256 workgroups x 256 threads (one workgroup per compute unit), 2 000 steps; prints cycles per step (s_memtime) -- 256 is MFMA-bound.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o build/kstep_probe tests/gpu_probe/kstep_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// the REAL pack's addresses for a 512 -> 512-channel 3x3 (32 channel tiles x 144 k-steps x 1 KB; 8 tiles per workgroup, 4 workgroup rows `by`):
//   pat 1: [channel tile][k-step], as imk_pack lays it out (tile stride 144 KB)     pat 2: [k-step][channel tile]     pat 3: pat 1 with a 145 KB stride
__device__ inline void pack_offset(int pat, int by, int wn, int pn, int step, unsigned &so, unsigned &ms) {
    const unsigned ct = (unsigned)(by * 8 + wn * pn), f = (unsigned)(step % 144);
    if (pat == 2) { so = f * 32u * 1024u + ct * 1024u; ms = 1024u; }
    else { ms = (pat == 3 ? 145u : 144u) * 1024u; so = ct * ms + f * 1024u; }
}

template <int MODE, int AD>
__global__ __launch_bounds__(256, 2) void kstep(const f16 *w, unsigned w_bytes, float *out, unsigned long long *cyc, int steps, int pat) {
    __shared__ __attribute__((aligned(16))) f16 tile[128 * 40];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 128 * 40; i += 256) tile[i] = (f16)(0.001f * (i & 63));
    __syncthreads();
    f32x4 acc[4][4];
    for (int m = 0; m < 4; ++m) for (int p = 0; p < 4; ++p) acc[m][p] = f32x4{0, 0, 0, 0};
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16 *>(w), 0, 0x7fffffff, 0x00020000);
    constexpr int NA = AD + 1;
    f16x8 af[NA][4], bb[2][4];
    const unsigned mask = w_bytes - 1;
    auto loadA = [&](f16x8 (&a)[4], int step) {
        unsigned so = ((unsigned)(blockIdx.x & 7) * 4096u + (unsigned)step * 4096u * 8u + (unsigned)wave * 65536u) & mask, ms = 1024u;
        if (pat) pack_offset(pat, (blockIdx.x >> 3) & 3, wave >> 1, 4, step, so, ms);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if constexpr (MODE >= 2 && MODE != 4) a[m] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, (int)((so + m * ms) & mask), 0));
            else a[m] = f16x8{(f16)1, (f16)2, (f16)3, (f16)4, (f16)5, (f16)6, (f16)7, (f16)(step & 3)};
        }
    };
    auto loadB = [&](f16x8 (&b)[4], int step) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if constexpr (MODE >= 1 && MODE != 4) b[p] = *reinterpret_cast<const f16x8 *>(tile + ((p * 16 + (lane & 15)) * 40 + (lane >> 4) * 8 + (step & 1) * 0));
            else b[p] = f16x8{(f16)1, (f16)1, (f16)1, (f16)1, (f16)1, (f16)1, (f16)1, (f16)(step & 1)};
        }
    };
    auto mma = [&](const f16x8 (&a)[4], const f16x8 (&b)[4]) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m], b[p], acc[m][p], 0, 0, 0);
    };
#pragma unroll
    for (int i = 0; i < AD; ++i) loadA(af[i], i);
    loadB(bb[0], 0);
    constexpr int U = (NA % 2 == 0) ? NA : 2 * NA;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int f = 0; f + U <= steps; f += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (MODE == 4) {
                // interleaved: one fragment request + one pixel-operand read, then the four MFMAs of one pixel group, four times
                const unsigned so = ((unsigned)(blockIdx.x & 7) * 4096u + (unsigned)(f + u + AD) * 4096u * 8u + (unsigned)wave * 65536u) & mask;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    af[(u + AD) % NA][q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, (int)((so + q * 1024u) & mask), 0));
                    bb[(u + 1) & 1][q] = *reinterpret_cast<const f16x8 *>(tile + ((q * 16 + (lane & 15)) * 40 + (lane >> 4) * 8));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[u % NA][m], bb[u & 1][q], acc[m][q], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
            loadA(af[(u + AD) % NA], f + u + AD);
            loadB(bb[(u + 1) & 1], f + u + 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(af[u % NA], bb[u & 1]);
            __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int m = 0; m < 4; ++m) for (int p = 0; p < 4; ++p) s += acc[m][p][0] + acc[m][p][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// The same 16 MFMAs per step with the wave tile turned: 128 pixels x 32 channels -- TWO weight fragments requested per step (no fragment is
// requested by two waves of a workgroup) and EIGHT pixel operands read from LDS
template <int AD>
__global__ __launch_bounds__(256, 2) void kstep_wm1(const f16 *w, unsigned w_bytes, float *out, unsigned long long *cyc, int steps, int pat) {
    __shared__ __attribute__((aligned(16))) f16 tile[128 * 40];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 128 * 40; i += 256) tile[i] = (f16)(0.001f * (i & 63));
    __syncthreads();
    f32x4 acc[2][8];
    for (int m = 0; m < 2; ++m) for (int p = 0; p < 8; ++p) acc[m][p] = f32x4{0, 0, 0, 0};
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16 *>(w), 0, 0x7fffffff, 0x00020000);
    constexpr int NA = AD + 1;
    f16x8 af[NA][2], bb[2][8];
    const unsigned mask = w_bytes - 1;
    auto loadA = [&](f16x8 (&a)[2], int step) {
        unsigned so = ((unsigned)(blockIdx.x & 7) * 4096u + (unsigned)step * 4096u * 8u + (unsigned)wave * 65536u) & mask, ms = 1024u;
        if (pat) pack_offset(pat, (blockIdx.x >> 3) & 3, wave, 2, step, so, ms);
#pragma unroll
        for (int m = 0; m < 2; ++m) a[m] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, (int)((so + m * ms) & mask), 0));
    };
    auto loadB = [&](f16x8 (&b)[8]) {
#pragma unroll
        for (int p = 0; p < 8; ++p) b[p] = *reinterpret_cast<const f16x8 *>(tile + ((p * 16 + (lane & 15)) * 40 + (lane >> 4) * 8));
    };
#pragma unroll
    for (int i = 0; i < AD; ++i) loadA(af[i], i);
    loadB(bb[0]);
    constexpr int U = (NA % 2 == 0) ? NA : 2 * NA;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int f = 0; f + U <= steps; f += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            loadA(af[(u + AD) % NA], f + u + AD);
            loadB(bb[(u + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < 8; ++p)
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[m][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[u % NA][m], bb[u & 1][p], acc[m][p], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int m = 0; m < 2; ++m) for (int p = 0; p < 8; ++p) s += acc[m][p][0] + acc[m][p][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int AD>
void run_wm1(const char *what, const f16 *w, unsigned w_bytes, float *out, unsigned long long *cyc, int grid, int pat = 0) {
    const int steps = 2040;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((kstep_wm1<AD>), dim3(grid), dim3(256), 0, 0, w, w_bytes, out, cyc, steps, pat);
    hipDeviceSynchronize();
    unsigned long long h[1024];
    hipMemcpy(h, cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mn = 1e30, mx = 0, sum = 0;
    for (int i = 0; i < grid; ++i) { const double c = (double)h[i] / steps; mn = c < mn ? c : mn; mx = c > mx ? c : mx; sum += c; }
    printf("%-58s grid %4d AD %d: %.0f cycles per k-step (min %.0f, max %.0f over workgroups)\n", what, grid, AD, sum / grid, mn, mx);
}

template <int MODE, int AD>
void run(const char *what, const f16 *w, unsigned w_bytes, float *out, unsigned long long *cyc, int grid, int pat = 0) {
    const int steps = 2040;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((kstep<MODE, AD>), dim3(grid), dim3(256), 0, 0, w, w_bytes, out, cyc, steps, pat);
    hipDeviceSynchronize();
    unsigned long long h[1024];
    hipMemcpy(h, cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mn = 1e30, mx = 0, sum = 0;
    for (int i = 0; i < grid; ++i) { const double c = (double)h[i] / steps; mn = c < mn ? c : mn; mx = c > mx ? c : mx; sum += c; }
    printf("%-58s grid %4d AD %d: %.0f cycles per k-step (min %.0f, max %.0f over workgroups)\n", what, grid, AD, sum / grid, mn, mx);
}

int main() {
    f16 *w; float *out; unsigned long long *cyc;
    const unsigned big = 64u << 20;
    hipMalloc(&w, big); hipMemset(w, 0, big);
    hipMalloc(&out, 1024 * 256 * sizeof(float)); hipMalloc(&cyc, 1024 * sizeof(unsigned long long));
    for (int grid : {256, 512, 768}) {
        run<0, 1>("MFMAs only", w, 1u << 16, out, cyc, grid);
        run<1, 1>("+ 4 ds_read_b128 per step", w, 1u << 16, out, cyc, grid);
        run<2, 1>("+ 4 buffer loads per step, L2-resident slab (2 MB)", w, 2u << 20, out, cyc, grid);
        run<2, 2>("+ 4 buffer loads per step, L2-resident slab (2 MB)", w, 2u << 20, out, cyc, grid);
        run<2, 5>("+ 4 buffer loads per step, L2-resident slab (2 MB)", w, 2u << 20, out, cyc, grid);
        run<2, 1>("+ 4 buffer loads per step, 64 MB slab", w, big, out, cyc, grid);
        run<2, 2>("+ 4 buffer loads per step, 64 MB slab", w, big, out, cyc, grid);
        run<2, 5>("+ 4 buffer loads per step, 64 MB slab", w, big, out, cyc, grid);
        run_wm1<1>("turned tile: 2 requests + 8 LDS reads per step, 2 MB slab", w, 2u << 20, out, cyc, grid);
        run_wm1<2>("turned tile: 2 requests + 8 LDS reads per step, 2 MB slab", w, 2u << 20, out, cyc, grid);
        run_wm1<2>("turned tile: 2 requests + 8 LDS reads per step, 64 MB slab", w, big, out, cyc, grid);
        run<4, 2>("interleaved (1 request + 1 LDS read, 4 MFMAs) x 4, 2 MB slab", w, 2u << 20, out, cyc, grid);
        run<4, 5>("interleaved (1 request + 1 LDS read, 4 MFMAs) x 4, 2 MB slab", w, 2u << 20, out, cyc, grid);
        run<4, 5>("interleaved (1 request + 1 LDS read, 4 MFMAs) x 4, 64 MB slab", w, big, out, cyc, grid);
        run<2, 2>("4 buffer loads, the real pack's addresses [tile][step]", w, 8u << 20, out, cyc, grid, 1);
        run<2, 2>("4 buffer loads, pack as [step][tile]", w, 8u << 20, out, cyc, grid, 2);
        run<2, 2>("4 buffer loads, [tile][step] with a 145 KB tile stride", w, 8u << 20, out, cyc, grid, 3);
        run_wm1<2>("turned tile, the real pack's addresses [tile][step]", w, 8u << 20, out, cyc, grid, 1);
        run_wm1<2>("turned tile, pack as [step][tile]", w, 8u << 20, out, cyc, grid, 2);
        run_wm1<2>("turned tile, [tile][step] with a 145 KB tile stride", w, 8u << 20, out, cyc, grid, 3);
    }
    return 0;
}
