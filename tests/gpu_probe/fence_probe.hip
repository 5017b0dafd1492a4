// Cost of the "last block finalises" pattern (agent-scope release fence + atomic ticket per workgroup) on a
// streaming kernel shaped like conv_pipe: 1024 persistent workgroups, 16 B in / 16 B out per thread and tile.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void stream_kernel(const uint4 *in, uint4 *out, int n_tiles, float *partial, int *counter,
                                                     float *result) {
    float acc = 0.f;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        uint4 v = in[(size_t)tile * 256 + threadIdx.x];
        acc += __uint_as_float(v.x & 0x3f800000);
        v.x ^= 1;
        out[(size_t)tile * 256 + threadIdx.x] = v;
    }
    __shared__ float s[256];
    s[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < 16) {
        float t = 0;
        for (int i = threadIdx.x; i < 256; i += 16) t += s[i];
        // MODE 2: the partial row is written through to the agent's coherence point (sc1 store), so that no release fence
        // (which writes back every dirty line of the XCD's L2, i.e. the kernel's own output) is needed before the ticket
        if (MODE == 2) __hip_atomic_store(partial + blockIdx.x * 16 + threadIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else partial[blockIdx.x * 16 + threadIdx.x] = t;
    }
    if (MODE == 0) return;
    if (MODE == 2) {
        __shared__ int s_last2;
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this thread's partial stores are acknowledged
        __syncthreads();
        if (threadIdx.x == 0)
            s_last2 = (__hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1);
        __syncthreads();
        if (!s_last2) return;
        __shared__ double sd2[256];
        const int c = threadIdx.x & 15, sl = threadIdx.x >> 4;
        double t = 0;
        for (int r = sl; r < (int)gridDim.x; r += 16) t += __hip_atomic_load(partial + r * 16 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sd2[threadIdx.x] = t;
        __syncthreads();
        if (threadIdx.x < 16) { double u = 0; for (int i = 0; i < 16; ++i) u += sd2[i * 16 + threadIdx.x]; result[threadIdx.x] = (float)u; }
        if (threadIdx.x == 0) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    __shared__ int s_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const int ticket = __hip_atomic_fetch_add(counter, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (ticket == (int)gridDim.x - 1);
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    {
        __shared__ double sd[256];
        const int c = threadIdx.x & 15, sl = threadIdx.x >> 4;
        double t = 0;
        for (int r = sl; r < (int)gridDim.x; r += 16) t += partial[r * 16 + c];
        sd[threadIdx.x] = t;
        __syncthreads();
        if (threadIdx.x < 16) { double u = 0; for (int i = 0; i < 16; ++i) u += sd[i * 16 + threadIdx.x]; result[threadIdx.x] = (float)u; }
    }
    if (threadIdx.x == 0) *counter = 0;
}

__global__ void finalize_kernel(const float *partial, int rows, float *result) {
    {
        __shared__ double sd[256];
        const int c = threadIdx.x & 15, sl = threadIdx.x >> 4;
        double t = 0;
        for (int r = sl; r < rows; r += 16) t += partial[r * 16 + c];
        sd[threadIdx.x] = t;
        __syncthreads();
        if (threadIdx.x < 16) { double u = 0; for (int i = 0; i < 16; ++i) u += sd[i * 16 + threadIdx.x]; result[threadIdx.x] = (float)u; }
    }
}

int main() {
    const int n_tiles = getenv("PROBE_TILES") ? atoi(getenv("PROBE_TILES")) : 32 * 256;   // default: 32 images of 256 tiles -> 32 MiB in, 32 MiB out
    uint4 *in, *out; float *partial, *result; int *counter;
    hipMalloc(&in, (size_t)n_tiles * 4096); hipMalloc(&out, (size_t)n_tiles * 4096);
    hipMalloc(&partial, 1024 * 16 * 4); hipMalloc(&result, 64); hipMalloc(&counter, 4);
    hipMemset(in, 0x3f, (size_t)n_tiles * 4096); hipMemset(counter, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        for (int grid : {128, 256, 512, 1024}) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                for (int i = 0; i < 200; ++i) {
                    if (mode == 0) { stream_kernel<0><<<grid, 256>>>(in, out, n_tiles, partial, counter, result); finalize_kernel<<<1, 256>>>(partial, grid > 1024 ? 1024 : grid, result); }
                    else if (mode == 1) stream_kernel<1><<<grid, 256>>>(in, out, n_tiles, partial, counter, result);
                    else stream_kernel<2><<<grid, 256>>>(in, out, n_tiles, partial, counter, result);
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep) printf("mode %d (%s) grid %d: %.2f us per layer\n", mode, mode == 0 ? "separate finalize" : (mode == 1 ? "fused last-block, release fence" : "fused last-block, write-through partials, no fence"), grid, ms * 5.0);
            }
        }
        float r[16]; hipMemcpy(r, result, 64, hipMemcpyDeviceToHost); printf("  check mode %d: %f\n", mode, r[0]);
        hipMemset(result, 0, 64);
    }
    return 0;
}
