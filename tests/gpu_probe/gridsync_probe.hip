// What would a cooperative "deep block" kernel pay per layer?  K dependent phases over a small tensor (2 MiB in, 2 MiB out per
// phase; every workgroup reads what workgroups on OTHER XCDs wrote in the previous phase), run three ways:
//   0  K kernel launches on one stream (what the per-layer launches of the training step do today)
//   1  one persistent kernel, grid barrier with agent-scope release / acquire fences around plain stores and loads
//   2  one persistent kernel, grid barrier without fences; the tensor is written through / read around the XCD-local L2
//      (sc1 stores and loads), so nothing has to be written back or invalidated at the barrier
// All spins are bounded: a barrier that does not complete sets an abort flag instead of hanging the GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) unsigned u4;
constexpr int NWG = 256, NT = 256, PER_PHASE = 2;   // 256 x 256 x 16 B = 1 MiB per pass, 2 passes per phase

__device__ __forceinline__ u4 ld_sc1(const u4 *p) {
    u4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void st_sc1(u4 *p, u4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}

template <int MODE>
__device__ __forceinline__ void phase(const u4 *in, u4 *out, int wg) {
    // workgroup wg reads the chunk that workgroup (wg * 37 + 11) % NWG wrote: a different XCD for almost every pair
    const int src = (wg * 37 + 11) % NWG;
#pragma unroll
    for (int q = 0; q < PER_PHASE; ++q) {
        const size_t i = ((size_t)q * NWG + src) * NT + threadIdx.x, o = ((size_t)q * NWG + wg) * NT + threadIdx.x;
        u4 v = MODE == 2 ? ld_sc1(in + i) : in[i];
        v.x += 1; v.y ^= v.x;
        if (MODE == 2) st_sc1(out + o, v); else out[o] = v;
    }
}

template <int MODE>
__global__ __launch_bounds__(NT) void one_phase(const u4 *in, u4 *out) { phase<MODE>(in, out, blockIdx.x); }

template <int MODE>
__global__ __launch_bounds__(NT) void persistent(u4 *a, u4 *b, int n_phase, unsigned *counter, int *abort_flag) {
    __shared__ int s_abort;
    for (int p = 0; p < n_phase; ++p) {
        phase<MODE>((p & 1) ? b : a, (p & 1) ? a : b, blockIdx.x);
        // ---- grid barrier ----
        if (MODE == 2) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this thread's write-through stores are acknowledged
        __syncthreads();
        if (threadIdx.x == 0) {
            if (MODE == 1) __threadfence();                  // release: write back this XCD's dirty lines
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(p + 1) * gridDim.x;
            int spins = 0, ab = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > (1 << 22) || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ab = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (ab) __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (MODE == 1) __threadfence();                  // acquire: drop stale lines
            s_abort = ab;
        }
        __syncthreads();
        if (s_abort) return;
    }
}

int main() {
    const int K = 20, REP = 50;
    const size_t bytes = (size_t)PER_PHASE * NWG * NT * sizeof(u4);
    u4 *a, *b; unsigned *counter; int *abort_flag;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&counter, 4); hipMalloc(&abort_flag, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned check[3] = {0, 0, 0};
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < REP; ++rep) {
            hipMemset(a, 0, bytes); hipMemset(b, 0, bytes); hipMemset(counter, 0, 4); hipMemset(abort_flag, 0, 4);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            if (mode == 0) for (int p = 0; p < K; ++p) one_phase<0><<<NWG, NT>>>((p & 1) ? b : a, (p & 1) ? a : b);
            else if (mode == 1) persistent<1><<<NWG, NT>>>(a, b, K, counter, abort_flag);
            else persistent<2><<<NWG, NT>>>(a, b, K, counter, abort_flag);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        int ab = 0; hipMemcpy(&ab, abort_flag, 4, hipMemcpyDeviceToHost);
        unsigned vv[4]; hipMemcpy(vv, (K & 1) ? b : a, 16, hipMemcpyDeviceToHost); struct { unsigned x; } v{vv[0]};   // x counts the phases a value went through
        check[mode] = v.x;
        printf("mode %d (%s): %.2f us per phase (best of %d, %d phases)%s  check %u\n", mode,
               mode == 0 ? "one launch per phase" : (mode == 1 ? "persistent, fences at the barrier" : "persistent, sc1 write-through / read-around, no fence"),
               best * 1000.f / K, REP, K, ab ? "  ABORTED" : "", v.x);
    }
    printf("%s\n", (check[0] == (unsigned)K && check[1] == (unsigned)K && check[2] == (unsigned)K) ? "results agree" : "RESULTS DIFFER");
    return 0;
}
