// Would it pay to drop the bn_finalize launch and let every CONSUMER workgroup re-reduce the producer's partial rows in its
// prologue ("probe it first with a stand-alone kernel pair")?
//   chain A (today):  producer (persistent, one partial row [2 cs] per workgroup) -> finalize (cs blocks) -> consumer (reads cs scale / shift)
//   chain B:          producer -> consumer whose workgroups each sum ALL rows before their first tile (no finalize launch)
//   chain C:          producer -> consumer launch whose FIRST 2cs workgroups are the finalize (workgroups are dispatched in index order, so
//                     they are resident before any workgroup that waits for them): they write the coefficients through to memory and
//                     bump a counter; the other workgroups issue their first tile's load, poll the counter, read the coefficients
// Both stream a tensor of `mb` MiB (read + written, 16 B per lane) so that the reduction competes with real traffic.
//   hipcc --offload-arch=gfx950 -O3 -o bn_consumer_probe bn_consumer_probe.hip && ./bn_consumer_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void producer(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n, float *__restrict__ rows, int cs2) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint4 v = in[i];
        acc += __uint_as_float((v.x & 0x007fffffu) | 0x3f800000u);
        out[i] = v;
    }
    __shared__ float s[256];
    s[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < cs2) rows[(size_t)blockIdx.x * cs2 + threadIdx.x] = s[threadIdx.x] + s[threadIdx.x + 64] + s[(threadIdx.x + 128) & 255];
}

__global__ __launch_bounds__(256) void finalize(const float *__restrict__ rows, int n_rows, int cs2, float *__restrict__ scale) {
    const int ch = blockIdx.x;        // one block per (channel, statistic)
    float a = 0.f;
    for (int r = threadIdx.x; r < n_rows; r += 256) a += rows[(size_t)r * cs2 + ch];
    __shared__ float s[256];
    s[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) scale[ch] = rsqrtf(fabsf(s[0]) + 1.0f);
}

template <bool REREDUCE>
__global__ __launch_bounds__(256) void consumer(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n, const float *__restrict__ rows,
                                                int n_rows, int cs2, const float *__restrict__ scale) {
    __shared__ float s_sc[256];
    __shared__ float s_p[256];
    // the first tile's load goes out first in both forms (it does not need the scale)
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint4 v = i < n ? in[i] : uint4{0, 0, 0, 0};
    if (REREDUCE) {
        // thread t sums column (t % cs2) over rows t / cs2, t / cs2 + 256 / cs2, ...: coalesced (a row is cs2 consecutive floats)
        const int per = 256 / cs2, col = threadIdx.x % cs2, r0 = threadIdx.x / cs2;
        float a = 0.f;
        for (int r = r0; r < n_rows; r += per) a += rows[(size_t)r * cs2 + col];
        s_p[threadIdx.x] = a;
        __syncthreads();
        if (threadIdx.x < cs2) {
            float t = 0.f;
            for (int k = 0; k < per; ++k) t += s_p[k * cs2 + threadIdx.x];
            s_sc[threadIdx.x] = rsqrtf(fabsf(t) + 1.0f);
        }
    } else {
        if (threadIdx.x < cs2) s_sc[threadIdx.x] = scale[threadIdx.x];
    }
    __syncthreads();
    const float sc = s_sc[threadIdx.x % cs2];
    for (; i < n; i += (size_t)gridDim.x * 256) {
        v.x = __float_as_uint(__uint_as_float(v.x) * sc);
        out[i] = v;
        const size_t nx = i + (size_t)gridDim.x * 256;
        if (nx < n) v = in[nx];
    }
}

// chain C.  `flag` counts finalize blocks since the start of the process (target = launches so far * cs2); bounded polling: a hang
// would cost a GPU box, a timeout only a wrong number (reported through `err`)
__global__ __launch_bounds__(256) void consumer_lead(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n, const float *__restrict__ rows,
                                                     int n_rows, int cs2, float *scale, unsigned *flag, unsigned target, unsigned *err) {
    __shared__ float s_sc[256];
    __shared__ float s[256];
    if ((int)blockIdx.x < cs2) {
        const int ch = blockIdx.x;
        float a = 0.f;
        for (int r = threadIdx.x; r < n_rows; r += 256) a += rows[(size_t)r * cs2 + ch];
        s[threadIdx.x] = a;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o]; __syncthreads(); }
        if (threadIdx.x == 0) {
            __hip_atomic_store(&scale[ch], rsqrtf(fabsf(s[0]) + 1.0f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // written through (sc1)
            __builtin_amdgcn_s_waitcnt(0x0F70);                                                                      // vmcnt(0): it has arrived
            __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    const unsigned cb = blockIdx.x - cs2, cg = gridDim.x - cs2;
    size_t i = (size_t)cb * 256 + threadIdx.x;
    uint4 v = i < n ? in[i] : uint4{0, 0, 0, 0};
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 22)) { atomicAdd(err, 1u); break; }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < cs2) s_sc[threadIdx.x] = __hip_atomic_load(&scale[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const float sc = s_sc[threadIdx.x % cs2];
    for (; i < n; i += (size_t)cg * 256) {
        v.x = __float_as_uint(__uint_as_float(v.x) * sc);
        out[i] = v;
        const size_t nx = i + (size_t)cg * 256;
        if (nx < n) v = in[nx];
    }
}

int main(int argc, char **argv) {
    const size_t mb = argc > 1 ? atoi(argv[1]) : 32;
    const size_t n = mb * (1 << 20) / 16;
    uint4 *a, *b, *c; float *rows, *scale; unsigned *flag;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&c, n * 16); hipMalloc(&rows, 4096 * 256 * 4); hipMalloc(&scale, 256 * 4);
    hipMalloc(&flag, 256); hipMemset(flag, 0, 256);
    unsigned launches = 0, cur_cs2 = 0;
    hipMemset(a, 1, n * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int cs2 : {16, 32, 128, 256})
        for (int grid : {256, 512, 1024, 1536})
            for (int mode = 0; mode < 3; ++mode) {
                if (mode == 2 && cur_cs2 != (unsigned)cs2) { hipMemset(flag, 0, 256); launches = 0; cur_cs2 = cs2; hipDeviceSynchronize(); }
                auto chain = [&]() {
                    producer<<<grid, 256>>>(a, b, n, rows, cs2);
                    if (mode == 0) { finalize<<<cs2, 256>>>(rows, grid, cs2, scale); consumer<false><<<grid, 256>>>(b, c, n, rows, grid, cs2, scale); }
                    else if (mode == 1) consumer<true><<<grid, 256>>>(b, c, n, rows, grid, cs2, scale);
                    else { ++launches; consumer_lead<<<grid + cs2, 256>>>(b, c, n, rows, grid, cs2, scale, flag, launches * cs2, flag + 32); }
                };
                for (int w = 0; w < 5; ++w) chain();
                hipEventRecord(e0);
                for (int it = 0; it < 100; ++it) chain();
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("%zu MiB, 2cs %3d, %4d rows (%3d KB per consumer workgroup): %s %.2f us per producer + consumer\n", mb, cs2, grid, grid * cs2 * 4 / 1024,
                       mode == 0 ? "finalize launch (today)" : (mode == 1 ? "consumer re-reduces    " : "leading finalize blocks"), ms * 1000 / 100);
                if (mode == 2) { unsigned e = 0; hipMemcpy(&e, flag + 32, 4, hipMemcpyDeviceToHost); if (e) printf("   !! %u poll time-outs\n", e); }
            }
    return 0;
}
