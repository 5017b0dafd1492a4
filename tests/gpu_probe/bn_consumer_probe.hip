// Would it pay to drop the bn_finalize launch and let every CONSUMER workgroup re-reduce the producer's partial rows in its
// prologue ("probe it first with a stand-alone kernel pair")?
//   chain A (today):  producer (persistent, one partial row [2 cs] per workgroup) -> finalize (cs blocks) -> consumer (reads cs scale / shift)
//   chain B:          producer -> consumer whose workgroups each sum ALL rows before their first tile (no finalize launch)
//   chain C:          producer -> consumer launch whose FIRST 2cs workgroups are the finalize (workgroups are dispatched in index order, so
//                     they are resident before any workgroup that waits for them): they write the coefficients through to memory and
//                     bump a counter; the other workgroups issue their first tile's load, poll the counter, read the coefficients
//   chain D (round 5):  XCD-local ticket: a producer workgroup stores its row (plain stores: written through to ITS XCD's L2), waits for
//                     them, takes a ticket on the counter of its group (blockIdx % 8: the blocks the dispatcher deals to one XCD) with a
//                     WORKGROUP-scope atomic (no sc1: executed in that XCD's L2); the group's last workgroup re-reads the group's rows
//                     with L1-bypassing loads, sums them in block order and writes ONE row; the consumer sums 8 rows in its prologue.
//                     Correct only while blocks b and b + 8 share an XCD (checked here with HW_REG_XCC_ID, reported as `split groups`).
//   chain E (round 5):  no hand-off at all: a producer workgroup converts its fp32 partial sums to 2 x int64 fixed point (exact for
//                     |v| < 2^43, resolution 2^-60) and ADDS them (device-scope integer atomics, no return value) into row blockIdx % R of
//                     R replica rows; integer addition is associative, so the total is the exact sum of the partials in any order and on
//                     any placement; the consumer sums R rows in its prologue and zeroes the rows of the next launch.
// Both stream a tensor of `mb` MiB (read + written, 16 B per lane) so that the reduction competes with real traffic.
//   hipcc --offload-arch=gfx950 -O3 -o bn_consumer_probe bn_consumer_probe.hip && ./bn_consumer_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>

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

// ---- chain D
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (3 << 11)) & 0xf; }

__global__ __launch_bounds__(256) void producer_xcd(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n, float *rows, int cs2,
                                                    unsigned *counters /* 8 x 16 words */, float *rows8, unsigned *xcc_of, unsigned *split) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint4 v = in[i];
        acc += __uint_as_float((v.x & 0x007fffffu) | 0x3f800000u);
        out[i] = v;
    }
    __shared__ float s[256];
    __shared__ unsigned s_last;
    s[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < cs2) rows[(size_t)blockIdx.x * cs2 + threadIdx.x] = s[threadIdx.x] + s[threadIdx.x + 64] + s[(threadIdx.x + 128) & 255];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned g = blockIdx.x & 7, members = (gridDim.x - g + 7) >> 3;
    if (threadIdx.x == 0) {
        const unsigned x = xcc_id();
        if (blockIdx.x < 8) xcc_of[g] = x;                 // diagnostic only (racy by design: read by later launches)
        else if (xcc_of[g] != x) atomicAdd(split, 1u);
        const unsigned t = __hip_atomic_fetch_add(&counters[g * 16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        s_last = (t == members - 1);
        if (s_last) __hip_atomic_store(&counters[g * 16], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    if (!s_last) return;
    // the group's rows, in block order: thread t sums column t % cs2 over members t / cs2, + 256 / cs2, ... (two-level, fixed order)
    const int per = 256 / cs2, col = threadIdx.x % cs2, r0 = threadIdx.x / cs2;
    float a = 0.f;
    for (unsigned m = r0; m < members; m += per)
        a += __hip_atomic_load(&rows[(size_t)(g + 8 * m) * cs2 + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // sc1: past L1, from this XCD's L2
    s[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x < cs2) {
        float t = 0.f;
        for (int k = 0; k < per; ++k) t += s[k * cs2 + threadIdx.x];
        rows8[g * cs2 + threadIdx.x] = t;
    }
}

// consumer of R pre-summed float rows (chain D: R = 8)
__global__ __launch_bounds__(256) void consumer_rows(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n, const float *__restrict__ rows8,
                                                     int R, int cs2) {
    __shared__ float s_sc[256];
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint4 v = i < n ? in[i] : uint4{0, 0, 0, 0};
    if ((int)threadIdx.x < cs2) {
        float t = 0.f;
        for (int r = 0; r < R; ++r) t += rows8[r * cs2 + threadIdx.x];
        s_sc[threadIdx.x] = rsqrtf(fabsf(t) + 1.0f);
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

// ---- chain E: exact integer accumulation.  v = A * 2^-10 + B * 2^-60, A = trunc(v * 2^10), B = rint((v - A * 2^-10) * 2^60)
__global__ __launch_bounds__(256) void producer_int(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n, long long *acc_rows /* [R][cs2][2] */,
                                                    int R, int cs2) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint4 v = in[i];
        acc += __uint_as_float((v.x & 0x007fffffu) | 0x3f800000u);
        out[i] = v;
    }
    __shared__ float s[256];
    s[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < 2 * cs2) {
        const int c = threadIdx.x >> 1, w = threadIdx.x & 1;
        const double v = (double)(s[c] + s[c + 64] + s[(c + 128) & 255]);
        const double a = trunc(v * 1024.0);
        const long long q = w ? (long long)rint((v - a * (1.0 / 1024.0)) * 0x1p60) : (long long)a;
        long long *dst = acc_rows + ((size_t)(blockIdx.x % R) * cs2 + c) * 2 + w;
        __hip_atomic_fetch_add(dst, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ __launch_bounds__(256) void consumer_int(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n, const long long *__restrict__ acc_rows,
                                                    long long *__restrict__ zero_rows, int R, int cs2) {
    __shared__ float s_sc[256];
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint4 v = i < n ? in[i] : uint4{0, 0, 0, 0};
    if ((int)threadIdx.x < cs2) {
        long long a = 0, b = 0;
        for (int r = 0; r < R; ++r) {
            const longlong2 q = *reinterpret_cast<const longlong2 *>(acc_rows + ((size_t)r * cs2 + threadIdx.x) * 2);
            a += q.x; b += q.y;
        }
        const float t = (float)((double)a * (1.0 / 1024.0) + (double)b * 0x1p-60);
        s_sc[threadIdx.x] = rsqrtf(fabsf(t) + 1.0f);
    }
    if (blockIdx.x == gridDim.x - 1)                       // the rows the NEXT producer adds into (the optimizer's launch in the real step)
        for (int k = threadIdx.x; k < R * cs2 * 2; k += 256) zero_rows[k] = 0;
    __syncthreads();
    const float sc = s_sc[threadIdx.x % cs2];
    for (; i < n; i += (size_t)gridDim.x * 256) {
        v.x = __float_as_uint(__uint_as_float(v.x) * sc);
        out[i] = v;
        const size_t nx = i + (size_t)gridDim.x * 256;
        if (nx < n) v = in[nx];
    }
}

// ---- write-through stores (round 5): does a kernel that leaves nothing dirty in the L2s hand over faster?  The floor chain with
// every 16-byte output store as `global_store_dwordx4 ... sc1`.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_sc1(uint4 *p, uint4 v) {
    const u32x4_t q = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(q) : "memory");
}
template <int WT>
__global__ __launch_bounds__(256) void producer_wt(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n, float *__restrict__ rows, int cs2) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint4 v = in[i];
        acc += __uint_as_float((v.x & 0x007fffffu) | 0x3f800000u);
        if (WT == 1) store_sc1(out + i, v); else if (WT == 2) __builtin_nontemporal_store(v.x, &out[i].x), __builtin_nontemporal_store(v.y, &out[i].y), __builtin_nontemporal_store(v.z, &out[i].z), __builtin_nontemporal_store(v.w, &out[i].w); else out[i] = v;
    }
    __shared__ float s[256];
    s[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < cs2) rows[(size_t)blockIdx.x * cs2 + threadIdx.x] = s[threadIdx.x] + s[(threadIdx.x + 64) & 255] + s[(threadIdx.x + 128) & 255];
}
template <int WT>
__global__ __launch_bounds__(256) void consumer_wt(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n, int cs2, const float *__restrict__ scale) {
    __shared__ float s_sc[256];
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint4 v = i < n ? in[i] : uint4{0, 0, 0, 0};
    if (threadIdx.x < cs2) s_sc[threadIdx.x] = scale[threadIdx.x];
    __syncthreads();
    const float sc = s_sc[threadIdx.x % cs2];
    for (; i < n; i += (size_t)gridDim.x * 256) {
        v.x = __float_as_uint(__uint_as_float(v.x) * sc);
        if (WT == 1) store_sc1(out + i, v); else out[i] = v;
        const size_t nx = i + (size_t)gridDim.x * 256;
        if (nx < n) v = in[nx];
    }
}

int main(int argc, char **argv) {
    const size_t mb = argc > 1 ? atoi(argv[1]) : 32;
    const int first_mode = argc > 2 ? atoi(argv[2]) : 0;
    const size_t n = mb * (1 << 20) / 16;
    uint4 *a, *b, *c; float *rows, *scale, *rows8; unsigned *flag, *counters, *xcc_of, *split; long long *acc;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&c, n * 16); hipMalloc(&rows, 4096 * 256 * 4); hipMalloc(&scale, 256 * 4);
    hipMalloc(&flag, 256); hipMemset(flag, 0, 256);
    hipMalloc(&rows8, 8 * 256 * 4); hipMalloc(&counters, 8 * 64); hipMemset(counters, 0, 8 * 64);
    hipMalloc(&xcc_of, 64); hipMemset(xcc_of, 0, 64); hipMalloc(&split, 4); hipMemset(split, 0, 4);
    const size_t acc_words = 32 * 256 * 2;                      // up to 32 replica rows x 256 statistics x 2 words
    hipMalloc(&acc, 2 * acc_words * 8); hipMemset(acc, 0, 2 * acc_words * 8);
    unsigned launches = 0, cur_cs2 = 0;
    hipMemset(a, 1, n * 16);
    {   // different mantissas per element so that the sums are not trivial
        unsigned *h = (unsigned *)malloc(n * 16);
        unsigned x = 12345u;
        for (size_t k = 0; k < n * 4; ++k) { x = x * 1664525u + 1013904223u; h[k] = x; }
        hipMemcpy(a, h, n * 16, hipMemcpyHostToDevice); free(h);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    static float h_rows[4096 * 256], h_rows8[8 * 256]; static long long h_acc[32 * 256 * 2];
    for (int cs2 : {16, 32, 128, 256})
        for (int grid : {256, 512, 1024, 1536})
            for (int mode = first_mode; mode < 12; ++mode) {
                if (mode == 2 && cur_cs2 != (unsigned)cs2) { hipMemset(flag, 0, 256); launches = 0; cur_cs2 = cs2; hipDeviceSynchronize(); }
                const int R = mode == 4 ? 8 : (mode == 5 ? 32 : 1);        // modes 4 / 5 / 6: chain E with 8 / 32 / 1 replica rows
                unsigned it_no = 0;
                if (mode >= 4 && mode < 7) { hipMemset(acc, 0, 2 * acc_words * 8); hipDeviceSynchronize(); }
                auto chain = [&]() {
                    if (mode == 0) { producer<<<grid, 256>>>(a, b, n, rows, cs2); finalize<<<cs2, 256>>>(rows, grid, cs2, scale); consumer<false><<<grid, 256>>>(b, c, n, rows, grid, cs2, scale); }
                    else if (mode == 1) { producer<<<grid, 256>>>(a, b, n, rows, cs2); consumer<true><<<grid, 256>>>(b, c, n, rows, grid, cs2, scale); }
                    else if (mode == 2) { producer<<<grid, 256>>>(a, b, n, rows, cs2); ++launches; consumer_lead<<<grid + cs2, 256>>>(b, c, n, rows, grid, cs2, scale, flag, launches * cs2, flag + 32); }
                    else if (mode == 8) { producer_int<<<grid, 256>>>(a, b, n, acc, 8, cs2); consumer<false><<<grid, 256>>>(b, c, n, rows, grid, cs2, scale); }   // E's producer side only (totals grow: never read)
                    else if (mode == 9) { producer<<<grid, 256>>>(a, b, n, rows, cs2); consumer_int<<<grid, 256>>>(b, c, n, acc, acc + acc_words, 8, cs2); }      // E's consumer side only (stale rows)
                    else if (mode == 10) { producer_wt<1><<<grid, 256>>>(a, b, n, rows, cs2); consumer_wt<1><<<grid, 256>>>(b, c, n, cs2, scale); }     // floor, write-through stores
                    else if (mode == 11) { producer_wt<0><<<grid, 256>>>(a, b, n, rows, cs2); consumer_wt<0><<<grid, 256>>>(b, c, n, cs2, scale); }     // floor, plain stores (same kernels)
                    else if (mode == 7) { producer<<<grid, 256>>>(a, b, n, rows, cs2); consumer<false><<<grid, 256>>>(b, c, n, rows, grid, cs2, scale); }    // floor: no reduction at all (stale scale)
                    else if (mode == 3) { producer_xcd<<<grid, 256>>>(a, b, n, rows, cs2, counters, rows8, xcc_of, split); consumer_rows<<<grid, 256>>>(b, c, n, rows8, 8, cs2); }
                    else {
                        long long *cur = acc + (it_no & 1) * acc_words, *nxt = acc + ((it_no + 1) & 1) * acc_words; ++it_no;
                        producer_int<<<grid, 256>>>(a, b, n, cur, R, cs2); consumer_int<<<grid, 256>>>(b, c, n, cur, nxt, R, cs2);
                    }
                };
                for (int w = 0; w < 5; ++w) chain();
                hipEventRecord(e0);
                for (int it = 0; it < 100; ++it) chain();
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                static const char *names[] = {"finalize launch (today)", "consumer re-reduces    ", "leading finalize blocks", "XCD-local ticket, 8 rows", "int64 atomics,  8 rows  ",
                                              "int64 atomics, 32 rows  ", "int64 atomics,  1 row   ", "FLOOR: two launches, no reduction", "E producer side only (atomics)", "E consumer side only (8 rows) ", "FLOOR, sc1 (write-through) stores", "FLOOR, plain stores (same code) "};
                printf("%zu MiB, 2cs %3d, %4d rows (%3d KB per consumer workgroup): %s %.2f us per producer + consumer\n", mb, cs2, grid, grid * cs2 * 4 / 1024,
                       names[mode], ms * 1000 / 100);
                if (mode == 2) { unsigned e = 0; hipMemcpy(&e, flag + 32, 4, hipMemcpyDeviceToHost); if (e) printf("   !! %u poll time-outs\n", e); }
                if (mode == 0) hipMemcpy(h_rows, rows, (size_t)grid * cs2 * 4, hipMemcpyDeviceToHost);
                if (mode == 3) {    // the 8 rows against the rows of chain A (same producer arithmetic), and the placement assumption
                    hipMemcpy(h_rows8, rows8, 8 * cs2 * 4, hipMemcpyDeviceToHost);
                    unsigned sp = 0; hipMemcpy(&sp, split, 4, hipMemcpyDeviceToHost); hipMemset(split, 0, 4);
                    double worst = 0;
                    for (int ch = 0; ch < cs2; ++ch) {
                        double ref = 0, got = 0;
                        for (int r = 0; r < grid; ++r) ref += h_rows[(size_t)r * cs2 + ch];
                        for (int g = 0; g < 8; ++g) got += h_rows8[g * cs2 + ch];
                        const double d = fabs(ref - got) / (fabs(ref) + 1e-30); if (d > worst) worst = d;
                    }
                    printf("   chain D: worst relative difference of a total %.2e, workgroups not on their group's XCD %u (of %d x 105 launches)\n", worst, sp, grid);
                }
                if (mode >= 4 && mode < 7) {    // the integer total must be the exact sum of chain A's rows
                    const long long *cur = acc + ((it_no - 1) & 1) * acc_words;
                    hipMemcpy(h_acc, cur, (size_t)R * cs2 * 16, hipMemcpyDeviceToHost);
                    int bad = 0;
                    for (int ch = 0; ch < cs2; ++ch) {
                        long long A = 0, B = 0, rA = 0, rB = 0;
                        for (int r = 0; r < R; ++r) { A += h_acc[((size_t)r * cs2 + ch) * 2]; B += h_acc[((size_t)r * cs2 + ch) * 2 + 1]; }
                        for (int r = 0; r < grid; ++r) { const double v = h_rows[(size_t)r * cs2 + ch]; const double t = trunc(v * 1024.0); rA += (long long)t; rB += (long long)rint((v - t / 1024.0) * 0x1p60); }
                        bad += (A != rA) || (B != rB);
                    }
                    if (bad) printf("   !! chain E: %d of %d totals differ from the exact sum of the rows\n", bad, cs2);
                }
            }
    return 0;
}
