// How much does it cost a streaming kernel when every workgroup ends with 2*cs 64-bit integer atomics into one of `bins`
// accumulator rows (the fixed-point BatchNorm statistics idea: consumers would derive scale / shift from the bins and the
// bn_finalize / bn_bwd_coef launches would disappear)?   hipcc --offload-arch=gfx950 -O3 -o atomic_bins_probe atomic_bins_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ __launch_bounds__(256) void body(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n, unsigned long long *bins,
                                            int nbins, int cs2, int mode) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint4 v = in[i];
        acc += v.x + v.y + v.z + v.w;
        out[i] = v;
    }
    __shared__ unsigned s[256];
    s[threadIdx.x] = acc;
    __syncthreads();
    if (mode == 0) { if (threadIdx.x < cs2) out[n + (size_t)blockIdx.x * cs2 + threadIdx.x].x = s[threadIdx.x]; return; }   // partial row
    if (threadIdx.x < cs2) atomicAdd(&bins[(size_t)(blockIdx.x % nbins) * cs2 + threadIdx.x], (unsigned long long)s[threadIdx.x]);
}
__global__ void consume(const unsigned long long *bins, int nbins, int cs2, float *out) {   // what a consumer's prologue would do
    if (threadIdx.x < cs2) {
        unsigned long long t = 0;
        for (int b = 0; b < nbins; ++b) t += bins[(size_t)b * cs2 + threadIdx.x];
        out[blockIdx.x * cs2 + threadIdx.x] = (float)t;
    }
}
int main(int argc, char **argv) {
    const size_t mb = argc > 1 ? atoi(argv[1]) : 32;       // tensor size in MiB (read once, written once)
    const int grid = argc > 2 ? atoi(argv[2]) : 1280;
    const size_t n = mb * (1 << 20) / 16;
    uint4 *in, *out; unsigned long long *bins; float *cout;
    hipMalloc(&in, n * 16); hipMalloc(&out, (n + (size_t)grid * 256) * 16); hipMalloc(&bins, 64 * 256 * 8); hipMalloc(&cout, 4096 * 256 * 4);
    hipMemset(in, 1, n * 16); hipMemset(bins, 0, 64 * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int cs2 : {16, 32, 256})
        for (int mode = 0; mode < 2; ++mode)
            for (int nbins : {1, 4, 16, 64}) {
                if (mode == 0 && nbins != 1) continue;
                for (int w = 0; w < 3; ++w) body<<<grid, 256>>>(in, out, n, bins, nbins, cs2, mode);
                hipEventRecord(e0);
                for (int it = 0; it < 50; ++it) { body<<<grid, 256>>>(in, out, n, bins, nbins, cs2, mode); consume<<<1024, 256>>>(bins, nbins, cs2, cout); }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("%zu MiB grid %d 2cs %3d %s bins %2d: %.2f us per (body + consumer)\n", mb, grid, cs2, mode ? "atomics" : "rows   ", nbins, ms * 1000 / 50);
            }
    return 0;
}
