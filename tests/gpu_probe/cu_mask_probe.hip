// (1) Which compute units does bit j of a hipExtStreamCreateWithCUMask mask name?  A grid of 2 048 spinning workgroups on streams
//     with a few masks; every workgroup records XCC_ID / HW_ID; the host prints the XCDs and units used.
// (2) What does a latency-bound chain on one stream pay when another stream's long workgroups hold every wave slot of the chip, and
//     what if the other stream may not use R units per XCD?  main: 40 dependent launches of 64 x 256 threads (a BatchNorm reduction's
//     shape); side: one launch of 512 x 1 024 threads spinning 400 us (two workgroups fill a unit's 32 wave slots).
//   hipcc --offload-arch=gfx950 -O2 -o build/cu_mask_probe tests/gpu_probe/cu_mask_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>

__global__ __launch_bounds__(1024) void spin(unsigned *out, long long ticks) {
    if (threadIdx.x == 0 && out) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
}
__global__ __launch_bounds__(256) void tiny(float *p) {
    if (threadIdx.x == 0) p[blockIdx.x] += 1.f;
}

static hipStream_t masked(const std::vector<int> &bits) {
    uint32_t m[8] = {0};
    for (int b : bits) m[b / 32] |= 1u << (b % 32);
    hipStream_t s = nullptr;
    if (hipExtStreamCreateWithCUMask(&s, 8, m) != hipSuccess) { printf("mask stream failed\n"); return nullptr; }
    return s;
}

int main() {
    unsigned *d; float *f;
    hipMalloc(&d, 2 * 4096 * sizeof(unsigned)); hipMalloc(&f, 4096 * sizeof(float)); hipMemset(f, 0, 4096 * sizeof(float));
    auto where = [&](const char *what, hipStream_t s) {
        const int grid = 2048;
        std::vector<unsigned> h(2 * grid);
        hipLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, s, d, 2000LL);
        hipStreamSynchronize(s);
        hipMemcpy(h.data(), d, 2 * grid * sizeof(unsigned), hipMemcpyDeviceToHost);
        std::map<unsigned, std::set<unsigned>> cus;     // xcc -> units
        for (int b = 0; b < grid; ++b) {
            const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
            cus[xcc].insert((hw >> 8) & 0x7fff);        // cu | sh | se
        }
        size_t n = 0;
        printf("%-44s:", what);
        for (auto &kv : cus) { printf(" xcd%u:%zu", kv.first, kv.second.size()); n += kv.second.size(); }
        printf("  = %zu units\n", n);
    };
    std::vector<int> v;
    where("no mask", nullptr);
    v.clear(); for (int j = 0; j < 32; ++j) v.push_back(j);
    where("bits 0..31", masked(v));
    v.clear(); for (int j = 0; j < 256; ++j) if (j % 8 == 0) v.push_back(j);
    where("bits j % 8 == 0", masked(v));
    v.clear(); for (int j = 0; j < 224; ++j) v.push_back(j);
    where("bits 0..223", masked(v));
    v.clear(); for (int j = 0; j < 256; ++j) if (j % 32 < 28) v.push_back(j);
    where("bits j % 32 < 28", masked(v));

    // (2) the chain beside a chip-filling launch
    hipStream_t mainS; hipStreamCreateWithFlags(&mainS, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto chain = [&](const char *what, hipStream_t side, bool hog) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipDeviceSynchronize();
            if (hog) hipLaunchKernelGGL(spin, dim3(512), dim3(1024), 0, side, (unsigned *)nullptr, 40000LL);   // 400 us
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, mainS, (unsigned *)nullptr, 2000LL);                // let the side launch spread first
            hipEventRecord(e0, mainS);
            for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, mainS, f);
            hipEventRecord(e1, mainS);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("%-64s: %.2f us per launch of the chain\n", what, best * 1000.f / 40);
    };
    hipStream_t plain; hipStreamCreateWithFlags(&plain, hipStreamNonBlocking);
    chain("chain alone", plain, false);
    chain("beside 512 x 1 024 threads, side stream unmasked", plain, true);
    for (int R : {1, 2, 4}) {
        char name[128];
        // two candidate layouts of "R units per XCD kept free": by bit position within a group of 32, and by bit position modulo 8 groups
        v.clear(); for (int j = 0; j < 256; ++j) if (j % 32 < 32 - R) v.push_back(j);
        snprintf(name, sizeof name, "side masked: bits j %% 32 < %d", 32 - R);
        chain(name, masked(v), true);
        v.clear(); for (int j = 0; j < 256 - 8 * R; ++j) v.push_back(j);
        snprintf(name, sizeof name, "side masked: bits 0..%d", 255 - 8 * R);
        chain(name, masked(v), true);
    }
    return 0;
}
