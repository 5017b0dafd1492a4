// Which compute units does bit j of a hipExtStreamCreateWithCUMask mask name?  A grid of 2 048 spinning workgroups on streams
//     with a few masks; every workgroup records XCC_ID / HW_ID; the host prints the XCDs and units used.
// (What such a stream is worth beside the library's main stream: profiles/r06_ab_side_cu_reserve.txt -- it is created with default, blocking
//  flags and synchronises with the null stream at every launch.)
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
            cus[xcc].insert((hw >> 8) & 0xff);          // cu_id[11:8] | sh_id[12] | se_id[15:13]
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

    return 0;
}
