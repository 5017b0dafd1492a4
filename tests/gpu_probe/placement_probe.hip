// Where does the hardware put the workgroups of a SMALL grid?  256 / 512 / 1 024 workgroups of 256 threads with an LDS request like the
// GEMM-class kernels' (35 KB: up to 4 fit a compute unit; 56 KB: 2; 120 KB: 1) -- every workgroup records its XCC id and HW_ID (shader
// engine / array / compute unit) and spins for ~20 us so that all of them are resident together; the host counts the distinct compute
// units used and the workgroups per unit.   hipcc --offload-arch=gfx950 -O2 -o build/placement_probe tests/gpu_probe/placement_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void probe(unsigned *out, long long spin) {
    extern __shared__ char lds[];
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
        lds[0] = 1;
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) { }
}
int main() {
    unsigned *d;
    hipMalloc(&d, 2 * 4096 * sizeof(unsigned));
    for (int lds : {35 * 1024, 56 * 1024, 120 * 1024}) {
        hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        for (int grid : {64, 128, 256, 512, 1024}) {
            std::vector<unsigned> h(2 * grid);
            hipLaunchKernelGGL(probe, dim3(grid), dim3(256), lds, 0, d, 2000LL);      // 100 MHz counter: 20 us
            if (hipDeviceSynchronize() != hipSuccess) { printf("lds %d grid %d: launch failed\n", lds, grid); continue; }
            hipMemcpy(h.data(), d, 2 * grid * sizeof(unsigned), hipMemcpyDeviceToHost);
            std::map<unsigned, int> per_cu;
            std::map<unsigned, int> per_xcc;
            for (int b = 0; b < grid; ++b) {
                const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
                const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
                per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
                per_xcc[xcc]++;
            }
            int hist[9] = {0};
            for (auto &kv : per_cu) hist[kv.second > 8 ? 8 : kv.second]++;
            printf("lds %3d KB grid %4d: %3zu compute units used on %zu XCDs; units with 1/2/3/4+ workgroups: %d / %d / %d / %d\n", lds / 1024, grid,
                   per_cu.size(), per_xcc.size(), hist[1], hist[2], hist[3], hist[4] + hist[5] + hist[6] + hist[7] + hist[8]);
        }
    }
    return 0;
}
