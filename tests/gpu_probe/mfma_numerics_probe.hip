// How does v_mfma_f32_16x16x32_f16 round?  Runs N independent 16x16x32 products read from a file (A[16][32], B[32][16] fp16, C[16][16]
// fp32 per case) and writes D; tests/gpu_probe/mfma_numerics.py generates the cases and compares D with candidate accumulation models.
//   hipcc --offload-arch=gfx950 -O2 -o build/mfma_numerics_probe tests/gpu_probe/mfma_numerics_probe.hip
//   ./build/mfma_numerics_probe in.bin out.bin N
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void k(const _Float16 *A, const _Float16 *B, const float *C, float *D) {
    const int cs = blockIdx.x, l = threadIdx.x, r = l & 15, g = l >> 4;
    A += (size_t)cs * 512; B += (size_t)cs * 512; C += (size_t)cs * 256; D += (size_t)cs * 256;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = A[r * 32 + 8 * g + j]; b[j] = B[(8 * g + j) * 16 + r]; }
    f32x4 acc;
    for (int q = 0; q < 4; ++q) acc[q] = C[(4 * g + q) * 16 + r];
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    for (int q = 0; q < 4; ++q) D[(4 * g + q) * 16 + r] = acc[q];
}

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    const int n = atoi(argv[3]);
    std::vector<char> in((size_t)n * (1024 + 1024 + 1024));
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(in.data(), 1, in.size(), f) != in.size()) return 3;
    fclose(f);
    char *d;
    float *out;
    hipMalloc(&d, in.size()); hipMalloc(&out, (size_t)n * 1024);
    hipMemcpy(d, in.data(), in.size(), hipMemcpyHostToDevice);
    k<<<n, 64>>>((const _Float16 *)d, (const _Float16 *)(d + (size_t)n * 1024), (const float *)(d + (size_t)n * 2048), out);
    std::vector<float> o((size_t)n * 256);
    if (hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return 4;
    f = fopen(argv[2], "wb");
    fwrite(o.data(), 4, o.size(), f);
    fclose(f);
    return 0;
}
