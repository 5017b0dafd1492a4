// GPU probe: checks the gfx950 fragment layouts the conv kernels rely on, with exact integer data.
//   1. v_mfma_f32_16x16x32_f16: lane l holds A[l&15][8*(l>>4)+j], B[8*(l>>4)+j][l&15], j=0..7;
//      D[4*(l>>4)+r][l&15] in accumulator register r.
//   2. ds_read_b64_tr_b16: per 16-lane group, lane 4q+p supplies the address of row q, columns 4p..4p+3
//      of a 4x16 block; lane i receives column i, row q in element q.
// Build: hipcc --offload-arch=gfx950 -O2 -o build/mfma_probe tests/gpu_probe/mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 h4;

__global__ void mfma_k(const _Float16 *A, const _Float16 *B, float *D) {  // A[16][32], B[32][16] row-major
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = A[r * 32 + 8 * g + j]; b[j] = B[(8 * g + j) * 16 + r]; }
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(4 * g + i) * 16 + r] = acc[i];
}

__global__ void tr_k(const _Float16 *src, float *out, int row_stride /*halfs*/) {  // src [4*4 rows][row_stride]
    extern __shared__ _Float16 s[];
    for (int i = threadIdx.x; i < 16 * row_stride; i += 64) s[i] = src[i];
    __syncthreads();
    const int l = threadIdx.x, grp = l >> 4, i16 = l & 15, q = i16 >> 2, p = i16 & 3;
    // group grp reads rows 4*grp .. 4*grp+3, columns 0..15
    const _Float16 *addr = s + (4 * grp + q) * row_stride + 4 * p;
    h4 t = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4 *)addr);
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = (float)t[e];
}

int main() {
    int fails = 0;
    {
        std::vector<_Float16> A(16 * 32), B(32 * 16);
        srand(1);
        for (auto &v : A) v = (_Float16)(rand() % 7 - 3);
        for (auto &v : B) v = (_Float16)(rand() % 5 - 2);
        _Float16 *dA, *dB; float *dD;
        hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dD, 256 * 4);
        hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
        mfma_k<<<1, 64>>>(dA, dB, dD);
        std::vector<float> D(256);
        hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
            float ref = 0; for (int k = 0; k < 32; ++k) ref += (float)A[m * 32 + k] * (float)B[k * 16 + n];
            if (ref != D[m * 16 + n]) ++bad;
        }
        printf("mfma_f32_16x16x32_f16 layout: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
        fails += bad != 0;
    }
    for (int stride : {16, 24, 40, 72}) {
        std::vector<_Float16> S(16 * stride);
        for (int i = 0; i < 16 * stride; ++i) S[i] = (_Float16)(float)(i % 2048);
        _Float16 *dS; float *dO;
        hipMalloc(&dS, S.size() * 2); hipMalloc(&dO, 256 * 4);
        hipMemcpy(dS, S.data(), S.size() * 2, hipMemcpyHostToDevice);
        tr_k<<<1, 64, 16 * stride * 2>>>(dS, dO, stride);
        std::vector<float> O(256);
        hipMemcpy(O.data(), dO, 256 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
            const int grp = l >> 4, col = l & 15;
            float ref = (float)S[(4 * grp + e) * stride + col];
            if (ref != O[l * 4 + e]) ++bad;
        }
        printf("ds_read_b64_tr_b16 stride %d: %s (%d mismatches)\n", stride, bad ? "FAIL" : "OK", bad);
        if (bad) for (int l = 0; l < 20; ++l) printf("  lane %d: %g %g %g %g\n", l, O[l*4], O[l*4+1], O[l*4+2], O[l*4+3]);
        fails += bad != 0;
    }
    hipError_t e = hipDeviceSynchronize();
    printf("probe done: %d failing checks, hip=%s\n", fails, hipGetErrorString(e));
    return fails;
}
