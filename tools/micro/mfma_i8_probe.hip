// mfma_i8_probe.hip -- operand/result layout and issue rate of v_mfma_i32_32x32x32_i8 / v_mfma_i32_16x16x64_i8 on gfx950.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_i8_probe mfma_i8_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// layout candidate L: 0: k = 16*(l/32)+j ; 1: k = 8*(l/32) + (j&7) + 16*(j>>3)
__global__ void k32(const signed char *A, const signed char *B, int *C, int L)
{
    const int l = threadIdx.x;
    union { v4i v; signed char b[16]; } a, b;
    for (int j = 0; j < 16; ++j) {
        const int k = L == 0 ? 16 * (l / 32) + j : 8 * (l / 32) + (j & 7) + 16 * (j >> 3);
        a.b[j] = A[(l % 32) * 32 + k];
        b.b[j] = B[(l % 32) * 32 + k];
    }
    v16i c = {0};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a.v, b.v, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int row = (r / 4) * 8 + (l / 32) * 4 + (r % 4), col = l % 32;
        C[row * 32 + col] = c[r];
    }
}
// 16x16x64: candidate 0: k = 16*(l/16)+j ; 1: k = 8*(l/16) + (j&7) + 32*(j>>3)
__global__ void k16(const signed char *A, const signed char *B, int *C, int L)
{
    const int l = threadIdx.x;
    union { v4i v; signed char b[16]; } a, b;
    for (int j = 0; j < 16; ++j) {
        const int k = L == 0 ? 16 * (l / 16) + j : 8 * (l / 16) + (j & 7) + 32 * (j >> 3);
        a.b[j] = A[(l % 16) * 64 + k];
        b.b[j] = B[(l % 16) * 64 + k];
    }
    v4i c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a.v, b.v, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l / 16) + r, col = l % 16;
        C[row * 16 + col] = c[r];
    }
}
// issue rate: NACC independent accumulators, one wave per SIMD
template <int NACC>
__global__ void __launch_bounds__(256) rate32(int iters, int *sink)
{
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)blockIdx.x, 7};
    v16i c[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = (v16i){0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[i], 0, 0, 0);
    int s = 0;
    for (int i = 0; i < NACC; ++i) s += c[i][0] + c[i][5];
    if (s == 0x12345678) *sink = s;
}
template <int NACC>
__global__ void __launch_bounds__(256) rate16(int iters, int *sink)
{
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)blockIdx.x, 7};
    v4i c[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = (v4i){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[i], 0, 0, 0);
    int s = 0;
    for (int i = 0; i < NACC; ++i) s += c[i][0] + c[i][3];
    if (s == 0x12345678) *sink = s;
}
int main()
{
    std::vector<signed char> A(32 * 64), B(32 * 64);
    srand(5);
    for (auto &v : A) v = (signed char)(rand() % 256 - 128);
    for (auto &v : B) v = (signed char)(rand() % 256 - 128);
    signed char *dA, *dB; int *dC;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dC, 32 * 32 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    std::vector<int> C(32 * 32);
    for (int L = 0; L < 2; ++L) {
        hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dC, L);
        hipMemcpy(C.data(), dC, 32 * 32 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            int s = 0;
            for (int k = 0; k < 32; ++k) s += (int)A[i * 32 + k] * (int)B[j * 32 + k];
            bad += s != C[i * 32 + j];
        }
        printf("32x32x32 layout %d: %d mismatches\n", L, bad);
    }
    for (int L = 0; L < 2; ++L) {
        hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dC, L);
        hipMemcpy(C.data(), dC, 16 * 16 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            int s = 0;
            for (int k = 0; k < 64; ++k) s += (int)A[i * 64 + k] * (int)B[j * 64 + k];
            bad += s != C[i * 16 + j];
        }
        printf("16x16x64 layout %d: %d mismatches\n", L, bad);
    }
    int *sink; hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 256;
    auto timeit = [&](auto kern, int nacc, double ops_per_mfma, const char *name) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, 1000, sink);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, iters, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double ops = (double)blocks * 4 * iters * nacc * ops_per_mfma;
        printf("%s nacc=%d: %.2f ms  %.1f Tops  (%.1f ns per MFMA per SIMD)\n", name, nacc, ms, ops / ms / 1e9,
               ms * 1e6 / ((double)iters * nacc));
    };
    timeit(rate32<4>, 4, 2.0 * 32 * 32 * 32, "i8 32x32x32");
    timeit(rate32<15>, 15, 2.0 * 32 * 32 * 32, "i8 32x32x32");
    timeit(rate16<4>, 4, 2.0 * 16 * 16 * 64, "i8 16x16x64");
    timeit(rate16<16>, 16, 2.0 * 16 * 16 * 64, "i8 16x16x64");
    return 0;
}
