// Probe: is a float32 MFMA accumulation bit-identical to a sequential fmaf chain over k?
// (decides whether the DTW distance tiles may use the matrix cores and stay bit-exact
// against oracle/dtw.c).  hipcc --offload-arch=gfx950 -O2 tools/mfma_exact_probe.hip -o tools/mfma_exact_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* X, const float* Y, int K, float* out_x2, float* out_x1, float* out_4x4)
{
    const int l = threadIdx.x;
    f32x16 acc = {0};
    for (int k = 0; k < K; k += 2) {
        const float a = X[(l % 32) * K + k + l / 32];
        const float b = Y[(l % 32) * K + k + l / 32];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int q = 0; q < 16; ++q) {
        const int row = 8 * (q / 4) + (l / 32) * 4 + (q % 4), col = l % 32;
        out_x2[row * 32 + col] = acc[q];
    }
    // 16x16x4 variant
    f32x4 acc4 = {0};
    for (int k = 0; k < K; k += 4) {
        const float a = X[(l % 16) * K + k + l / 16];
        const float b = Y[(l % 16) * K + k + l / 16];
        acc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4, 0, 0, 0);
    }
    for (int q = 0; q < 4; ++q) out_4x4[((l / 16) * 4 + q) * 16 + l % 16] = acc4[q];
    // 32x32x1, 2 blocks: block b of A = rows of X (lanes 0-31 block 0, 32-63 block 1)
    f32x16 c1 = {0};
    f32x16 c2 = {0};
    for (int k = 0; k < K; ++k) {
        const float a = X[(l % 32) * K + k];
        const float b = Y[(l % 32) * K + k];
        typedef float f32x32 __attribute__((ext_vector_type(32)));
        f32x32 cc;
        for (int q = 0; q < 16; ++q) { cc[q] = c1[q]; cc[16 + q] = c2[q]; }
        cc = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, cc, 0, 0, 0);
        for (int q = 0; q < 16; ++q) { c1[q] = cc[q]; c2[q] = cc[16 + q]; }
    }
    for (int q = 0; q < 16; ++q) {
        const int row = 8 * (q / 4) + (l / 32) * 4 + (q % 4), col = l % 32;
        out_x1[row * 32 + col] = c1[q];     // block 0: A block 0 (lanes 0..31) x B block 0
    }
}

int main()
{
    const int K = 40;
    std::vector<float> X(32 * K), Y(32 * K);
    srand(1);
    int bad2 = 0, bad1 = 0, bad4 = 0, trials = 200;
    float *dX, *dY, *o2, *o1, *o4;
    hipMalloc(&dX, 32 * K * 4); hipMalloc(&dY, 32 * K * 4);
    hipMalloc(&o2, 4096); hipMalloc(&o1, 4096); hipMalloc(&o4, 1024);
    std::vector<float> h2(1024), h1(1024), h4(256);
    for (int t = 0; t < trials; ++t) {
        for (auto& v : X) v = (float)rand() / RAND_MAX * 2 - 1;
        for (auto& v : Y) v = (float)rand() / RAND_MAX * 2 - 1;
        hipMemcpy(dX, X.data(), 32 * K * 4, hipMemcpyHostToDevice);
        hipMemcpy(dY, Y.data(), 32 * K * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dX, dY, K, o2, o1, o4);
        hipMemcpy(h2.data(), o2, 4096, hipMemcpyDeviceToHost);
        hipMemcpy(h1.data(), o1, 4096, hipMemcpyDeviceToHost);
        hipMemcpy(h4.data(), o4, 1024, hipMemcpyDeviceToHost);
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                float c = 0.f;
                for (int k = 0; k < K; ++k) c = fmaf(X[i * K + k], Y[j * K + k], c);
                if (memcmp(&c, &h2[i * 32 + j], 4)) ++bad2;
                if (memcmp(&c, &h1[i * 32 + j], 4)) ++bad1;
                if (i < 16 && j < 16 && memcmp(&c, &h4[i * 16 + j], 4)) ++bad4;
            }
    }
    printf("cells %d  mismatches vs fmaf chain: 32x32x2=%d  32x32x1(block0)=%d  16x16x4=%d (of %d)\n", trials * 1024, bad2, bad1,
           bad4, trials * 256);
    return 0;
}
