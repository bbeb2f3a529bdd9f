// mfma_probe.hip -- measured-achievable fp32 MFMA rate and shader clock on the
// box (SURVEY.md 8d asks for measured peaks next to vendor peaks).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/mfma_probe.hip && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void probe(float* out, int iters, unsigned long long* stamps)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f - threadIdx.x * 1e-3f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
}

template <int NACC>
void run(int blocks, int iters, const char* name)
{
    float* out; unsigned long long* st;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&st, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, st);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
        double flop = (double)blocks * 4 * iters * NACC * 4096.0;
        printf("%s blocks=%d iters=%d: %.3f ms  %.1f TFLOP/s  clock %.0f MHz  (%.1f cycles per MFMA per wave)\n", name, blocks,
               iters, ms, flop / ms / 1e9, (double)h[0] / (double)h[1] * 100.0, (double)h[0] / ((double)iters * NACC));
    }
}

int main()
{
    run<4>(256, 20000, "4acc 1wave/SIMD short");
    run<4>(256, 400000, "4acc 1wave/SIMD long ");
    run<2>(512, 200000, "2acc 2waves/SIMD     ");
    run<1>(1024, 200000, "1acc 4waves/SIMD     ");
    return 0;
}
