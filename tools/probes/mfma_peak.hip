// sustained v_mfma_f32_32x32x16_bf16 rate with every SIMD busy (what clock the chip holds under MFMA load)
//   hipcc -O3 --offload-arch=gfx950 -o tools/probes/mfma_peak tools/probes/mfma_peak.hip && tools/probes/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void k(float* out, int iters, unsigned long long* clk)
{
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(j); }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int q = 0; q < 16; ++q) s += c0[q] + c1[q] + c2[q] + c3[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
int main()
{
    float* out; unsigned long long* clk;
    hipMalloc(&out, 256 * 512 * 8 * sizeof(float));
    hipMalloc(&clk, 16);
    for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd)
    for (int iters : {2000, 20000, 200000}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int blocks = 256, threads = 256 * waves_per_simd;
        k<<<blocks, threads>>>(out, iters, clk);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<<<blocks, threads>>>(out, iters, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double mf = (double)blocks * (threads / 64) * iters * 4;
        printf("waves/SIMD %d iters %6d: %.3f ms  %.1f TFLOP/s  cycles/MFMA/SIMD (at 2.4 GHz) %.1f | s_memtime %.0f ticks/us, s_memrealtime %.0f ticks/us, memtime per MFMA per SIMD %.1f\n",
               waves_per_simd, iters, ms, mf * 32768 / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (iters * 4.0 * waves_per_simd),
               h[0] / (ms * 1e3), h[1] / (ms * 1e3), (double)h[0] / (iters * 4.0 * waves_per_simd));
    }
    return 0;
}
