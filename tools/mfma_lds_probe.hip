// mfma_lds_probe.hip -- isolates the GEMM k-loop: LDS fragment reads + 32 fp32
// MFMAs (+ barrier) per tile, 2 workgroups per CU, no global traffic.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: reads + mfma + barrier, 1: no barrier, 2: no LDS reads (regs), 3: reads+barrier, no mfma
__global__ __launch_bounds__(256) void probe(float* out, int tiles)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    for (int i = threadIdx.x; i < 13824; i += 256) smem[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 32;
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const float* As = smem; const float* Bs = smem + 9216;
    f32x4 ka = {1.f, 2.f, 3.f, 4.f};
    for (int t = 0; t < tiles; ++t) {
        const float* as = As + (t & 1) * 4608; const float* bs = Bs + (t & 1) * 2304;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 a0, a1, b0;
            if (MODE == 2) { a0 = ka; a1 = ka; b0 = ka; }
            else {
                const int r = lane & 31, h = lane >> 5;
                a0 = *reinterpret_cast<const f32x4*>(as + (wm0 + r) * 36 + 8 * g + 4 * h);
                a1 = *reinterpret_cast<const f32x4*>(as + (wm0 + 32 + r) * 36 + 8 * g + 4 * h);
                b0 = *reinterpret_cast<const f32x4*>(bs + (wn0 + r) * 36 + 8 * g + 4 * h);
            }
            if (MODE != 3) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b0[e], acc1, 0, 0, 0);
                }
            } else { acc0[0] += a0[0] + a1[1] + b0[2]; }
        }
        if (MODE != 1) __syncthreads();
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int blocks, int tiles)
{
    float* out; hipMalloc(&out, blocks * 256 * 4);
    hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 55296);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 55296, 0, out, tiles);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double mf = (double)blocks * 4 * tiles * 32;
        printf("%-34s blocks=%d tiles=%d: %.3f ms -> %.1f TFLOP/s-equivalent, %.0f ns per tile\n", name, blocks, tiles, ms,
               MODE == 3 ? 0.0 : mf * 4096 / ms / 1e9, ms * 1e6 / tiles);
    }
    hipFree(out);
}

int main()
{
    run<0>("reads + mfma + barrier", 512, 16000);
    run<1>("reads + mfma, no barrier", 512, 16000);
    run<2>("mfma from registers + barrier", 512, 16000);
    run<3>("reads + barrier, no mfma", 512, 16000);
    run<0>("reads + mfma + barrier 1 WG/CU", 256, 16000);
    return 0;
}
