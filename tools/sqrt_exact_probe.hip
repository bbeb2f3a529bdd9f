// Probe: which cheap float32 sqrt sequences are correctly rounded for EVERY float in
// [2^-24, 1] (the domain of sqrt(1 - |c|) in the DTW angular distance)?  Exhaustive.
// Also: does a float32 MFMA treat denormal products like fmaf does?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ float cand_a(float x)    // rsq + one Markstein step
{
    const float r = __builtin_amdgcn_rsqf(x);
    const float s = x * r, h = 0.5f * r;
    const float e = fmaf(-s, s, x);
    return fmaf(e, h, s);
}
__device__ float cand_b(float x)    // v_sqrt_f32 + two-sided one-ulp fix-up
{
    float s = __builtin_amdgcn_sqrtf(x);
    const float su = __int_as_float(__float_as_int(s) + 1), sd = __int_as_float(__float_as_int(s) - 1);
    const float eu = fmaf(-su, s, x), ed = fmaf(-sd, s, x);
    s = ed <= 0.0f ? sd : s;
    s = eu > 0.0f ? su : s;
    return s;
}
__device__ float cand_c(float x)    // v_sqrt_f32 + one Newton correction with rcp
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float e = fmaf(-s, s, x);
    const float h = 0.5f * __builtin_amdgcn_rcpf(s);
    return fmaf(e, h, s);
}
__global__ void sweep(uint32_t lo, uint32_t hi, unsigned long long* bad)
{
    unsigned long long a = 0, b = 0, c = 0, d = 0;
    for (uint64_t u = lo + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; u <= hi; u += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __int_as_float((uint32_t)u);
        const float ref = (float)sqrt((double)x);
        a += cand_a(x) != ref;
        b += cand_b(x) != ref;
        c += cand_c(x) != ref;
        d += sqrtf(x) != ref;
    }
    if (a) atomicAdd(&bad[0], a);
    if (b) atomicAdd(&bad[1], b);
    if (c) atomicAdd(&bad[2], c);
    if (d) atomicAdd(&bad[3], d);
}
__global__ void denorm(float* out)
{
    const int l = threadIdx.x;
    f32x16 acc = {0};
    // products 1e-20 * 1e-20 = 1e-40 are denormal; accumulate 4 of them
    for (int s = 0; s < 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(1e-20f * (1 + l % 7), 1e-20f, acc, 0, 0, 0);
    float c = 0.f;
    for (int k = 0; k < 4; ++k) c = fmaf(1e-20f * (1 + l % 7), 1e-20f, c);   // A row value depends on l%32... same lane row only for col=any
    out[l] = acc[0];
    out[64 + l] = c;
}
int main()
{
    unsigned long long* bad;
    hipMalloc(&bad, 32);
    hipMemset(bad, 0, 32);
    uint32_t lo, hi;
    float flo = ldexpf(1.0f, -24), fhi = 1.0f;
    memcpy(&lo, &flo, 4); memcpy(&hi, &fhi, 4);
    hipLaunchKernelGGL(sweep, dim3(4096), dim3(256), 0, 0, lo, hi, bad);
    unsigned long long h[4];
    hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost);
    printf("floats swept %u  mismatches: rsq+markstein=%llu  sqrt+fixup=%llu  sqrt+newton=%llu  sqrtf()=%llu\n", hi - lo + 1, h[0], h[1], h[2], h[3]);
    float* o; hipMalloc(&o, 512); float ho[128];
    hipLaunchKernelGGL(denorm, dim3(1), dim3(64), 0, 0, o);
    hipMemcpy(ho, o, 512, hipMemcpyDeviceToHost);
    // acc[0] of lane l = C[row 4*(l/32)][col l%32] = sum_k A[row][k] B[k][col]; A row 0 -> lane 0 value (1), row 4 -> lane 4 (5)
    printf("denormal accumulation: mfma lane0 %.9g  fmaf(lane0 operands) %.9g ; mfma lane32 %.9g expect %.9g\n", ho[0], ho[64], ho[32], ho[64 + 4]);
    return 0;
}
