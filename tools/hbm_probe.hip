// STREAM-style probe: achievable HBM bandwidth on this box (read, write, copy), to put
// next to the 8 TB/s vendor figure the rooflines divide by.
// hipcc --offload-arch=gfx950 -O3 tools/hbm_probe.hip -o tools/hbm_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void k_read(const float4* __restrict__ a, int64_t n, float* out)
{
    float4 s = make_float4(0, 0, 0, 0);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = a[i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = s.x;
}
__global__ void k_write(float4* __restrict__ a, int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        a[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void k_copy(const float4* __restrict__ a, float4* __restrict__ b, int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main()
{
    const int64_t bytes = 4LL << 30, n = bytes / 16;
    float4 *a, *b; float* o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const dim3 grid(256 * 16), block(256);
    for (int which = 0; which < 3; ++which) {
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k_read, grid, block, 0, 0, a, n, o);
            if (which == 1) hipLaunchKernelGGL(k_write, grid, block, 0, 0, a, n);
            if (which == 2) hipLaunchKernelGGL(k_copy, grid, block, 0, 0, a, b, n);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        const double moved = which == 2 ? 2.0 * bytes : 1.0 * bytes;
        printf("%s 4 GiB: %.3f ms  %.2f TB/s\n", which == 0 ? "read " : which == 1 ? "write" : "copy ", best, moved / best / 1e9);
    }
    return 0;
}
