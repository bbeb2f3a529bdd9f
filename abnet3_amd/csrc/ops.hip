// ops.hip -- the small HBM-bound kernels around the tower: fused optimizer
// step over the flat parameter buffer, row gathers, frame stacking.
#include "common.h"

namespace abn {

// torch.optim single-tensor update rules (abnet3/trainer.py:68-87 picks the
// class, torch supplies the defaults); one flat buffer, one launch.
__global__ void optimizer_kernel(int kind, float* __restrict__ p, const float* __restrict__ g,
                                 float* __restrict__ s1, float* __restrict__ s2, int64_t n, float lr,
                                 float hp0, float hp1, float eps, int first, float bc1, float bc2_sqrt,
                                 float gscale)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * gscale;
        float pi = p[i];
        switch (kind) {
            case ABN_OPT_SGD: {          // buf = g (first step) | mu*buf + g ; p -= lr*buf
                const float buf = first ? gi : hp0 * s1[i] + gi;
                s1[i] = buf;
                pi -= lr * buf;
            } break;
            case ABN_OPT_ADADELTA: {     // rho = hp0
                const float sq = hp0 * s1[i] + (1.0f - hp0) * gi * gi;
                const float delta = sqrtf(s2[i] + eps) / sqrtf(sq + eps) * gi;
                s1[i] = sq;
                s2[i] = hp0 * s2[i] + (1.0f - hp0) * delta * delta;
                pi -= lr * delta;
            } break;
            case ABN_OPT_ADAM: {         // beta1 = hp0, beta2 = hp1
                const float m = s1[i] + (gi - s1[i]) * (1.0f - hp0);      // lerp_
                const float v = hp1 * s2[i] + (1.0f - hp1) * gi * gi;
                s1[i] = m;
                s2[i] = v;
                const float denom = sqrtf(v) / bc2_sqrt + eps;
                pi -= (lr / bc1) * (m / denom);
            } break;
            case ABN_OPT_ADAGRAD: {
                const float s = s1[i] + gi * gi;
                s1[i] = s;
                pi -= lr * (gi / (sqrtf(s) + eps));
            } break;
            default: {                   // RMSprop, alpha = hp0
                const float sq = hp0 * s1[i] + (1.0f - hp0) * gi * gi;
                s1[i] = sq;
                pi -= lr * (gi / (sqrtf(sq) + eps));
            } break;
        }
        p[i] = pi;
    }
}

// out[i][:] = table[idx[i]][:]  (dataloader.py:204-205 feat[path, :])
__global__ void gather_rows_kernel(const float* __restrict__ table, const int64_t* __restrict__ idx, int64_t n,
                                   int D, float* __restrict__ out)
{
    const int64_t total = n * D;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / D;
        const int c = (int)(i - r * D);
        out[i] = table[idx[r] * D + c];
    }
}

// features.py:135-159: out[t][k*D + c] = in[t + k - nframes/2][c], zero outside
__global__ void stack_frames_kernel(const float* __restrict__ in, int64_t T, int D, int nframes,
                                    float* __restrict__ out)
{
    const int W = D * nframes, h = nframes / 2;
    const int64_t total = T * W;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = i / W;
        const int j = (int)(i - t * W);
        const int k = j / D, c = j - k * D;
        const int64_t src = t + k - h;
        out[i] = (src >= 0 && src < T) ? in[src * D + c] : 0.0f;
    }
}

static inline int grid_for(int64_t n) { int64_t g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }

}  // namespace abn

using namespace abn;

extern "C" {

int abn_optimizer_step(int kind, float* params, const float* grads, float* state1, float* state2, int64_t n,
                       float lr, float hp0, float hp1, float eps, int64_t step, float grad_scale, void* stream)
{
    ABN_REQUIRE(kind >= ABN_OPT_SGD && kind <= ABN_OPT_RMSPROP, "optimizer_step: unknown optimizer %d", kind);
    ABN_REQUIRE(params && grads && state1, "optimizer_step: null pointer");
    ABN_REQUIRE(state2 || (kind != ABN_OPT_ADADELTA && kind != ABN_OPT_ADAM), "optimizer_step: state2 required");
    ABN_REQUIRE(n >= 0 && step >= 1, "optimizer_step: bad n/step");
    if (n == 0) return ABN_OK;
    float bc1 = 1.0f, bc2s = 1.0f;
    if (kind == ABN_OPT_ADAM) {
        bc1 = (float)(1.0 - pow((double)hp0, (double)step));
        bc2s = (float)sqrt(1.0 - pow((double)hp1, (double)step));
    }
    hipLaunchKernelGGL(optimizer_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, kind, params, grads,
                       state1, state2, n, lr, hp0, hp1, eps, step == 1 ? 1 : 0, bc1, bc2s, grad_scale);
    ABN_CHECK_LAUNCH("optimizer_step");
    return ABN_OK;
}

int abn_gather_rows(const float* table, const int64_t* idx, int64_t n, int64_t D, float* out, void* stream)
{
    ABN_REQUIRE(n >= 0 && D >= 1 && D < (1 << 24), "gather_rows: bad shape");
    if (n == 0) return ABN_OK;
    ABN_REQUIRE(table && idx && out, "gather_rows: null pointer");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(n * D)), dim3(256), 0, (hipStream_t)stream, table, idx, n,
                       (int)D, out);
    ABN_CHECK_LAUNCH("gather_rows");
    return ABN_OK;
}

int abn_stack_frames(const float* feats, int64_t T, int64_t D, int32_t nframes, float* out, void* stream)
{
    ABN_REQUIRE(nframes >= 1 && nframes % 2 == 1, "stack_frames: number of stacked frames must be odd");
    ABN_REQUIRE(T >= 0 && D >= 1 && D < (1 << 20), "stack_frames: bad shape");
    if (T == 0) return ABN_OK;
    ABN_REQUIRE(feats && out, "stack_frames: null pointer");
    hipLaunchKernelGGL(stack_frames_kernel, dim3(grid_for(T * D * nframes)), dim3(256), 0, (hipStream_t)stream, feats, T,
                       (int)D, (int)nframes, out);
    ABN_CHECK_LAUNCH("stack_frames");
    return ABN_OK;
}

}  // extern "C"
