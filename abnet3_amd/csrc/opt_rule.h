// opt_rule.h -- torch.optim single-tensor update rules (abnet3/trainer.py:68-87 picks the class,
// torch supplies the defaults), one element at a time: shared by optimizer_kernel (ops.hip) and by
// the fused slab-reduction + step kernel (tower.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/abnet3_hip.h"

namespace abn {

struct OptP {
    int kind;
    float lr, hp0, hp1, eps;
    int first;              // step == 1 (SGD's momentum buffer starts as the gradient)
    float bc1, bc2_sqrt;    // Adam's bias corrections
    float gscale;           // multiplies the gradient first (1 / world_size for a mean loss under DP)
};

// one element: parameter pi, gradient g, the element's two state values (s2 unused by most rules)
__device__ __forceinline__ float opt_update_reg(const OptP& o, float pi, float g, float& s1, float& s2)
{
    const float gi = g * o.gscale;
    switch (o.kind) {
        case ABN_OPT_SGD: {          // buf = g (first step) | mu*buf + g ; p -= lr*buf
            const float buf = o.first ? gi : o.hp0 * s1 + gi;
            s1 = buf;
            pi -= o.lr * buf;
        } break;
        case ABN_OPT_ADADELTA: {     // rho = hp0
            const float sq = o.hp0 * s1 + (1.0f - o.hp0) * gi * gi;
            const float delta = sqrtf(s2 + o.eps) / sqrtf(sq + o.eps) * gi;
            s1 = sq;
            s2 = o.hp0 * s2 + (1.0f - o.hp0) * delta * delta;
            pi -= o.lr * delta;
        } break;
        case ABN_OPT_ADAM: {         // beta1 = hp0, beta2 = hp1
            const float m = s1 + (gi - s1) * (1.0f - o.hp0);      // lerp_
            const float v = o.hp1 * s2 + (1.0f - o.hp1) * gi * gi;
            s1 = m;
            s2 = v;
            const float denom = sqrtf(v) / o.bc2_sqrt + o.eps;
            pi -= (o.lr / o.bc1) * (m / denom);
        } break;
        case ABN_OPT_ADAGRAD: {
            const float s = s1 + gi * gi;
            s1 = s;
            pi -= o.lr * (gi / (sqrtf(s) + o.eps));
        } break;
        default: {                   // RMSprop, alpha = hp0
            const float sq = o.hp0 * s1 + (1.0f - o.hp0) * gi * gi;
            s1 = sq;
            pi -= o.lr * (gi / (sqrtf(sq) + o.eps));
        } break;
    }
    return pi;
}

__device__ __forceinline__ bool opt_uses_s2(const OptP& o) { return o.kind == ABN_OPT_ADADELTA || o.kind == ABN_OPT_ADAM; }

__device__ __forceinline__ float opt_update(const OptP& o, float pi, float g, float* __restrict__ s1, float* __restrict__ s2,
                                            int64_t i)
{
    float a = s1[i], b = opt_uses_s2(o) ? s2[i] : 0.0f;
    pi = opt_update_reg(o, pi, g, a, b);
    s1[i] = a;
    if (opt_uses_s2(o)) s2[i] = b;
    return pi;
}

static inline OptP make_optp(int kind, float lr, float hp0, float hp1, float eps, int64_t step, float grad_scale)
{
    OptP o;
    o.kind = kind; o.lr = lr; o.hp0 = hp0; o.hp1 = hp1; o.eps = eps;
    o.first = step == 1 ? 1 : 0;
    o.bc1 = 1.0f; o.bc2_sqrt = 1.0f;
    if (kind == ABN_OPT_ADAM) {
        o.bc1 = (float)(1.0 - pow((double)hp0, (double)step));
        o.bc2_sqrt = (float)sqrt(1.0 - pow((double)hp1, (double)step));
    }
    o.gscale = grad_scale;
    return o;
}

}  // namespace abn
