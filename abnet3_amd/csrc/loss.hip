// loss.hip -- coscos2 / cosmargin pair loss, forward fused with backward.
//
// Replaces abnet3/loss.py:46-67 (coscos2.forward) and :85-105
// (cosmargin.forward) plus their autograd: nn.CosineSimilarity(dim=1, eps=1e-6),
// the masked per-label transform, the sum and the optional /B.
//
// HBM-bound: per pair it reads e1,e2 (2*D*4 B) + a label and writes de1,de2
// (2*D*4 B).  One half-wave (32 lanes) owns a row pair, 16-byte loads, the
// three dot products are reduced with cross-lane shuffles; the second read of
// the row for the gradient is an L1/L2 hit.  All per-row arithmetic is done in
// fp64: at the Siamese initialisation cos sits in [0.99998, 0.999997] and
// d cos/d e is a difference of nearly equal terms, which fp32 evaluates with
// 1e-4 relative noise (see DESIGN.md "parity").
#include "common.h"
#include "gemm_f32.h"      // act_grad

namespace abn {

constexpr int ROWS_PER_BLOCK = 8;      // 256 threads = 8 half-waves
constexpr double COS_EPS = 1e-6;

__device__ __forceinline__ int label_code(const void* y, int dtype, int64_t i)
{
    // torch.eq(y, 1) / torch.eq(y, -1) on any dtype (loss.py:60-63)
    double v;
    switch (dtype) {
        case ABN_Y_I8: v = ((const int8_t*)y)[i]; break;
        case ABN_Y_I32: v = ((const int32_t*)y)[i]; break;
        case ABN_Y_I64: v = (double)((const int64_t*)y)[i]; break;
        case ABN_Y_F32: v = ((const float*)y)[i]; break;
        default: v = ((const double*)y)[i]; break;
    }
    return v == 1.0 ? 1 : (v == -1.0 ? -1 : 0);
}

__device__ __forceinline__ double half_wave_sum(double v)
{
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <bool VEC>
__global__ __launch_bounds__(256) void pair_loss_kernel(const float* __restrict__ e1, const float* __restrict__ e2,
                                                        const void* __restrict__ y, int y_dtype, int64_t B, int D,
                                                        int kind, double margin, double scale,
                                                        float* __restrict__ de1, float* __restrict__ de2,
                                                        int act, const float* __restrict__ mask1,
                                                        const float* __restrict__ mask2,
                                                        double* __restrict__ partial, unsigned* __restrict__ counter,
                                                        float* __restrict__ loss_out, const int* __restrict__ n_valid,
                                                        double* __restrict__ loss_accum)
{
    __shared__ double row_term[ROWS_PER_BLOCK];
    __shared__ double sh[256];
    __shared__ int is_last;
    const int sub = threadIdx.x >> 5, l = threadIdx.x & 31;
    const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + sub;
    double term = 0.0;
    // a padded batch (n_valid): only the first *n_valid pairs are real, a mean loss divides by their number
    const int64_t Bv = n_valid ? (int64_t)*n_valid : B;
    if (n_valid && scale != 1.0) scale = 1.0 / (double)(Bv > 0 ? Bv : 1);
    if (row < B && row < Bv) {
        const float* a = e1 + row * D;
        const float* b = e2 + row * D;
        double dot = 0.0, s11 = 0.0, s22 = 0.0;
        if constexpr (VEC) {
            for (int c = l; c < D / 4; c += 32) {
                const float4 u = reinterpret_cast<const float4*>(a)[c];
                const float4 v = reinterpret_cast<const float4*>(b)[c];
                dot += (double)u.x * v.x + (double)u.y * v.y + (double)u.z * v.z + (double)u.w * v.w;
                s11 += (double)u.x * u.x + (double)u.y * u.y + (double)u.z * u.z + (double)u.w * u.w;
                s22 += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
            }
        } else {
            for (int c = l; c < D; c += 32) {
                const double u = a[c], v = b[c];
                dot += u * v;
                s11 += u * u;
                s22 += v * v;
            }
        }
        dot = half_wave_sum(dot);
        s11 = half_wave_sum(s11);
        s22 = half_wave_sum(s22);
        const double n1 = sqrt(s11), n2 = sqrt(s22);
        const double c1 = n1 > COS_EPS ? n1 : COS_EPS, c2 = n2 > COS_EPS ? n2 : COS_EPS;
        const double cs = dot / (c1 * c2);
        const int code = label_code(y, y_dtype, row);
        double dcos;
        if (kind == ABN_LOSS_COSCOS2) {
            if (code == 1) { term = (1.0 - cs) * 0.5; dcos = -0.5; }
            else if (code == -1) { term = cs * cs; dcos = 2.0 * cs; }
            else { term = cs; dcos = 1.0; }              // other labels keep raw cos
        } else {
            if (code == 1) { term = 1.0 - cs; dcos = -1.0; }
            else if (code == -1) { const double h = cs - margin; term = h > 0.0 ? h : 0.0; dcos = h >= 0.0 ? 1.0 : 0.0; }
            else { term = cs; dcos = 1.0; }
        }
        if (de1) {
            // ATen clamps the norms in place under NoGradGuard: the value is
            // max(|x|,eps) but autograd still differentiates |x| (0 at |x|=0):
            //   d cos/d x = y/(c1 c2) - cos/c1 * x/|x|
            dcos *= scale;
            const double inv = dcos / (c1 * c2);
            const double k1 = n1 > 0.0 ? dcos * cs / (c1 * n1) : 0.0;
            const double k2 = n2 > 0.0 ? dcos * cs / (c2 * n2) : 0.0;
            float* g1 = de1 + row * D;
            float* g2 = de2 + row * D;
            if constexpr (VEC) {
                for (int c = l; c < D / 4; c += 32) {
                    const float4 u = reinterpret_cast<const float4*>(a)[c];
                    const float4 v = reinterpret_cast<const float4*>(b)[c];
                    float4 o1, o2;
                    o1.x = (float)(v.x * inv - u.x * k1); o2.x = (float)(u.x * inv - v.x * k2);
                    o1.y = (float)(v.y * inv - u.y * k1); o2.y = (float)(u.y * inv - v.y * k2);
                    o1.z = (float)(v.z * inv - u.z * k1); o2.z = (float)(u.z * inv - v.z * k2);
                    o1.w = (float)(v.w * inv - u.w * k1); o2.w = (float)(u.w * inv - v.w * k2);
                    if (act != ACT_NONE) {      // d loss / d z of the layer that produced e = act(z): what act_bwd_kernel computes
                        o1.x *= act_grad(u.x, act); o1.y *= act_grad(u.y, act); o1.z *= act_grad(u.z, act); o1.w *= act_grad(u.w, act);
                        o2.x *= act_grad(v.x, act); o2.y *= act_grad(v.y, act); o2.z *= act_grad(v.z, act); o2.w *= act_grad(v.w, act);
                    }
                    if (mask1) {
                        const float4 m1 = reinterpret_cast<const float4*>(mask1 + row * D)[c];
                        const float4 m2 = reinterpret_cast<const float4*>(mask2 + row * D)[c];
                        o1.x *= m1.x; o1.y *= m1.y; o1.z *= m1.z; o1.w *= m1.w;
                        o2.x *= m2.x; o2.y *= m2.y; o2.z *= m2.z; o2.w *= m2.w;
                    }
                    reinterpret_cast<float4*>(g1)[c] = o1;
                    reinterpret_cast<float4*>(g2)[c] = o2;
                }
            } else {
                for (int c = l; c < D; c += 32) {
                    const double u = a[c], v = b[c];
                    float q1 = (float)(v * inv - u * k1), q2 = (float)(u * inv - v * k2);
                    if (act != ACT_NONE) { q1 *= act_grad(a[c], act); q2 *= act_grad(b[c], act); }
                    if (mask1) { q1 *= mask1[row * D + c]; q2 *= mask2[row * D + c]; }
                    g1[c] = q1;
                    g2[c] = q2;
                }
            }
        }
    }
    else if (row < B && de1) {                          // a padded pair: no gradient
        for (int c = l; c < D; c += 32) { de1[row * D + c] = 0.0f; de2[row * D + c] = 0.0f; }
    }
    if (l == 0) row_term[sub] = term;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < ROWS_PER_BLOCK; ++i) s += row_term[i];
        // the workgroup that draws the last ticket sums all partials in a fixed order (common.h: abn_ticket_publish)
        is_last = abn_ticket_publish(&partial[blockIdx.x], s, counter, gridDim.x);
    }
    __syncthreads();
    if (!is_last) return;
    const int64_t n = gridDim.x;
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += abn_ticket_partial(&partial[i]);
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float lv = (float)(sh[0] * scale);
        *loss_out = lv;
        if (loss_accum) *loss_accum += (double)lv;          // (one thread per call, calls in stream order)
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // ready for the next call
    }
}

}  // namespace abn

using namespace abn;

extern "C" {

int64_t abn_pair_loss_ws_bytes(int64_t B)
{
    const int64_t blocks = (B + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    return 8 + (blocks < 1 ? 1 : blocks) * (int64_t)sizeof(double);       // the ticket counter, then the partial sums
}

static int pair_loss_impl(const float* e1, const float* e2, const void* y, int y_dtype, int64_t B, int64_t D, int kind,
                          float margin, int avg, float* loss_out, float* de1, float* de2, int act, const float* mask1,
                          const float* mask2, void* ws, void* stream, const int32_t* n_valid = nullptr, double* loss_accum = nullptr)
{
    ABN_REQUIRE(e1 && e2 && y && loss_out && ws, "pair_loss: null pointer");
    ABN_REQUIRE((de1 == nullptr) == (de2 == nullptr), "pair_loss: de1/de2 must both be given or both be NULL");
    ABN_REQUIRE(B >= 1 && D >= 1 && D < (1 << 24), "pair_loss: bad shape B=%lld D=%lld", (long long)B, (long long)D);
    ABN_REQUIRE(kind == ABN_LOSS_COSCOS2 || kind == ABN_LOSS_COSMARGIN, "pair_loss: unknown loss kind %d", kind);
    ABN_REQUIRE(y_dtype >= ABN_Y_I8 && y_dtype <= ABN_Y_F64, "pair_loss: unknown label dtype %d", y_dtype);
    ABN_REQUIRE(!(kind == ABN_LOSS_COSMARGIN) || (margin >= 0.0f && margin <= 1.0f), "pair_loss: margin outside [0,1]");
    ABN_REQUIRE(act >= ABN_ACT_NONE && act <= ABN_ACT_TANH, "pair_loss: unsupported activation %d", act);
    ABN_REQUIRE((mask1 == nullptr) == (mask2 == nullptr), "pair_loss: mask1/mask2 must both be given or both be NULL");
    hipStream_t st = (hipStream_t)stream;
    const int64_t blocks = (B + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    const double scale = avg ? 1.0 / (double)B : 1.0;
    const bool vec = (D % 4 == 0) && aligned16(e1) && aligned16(e2) && (!de1 || (aligned16(de1) && aligned16(de2))) &&
                     (!mask1 || (aligned16(mask1) && aligned16(mask2)));
    unsigned* counter = (unsigned*)ws;                       // first 8 bytes: the ticket counter (fixed place whatever B is)
    double* partial = (double*)((char*)ws + 8);
    if (vec)
        hipLaunchKernelGGL(pair_loss_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, e1, e2, y, y_dtype, B,
                           (int)D, kind, (double)margin, scale, de1, de2, act, mask1, mask2, partial, counter, loss_out, n_valid, loss_accum);
    else
        hipLaunchKernelGGL(pair_loss_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, e1, e2, y, y_dtype, B,
                           (int)D, kind, (double)margin, scale, de1, de2, act, mask1, mask2, partial, counter, loss_out, n_valid, loss_accum);
    ABN_CHECK_LAUNCH("pair_loss");
    return ABN_OK;
}

int abn_pair_loss(const float* e1, const float* e2, const void* y, int y_dtype, int64_t B, int64_t D, int kind,
                  float margin, int avg, float* loss_out, float* de1, float* de2, void* ws, void* stream)
{
    return pair_loss_impl(e1, e2, y, y_dtype, B, D, kind, margin, avg, loss_out, de1, de2, ABN_ACT_NONE, nullptr, nullptr, ws,
                          stream);
}

int abn_pair_loss_padded(const float* e1, const float* e2, const void* y, int y_dtype, int64_t B, int64_t D, int kind,
                         float margin, int avg, const int32_t* n_valid, float* loss_out, double* loss_accum, float* de1,
                         float* de2, void* ws, void* stream)
{
    return pair_loss_impl(e1, e2, y, y_dtype, B, D, kind, margin, avg, loss_out, de1, de2, ABN_ACT_NONE, nullptr, nullptr, ws,
                          stream, n_valid, loss_accum);
}

int abn_pair_loss_dz(const float* e1, const float* e2, const void* y, int y_dtype, int64_t B, int64_t D, int kind,
                     float margin, int avg, int act, const float* mask1, const float* mask2, float* loss_out, float* dz1,
                     float* dz2, void* ws, void* stream)
{
    ABN_REQUIRE(dz1 && dz2, "pair_loss_dz: null gradient buffers");
    return pair_loss_impl(e1, e2, y, y_dtype, B, D, kind, margin, avg, loss_out, dz1, dz2, act, mask1, mask2, ws, stream);
}

}  // extern "C"
