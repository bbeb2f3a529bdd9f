// ops.hip -- the small HBM-bound kernels around the tower: fused optimizer
// step over the flat parameter buffer, row gathers, frame stacking.
#include <math.h>

#include "common.h"
#include "opt_rule.h"

namespace abn {

// one flat buffer, one launch (the update rules: opt_rule.h)
__global__ void optimizer_kernel(OptP o, float* __restrict__ p, const float* __restrict__ g,
                                 float* __restrict__ s1, float* __restrict__ s2, int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        p[i] = opt_update(o, p[i], g[i], s1, s2, i);
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

// The same for a batch of utterances laid end to end (utt_foff: cumulative frame counts, [n_utts + 1]):
// the window never crosses an utterance boundary.
__global__ void stack_frames_batched_kernel(const float* __restrict__ in, const int64_t* __restrict__ utt_foff, int n_utts,
                                            int64_t T, int D, int nframes, float* __restrict__ out)
{
    const int W = D * nframes, h = nframes / 2;
    const int64_t total = T * W;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = i / W;
        const int j = (int)(i - t * W);
        const int k = j / D, c = j - k * D;
        int lo = 0, hi = n_utts - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (utt_foff[mid] <= t) lo = mid; else hi = mid - 1;
        }
        const int64_t src = t + k - h;
        out[i] = (src >= utt_foff[lo] && src < utt_foff[lo + 1]) ? in[src * D + c] : 0.0f;
    }
}

// One training batch of frame pairs straight into the layout a (captured) train step reads:
//   x12[r]         = table[idx1[first + r]]   r < n          (tower 1)
//   x12[n_pad + r] = table[idx2[first + r]]   r < n          (tower 2)
// rows n .. n_pad - 1 of both halves zero; labels copied (padding 0); *n_valid = n.
// One 16-byte piece per thread (D % 4 == 0) or one element.
template <bool VEC>
__global__ void gather_pairs_kernel(const float* __restrict__ table, int64_t table_rows, int D, const int64_t* __restrict__ idx1,
                                    const int64_t* __restrict__ idx2, int64_t first, int n, int n_pad,
                                    const char* __restrict__ labels, int label_bytes, float* __restrict__ x12,
                                    char* __restrict__ y_out, int32_t* __restrict__ n_valid)
{
    const int per_row = VEC ? D / 4 : D;
    const int64_t total = (int64_t)2 * n_pad * per_row;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / per_row);
        const int c = (int)(i - (int64_t)row * per_row);
        const int tower = row >= n_pad, r = row - tower * n_pad;
        if (VEC) {
            float4 v = {0.f, 0.f, 0.f, 0.f};
            if (r < n) {
                const int64_t src = (tower ? idx2 : idx1)[first + r];
                if ((uint64_t)src < (uint64_t)table_rows) v = reinterpret_cast<const float4*>(table + src * D)[c];      // (a row outside the table reads as zeros)
            }
            reinterpret_cast<float4*>(x12 + (int64_t)row * D)[c] = v;
        } else {
            const int64_t src = r < n ? (tower ? idx2 : idx1)[first + r] : -1;
            x12[(int64_t)row * D + c] = (uint64_t)src < (uint64_t)table_rows ? table[src * D + c] : 0.0f;
        }
    }
    const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (labels && y_out)
        for (int64_t i = tid; i < (int64_t)n_pad * label_bytes; i += (int64_t)gridDim.x * blockDim.x)
            y_out[i] = i < (int64_t)n * label_bytes ? labels[first * label_bytes + i] : 0;
    if (n_valid && tid == 0) *n_valid = n;
}

// Mean / variance normalisation, abnet3/features.py:205-244 and :263-297:
//   mean = np.mean(features, axis), std = np.std(features, axis)   (axis 0 = per
//   channel, None = whole spectrum), out = (x - mean) / (std + eps).
// Stage 1: fp64 partial sums of x and x^2 per (row chunk, column).
__global__ __launch_bounds__(256) void mvn_partial_kernel(const float* __restrict__ x, int64_t T, int D, int64_t rows_per_chunk,
                                                          double* __restrict__ part)   // [chunks][D][2]
{
    __shared__ double s1[8][33], s2[8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t r1 = r0 + rows_per_chunk < T ? r0 + rows_per_chunk : T;
    double a = 0.0, b = 0.0;
    if (c < D)
        for (int64_t r = r0 + ty; r < r1; r += 8) {
            const double v = x[r * D + c];
            a += v;
            b += v * v;
        }
    s1[ty][tx] = a;
    s2[ty][tx] = b;
    __syncthreads();
    if (ty == 0 && c < D) {
        for (int k = 1; k < 8; ++k) { a += s1[k][tx]; b += s2[k][tx]; }
        part[((int64_t)blockIdx.y * D + c) * 2 + 0] = a;
        part[((int64_t)blockIdx.y * D + c) * 2 + 1] = b;
    }
}

// Stage 2 (one block): fixed-order reduction over chunks (and over columns for
// the whole-spectrum mode); writes mean/std as fp32 [D] (per channel) or [1].
__global__ __launch_bounds__(256) void mvn_finalize_kernel(const double* __restrict__ part, int chunks, int D, int64_t T,
                                                           int per_channel, float* __restrict__ mean, float* __restrict__ stdv)
{
    __shared__ double cs1[256], cs2[256];
    double t1 = 0.0, t2 = 0.0;
    for (int c = threadIdx.x; c < D; c += 256) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < chunks; ++k) { a += part[((int64_t)k * D + c) * 2]; b += part[((int64_t)k * D + c) * 2 + 1]; }
        if (per_channel) {
            const double m = a / (double)T;
            double var = b / (double)T - m * m;
            if (var < 0.0) var = 0.0;
            mean[c] = (float)m;
            stdv[c] = (float)sqrt(var);
        }
        t1 += a;
        t2 += b;
    }
    if (per_channel) return;
    cs1[threadIdx.x] = t1;
    cs2[threadIdx.x] = t2;
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < 256; ++k) { a += cs1[k]; b += cs2[k]; }
        const double n = (double)T * (double)D, m = a / n;
        double var = b / n - m * m;
        if (var < 0.0) var = 0.0;
        mean[0] = (float)m;
        stdv[0] = (float)sqrt(var);
    }
}

__global__ void mvn_apply_kernel(const float* __restrict__ x, int64_t n, int D, const float* __restrict__ mean,
                                 const float* __restrict__ stdv, int per_channel, float eps, float* __restrict__ out)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = per_channel ? (int)(i % D) : 0;
        out[i] = (x[i] - mean[c]) / (stdv[c] + eps);
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
    hipLaunchKernelGGL(optimizer_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                       make_optp(kind, lr, hp0, hp1, eps, step, grad_scale), params, grads, state1, state2, n);
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

int abn_stack_frames_batched(const float* feats, const int64_t* utt_frame_off, int64_t n_utts, int64_t T, int64_t D,
                             int32_t nframes, float* out, void* stream)
{
    ABN_REQUIRE(nframes >= 1 && nframes % 2 == 1, "stack_frames: number of stacked frames must be odd");
    ABN_REQUIRE(T >= 0 && D >= 1 && D < (1 << 20) && n_utts >= 1 && n_utts < (1LL << 30), "stack_frames_batched: bad shape");
    if (T == 0) return ABN_OK;
    ABN_REQUIRE(feats && out && utt_frame_off, "stack_frames_batched: null pointer");
    hipLaunchKernelGGL(stack_frames_batched_kernel, dim3(grid_for(T * D * nframes)), dim3(256), 0, (hipStream_t)stream, feats,
                       utt_frame_off, (int)n_utts, T, (int)D, (int)nframes, out);
    ABN_CHECK_LAUNCH("stack_frames_batched");
    return ABN_OK;
}

int abn_gather_pairs(const float* table, int64_t table_rows, int64_t D, const int64_t* idx1, const int64_t* idx2, int64_t first, int64_t n,
                     int64_t n_pad, const void* labels, int32_t label_bytes, float* x12, void* y_out, int32_t* n_valid,
                     void* stream)
{
    ABN_REQUIRE(D >= 1 && D < (1 << 20) && table_rows >= 0 && first >= 0 && n >= 0 && n_pad >= n && n_pad < (1 << 24), "gather_pairs: bad shape");
    ABN_REQUIRE((labels == nullptr) == (y_out == nullptr) && (!labels || (label_bytes >= 1 && label_bytes <= 8)), "gather_pairs: labels / y_out / label_bytes");
    if (n_pad == 0) return ABN_OK;
    ABN_REQUIRE(table && idx1 && idx2 && x12, "gather_pairs: null pointer");
    const bool vec = D % 4 == 0 && aligned16(table) && aligned16(x12);
    const int64_t work = 2 * n_pad * (vec ? D / 4 : D);
    if (vec) hipLaunchKernelGGL(gather_pairs_kernel<true>, dim3(grid_for(work)), dim3(256), 0, (hipStream_t)stream, table, table_rows, (int)D, idx1, idx2,
                                first, (int)n, (int)n_pad, (const char*)labels, (int)label_bytes, x12, (char*)y_out, n_valid);
    else hipLaunchKernelGGL(gather_pairs_kernel<false>, dim3(grid_for(work)), dim3(256), 0, (hipStream_t)stream, table, table_rows, (int)D, idx1, idx2,
                            first, (int)n, (int)n_pad, (const char*)labels, (int)label_bytes, x12, (char*)y_out, n_valid);
    ABN_CHECK_LAUNCH("gather_pairs");
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

namespace abn {

// last_non_linearity = 'softmax' (abnet3/model.py:161-166: nn.Softmax() on a 2-D
// input = softmax over each row).  One wavefront per row; max-shifted like ATen.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ z, int64_t rows, int n,
                                                           float* __restrict__ out)
{
    const int64_t row = blockIdx.x * 4LL + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* zr = z + row * n;
    float m = -INFINITY;
    for (int c = lane; c < n; c += 64) m = fmaxf(m, zr[c]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float s = 0.0f;
    for (int c = lane; c < n; c += 64) s += expf(zr[c] - m);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    float* orow = out + row * n;
    for (int c = lane; c < n; c += 64) orow[c] = expf(zr[c] - m) / s;
}

// dz = a * (da - sum_c(da * a)) per row
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ a, const float* __restrict__ da,
                                                               int64_t rows, int n, float* __restrict__ dz)
{
    const int64_t row = blockIdx.x * 4LL + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* ar = a + row * n;
    const float* dr = da + row * n;
    double s = 0.0;
    for (int c = lane; c < n; c += 64) s += (double)dr[c] * (double)ar[c];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float sf = (float)s;
    float* zr = dz + row * n;
    for (int c = lane; c < n; c += 64) zr[c] = ar[c] * (dr[c] - sf);
}

}  // namespace abn

extern "C" {

int abn_softmax_rows(const float* z, int64_t rows, int64_t n, float* out, void* stream)
{
    ABN_REQUIRE(rows >= 0 && n >= 1 && n < (1 << 24), "softmax_rows: bad shape");
    if (rows == 0) return ABN_OK;
    ABN_REQUIRE(z && out, "softmax_rows: null pointer");
    hipLaunchKernelGGL(abn::softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, z, rows,
                       (int)n, out);
    ABN_CHECK_LAUNCH("softmax_rows");
    return ABN_OK;
}

int abn_softmax_rows_backward(const float* a, const float* da, int64_t rows, int64_t n, float* dz, void* stream)
{
    ABN_REQUIRE(rows >= 0 && n >= 1 && n < (1 << 24), "softmax_rows_backward: bad shape");
    if (rows == 0) return ABN_OK;
    ABN_REQUIRE(a && da && dz, "softmax_rows_backward: null pointer");
    hipLaunchKernelGGL(abn::softmax_rows_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a,
                       da, rows, (int)n, dz);
    ABN_CHECK_LAUNCH("softmax_rows_backward");
    return ABN_OK;
}

}  // extern "C"

extern "C" {

int64_t abn_mvn_ws_bytes(int64_t T, int64_t D)
{
    if (T < 0 || D < 1) return -1;
    const int64_t chunks = T / 2048 < 1 ? 1 : (T / 2048 > 512 ? 512 : T / 2048);
    return chunks * D * 2 * (int64_t)sizeof(double);
}

int abn_mvn_stats(const float* feats, int64_t T, int64_t D, int per_channel, float* mean, float* stdv, void* ws,
                  void* stream)
{
    ABN_REQUIRE(feats && mean && stdv && ws, "mvn_stats: null pointer");
    ABN_REQUIRE(T >= 1 && D >= 1 && D < (1 << 20), "mvn_stats: bad shape T=%lld D=%lld", (long long)T, (long long)D);
    const int64_t chunks = T / 2048 < 1 ? 1 : (T / 2048 > 512 ? 512 : T / 2048);
    const int64_t rpc = (T + chunks - 1) / chunks;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(mvn_partial_kernel, dim3((unsigned)((D + 31) / 32), (unsigned)chunks), dim3(256), 0, st, feats, T,
                       (int)D, rpc, (double*)ws);
    hipLaunchKernelGGL(mvn_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, (int)chunks, (int)D, T,
                       per_channel, mean, stdv);
    ABN_CHECK_LAUNCH("mvn_stats");
    return ABN_OK;
}

int abn_mvn_apply(const float* feats, int64_t T, int64_t D, const float* mean, const float* stdv, int per_channel,
                  float eps, float* out, void* stream)
{
    ABN_REQUIRE(T >= 0 && D >= 1, "mvn_apply: bad shape");
    if (T == 0) return ABN_OK;
    ABN_REQUIRE(feats && mean && stdv && out, "mvn_apply: null pointer");
    hipLaunchKernelGGL(mvn_apply_kernel, dim3(grid_for(T * D)), dim3(256), 0, (hipStream_t)stream, feats, T * D, (int)D,
                       mean, stdv, per_channel, eps, out);
    ABN_CHECK_LAUNCH("mvn_apply");
    return ABN_OK;
}

}  // extern "C"
