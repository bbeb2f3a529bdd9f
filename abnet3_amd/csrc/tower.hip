// tower.hip -- SiameseNetwork tower forward / backward for gfx950.
//
// Replaces the torch op sequence behind abnet3/model.py:179-196 (forward_once /
// forward) and its autograd (abnet3/trainer.py:239): per layer
//   Linear (addmm) -> Dropout(p=0: identity) -> [BatchNorm1d] -> activation.
// GEMMs run on the fp32 matrix cores (gemm_f32.h); bias + activation (forward)
// and the activation derivative (dgrad) are fused into the GEMM epilogues; the
// bias gradient rides in the wgrad GEMM as an extra all-ones input column.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <math.h>
#include <atomic>

#include "common.h"
#include "gemm_f32.h"
#include "opt_rule.h"
#include "tower_fused.h"
#include "tower_planes.h"
#include "tower_bn_persist.h"
#include "tower_wide.h"
#include "tower_wgrad_step.h"

namespace abn {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// The A/B switches: one read of the environment (common.h).
static Switches read_switches()
{
    auto off = [](const char* name) { const char* v = getenv(name); return v && atoi(v) == 0; };
    Switches s;
    s.planes = !off("ABN_PLANES");
    s.fused = !off("ABN_FUSED");
    s.fused_min_rows = getenv("ABN_FUSED_MIN_ROWS") ? atoll(getenv("ABN_FUSED_MIN_ROWS")) : -1;
    s.bn_planes = !off("ABN_BN_PLANES");
    s.bn_persist = !off("ABN_BN_PERSIST");
    s.wgrad_xcd = !off("ABN_WGRAD_XCD");
    s.bf16x3_planes = !off("ABN_BF16X3_PLANES");
    s.bwd_pair = !off("ABN_BWD_PAIR");
    s.gemm_tile = getenv("ABN_GEMM_TILE") ? atoi(getenv("ABN_GEMM_TILE")) : -1;
    s.wgrad_rows_per_slab = getenv("ABN_WGRAD_ROWS_PER_SLAB") ? atoll(getenv("ABN_WGRAD_ROWS_PER_SLAB")) : -1;
    s.wide = !off("ABN_WIDE");
    s.wide_max_rows = getenv("ABN_WIDE_MAX_ROWS") ? atoll(getenv("ABN_WIDE_MAX_ROWS")) : -1;
    s.wide_max_groups = getenv("ABN_WIDE_MAXG") ? atoi(getenv("ABN_WIDE_MAXG")) : WD_MAXG;
    if (s.wide_max_groups < 1 || s.wide_max_groups > WD_MAXG) s.wide_max_groups = WD_MAXG;
    s.dtw_f40 = !off("ABN_DTW_F40");
    s.dtw_pc = !off("ABN_DTW_PC");
    s.wgrad_wgs_heavy = getenv("ABN_WGRAD_WGS_HEAVY") ? atoi(getenv("ABN_WGRAD_WGS_HEAVY")) : 128;
    s.wgrad_wgs_light = getenv("ABN_WGRAD_WGS_LIGHT") ? atoi(getenv("ABN_WGRAD_WGS_LIGHT")) : 128;
    s.wgrad_tile128 = getenv("ABN_WGRAD_TILE128") ? atoi(getenv("ABN_WGRAD_TILE128")) : -1;
    s.dtw_wgs_per_cu = getenv("ABN_DTW_WGS") ? atoi(getenv("ABN_DTW_WGS")) : 6;
    if (s.dtw_wgs_per_cu < 1 || s.dtw_wgs_per_cu > 9) s.dtw_wgs_per_cu = 6;
    s.wgrad_step = !off("ABN_WGRAD_STEP");
    s.dtw_dealt = !off("ABN_DTW_SCHED");
    s.oneshot_wgs = getenv("ABN_ONESHOT_WGS") ? atoi(getenv("ABN_ONESHOT_WGS")) : 32;
    if (s.oneshot_wgs < 1 || s.oneshot_wgs > 256) s.oneshot_wgs = 32;
    return s;
}
static Switches g_switches = read_switches();       // (at library load: no call ever reads the environment)
const Switches& switches() { return g_switches; }
void reload_switches() { g_switches = read_switches(); }

// ---------------------------------------------------------------------------
// GEMM dispatch
// ---------------------------------------------------------------------------
template <int BM, int BN, bool A_KC, bool B_KC, int EPI, bool VEC, int BF16>
static void launch_one(const GemmP& p, int splits, hipStream_t st)
{
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    constexpr size_t lds = BF16 == 3 ? gemm_lds_bytes3<BM, BN>() : gemm_lds_bytes<BM, BN, A_KC, B_KC>();
    auto k = gemm_f32_kernel<BM, BN, A_KC, B_KC, EPI, VEC, BF16>;
    static bool attr_set[16] = {};     // > 64 KiB of dynamic LDS needs the opt-in, once per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev = (dev >= 0 && dev < 16) ? dev : 0;
    if (!attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set[dev] = true;
    }
    GemmP q = p;
    q.splits = splits;
    hipLaunchKernelGGL(k, dim3(tiles * splits), dim3(256), lds, st, q);
}

// 16-byte global loads need both operands aligned with leading dimensions that
// are multiples of 4 floats; anything else takes the element-wise build.
template <int BM, int BN, bool A_KC, bool B_KC, int EPI>
static void launch_cfg(const GemmP& p, int splits, hipStream_t st)
{
    if (p.bf16 == 1) {                    // throughput mode (abn_tower_desc.precision = 1)
        if (p.a_vec && p.b_vec) launch_one<BM, BN, A_KC, B_KC, EPI, true, 1>(p, splits, st);
        else launch_one<BM, BN, A_KC, B_KC, EPI, false, 1>(p, splits, st);
        return;
    }
    if (p.bf16 == 2) {                    // bf16 x 3 (abn_tower_desc.precision = 2): fp32-grade products on the bf16 matrix cores
        const bool planes = switches().bf16x3_planes;   // A/B switch
        if (p.a_vec && p.b_vec && planes) launch_one<BM, BN, A_KC, B_KC, EPI, true, 3>(p, splits, st);   // split once per tile (LDS planes)
        else if (p.a_vec && p.b_vec) launch_one<BM, BN, A_KC, B_KC, EPI, true, 2>(p, splits, st);
        else launch_one<BM, BN, A_KC, B_KC, EPI, false, 2>(p, splits, st);
        return;
    }
    if (p.a_vec && p.b_vec) launch_one<BM, BN, A_KC, B_KC, EPI, true, 0>(p, splits, st);
    else launch_one<BM, BN, A_KC, B_KC, EPI, false, 0>(p, splits, st);
}

// Checks, epilogue vectorisation flag and tile choice of one GEMM.  Returns the tile
// code (0: 128x128, 1: 128x64, 2: 64x128, 3: 64x64), -1 for "nothing to do", or a
// negative ABN_E_* error shifted by -100.
template <bool A_KC, bool B_KC, int EPI>
static int prepare_gemm(GemmP& p, int splits)
{
    if (p.M <= 0 || p.N <= 0) return -1;
    if (p.K <= 0) { set_error("gemm: empty reduction"); return -100 + ABN_E_ARG; }
    {   // in-kernel lane offsets are 32-bit: every operand must span < 2^31 floats
        const int64_t span_a = (int64_t)(A_KC ? p.M : p.K) * p.lda, span_b = (int64_t)(B_KC ? p.N : p.K) * p.ldb;
        if (span_a >= (1LL << 31) || span_b >= (1LL << 31)) { set_error("gemm: operand larger than 2^31 floats"); return -100 + ABN_E_ARG; }
    }
    auto tiles = [&](int bm, int bn) {
        return (int64_t)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * splits;
    };
    // Largest tile that still gives every CU (256) two workgroups: a workgroup
    // is one wave per SIMD, so a second resident workgroup is what covers the
    // barrier / LDS-refill phases of the first with MFMA work.  Among the two
    // rectangular shapes prefer the one wasting less padding.
    const int64_t want = 2 * 256;
    {
        const bool slab_ok = (EPI != EPI_WGRAD) || (p.slab_stride % 4 == 0);
        p.c_vec = aligned16(p.C) && (p.ldc % 4 == 0) && slab_ok && (!p.bias || aligned16(p.bias)) &&
                  (!p.aux || (aligned16(p.aux) && p.ldaux % 4 == 0)) &&
                  (!p.mask || (aligned16(p.mask) && (EPI == EPI_DGRAD ? p.ldaux : p.ldc) % 4 == 0));
    }
    p.splits = splits;
#ifdef ABN_STAMPS
    p.stamps = getenv("ABN_STAMP_BUF") ? (unsigned long long*)strtoull(getenv("ABN_STAMP_BUF"), nullptr, 0) : nullptr;
#endif
    const int force = switches().gemm_tile;
    if (force >= 0 && force <= 3) return force;
    const int64_t pad_a = (int64_t)((p.M + 127) / 128 * 128) * ((p.N + 63) / 64 * 64);
    const int64_t pad_b = (int64_t)((p.M + 63) / 64 * 64) * ((p.N + 127) / 128 * 128);
    if (p.M > 64 && p.N > 64 && tiles(128, 128) >= want) return 0;
    if (p.M > 64 && tiles(128, 64) >= want && (pad_a <= pad_b || !(p.N > 64 && tiles(64, 128) >= want))) return 1;
    if (p.N > 64 && tiles(64, 128) >= want) return 2;
    return 3;
}

template <bool A_KC, bool B_KC, int EPI>
static int launch_gemm(GemmP p, int splits, hipStream_t st)
{
    const int tile = prepare_gemm<A_KC, B_KC, EPI>(p, splits);
    if (tile == -1) return ABN_OK;
    if (tile < -1) return tile + 100;
    if (tile == 0) launch_cfg<128, 128, A_KC, B_KC, EPI>(p, splits, st);
    else if (tile == 1) launch_cfg<128, 64, A_KC, B_KC, EPI>(p, splits, st);
    else if (tile == 2) launch_cfg<64, 128, A_KC, B_KC, EPI>(p, splits, st);
    else launch_cfg<64, 64, A_KC, B_KC, EPI>(p, splits, st);
    ABN_CHECK_LAUNCH("gemm_f32");
    return ABN_OK;
}

static bool bf16x3_planes()
{
    return switches().bf16x3_planes;
}

// wgrad + dgrad of one backward layer in ONE grid (gemm_bwd_pair_kernel) when both take
// their usual vectorised instantiations; otherwise two launches.
template <int WM, int WN, int BF16>
static void launch_pair_one(const GemmP& pw, int n0, const GemmP& pd, int n1, hipStream_t st)
{
    constexpr size_t lw = BF16 == 3 ? gemm_lds_bytes3<WM, WN>() : gemm_lds_bytes<WM, WN, false, false>();
    constexpr size_t ld = BF16 == 3 ? gemm_lds_bytes3<128, 64>() : gemm_lds_bytes<128, 64, true, false>();
    constexpr size_t lds = lw > ld ? lw : ld;
    auto k = gemm_bwd_pair_kernel<WM, WN, BF16>;
    static bool attr_set[16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev = (dev >= 0 && dev < 16) ? dev : 0;
    if (!attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(k, dim3(n0 + n1), dim3(256), lds, st, pw, n0, pd);
}

static int launch_bwd_pair(GemmP pw, int splits, GemmP pd, hipStream_t st)
{
    const bool enabled = switches().bwd_pair;
    const int tw = prepare_gemm<false, false, EPI_WGRAD>(pw, splits);
    const int td = prepare_gemm<true, false, EPI_DGRAD>(pd, 1);
    if (tw < -1) return tw + 100;
    if (td < -1) return td + 100;
    const bool vec = pw.a_vec && pw.b_vec && pd.a_vec && pd.b_vec;
    const bool same_prec = pw.bf16 == pd.bf16;
    if (enabled && vec && same_prec && td == 1 && (tw == 1 || tw == 3)) {
        const int wm = tw == 1 ? 128 : 64;
        const int n0 = ((pw.M + wm - 1) / wm) * ((pw.N + 63) / 64) * splits;
        const int n1 = ((pd.M + 127) / 128) * ((pd.N + 63) / 64);
        if (n0 % 8 == 0) {
            if (tw == 1) {
                if (pw.bf16 == 1) launch_pair_one<128, 64, 1>(pw, n0, pd, n1, st);
                else if (pw.bf16 == 2 && bf16x3_planes()) launch_pair_one<128, 64, 3>(pw, n0, pd, n1, st);
                else if (pw.bf16 == 2) launch_pair_one<128, 64, 2>(pw, n0, pd, n1, st);
                else launch_pair_one<128, 64, 0>(pw, n0, pd, n1, st);
            } else {
                if (pw.bf16 == 1) launch_pair_one<64, 64, 1>(pw, n0, pd, n1, st);
                else if (pw.bf16 == 2 && bf16x3_planes()) launch_pair_one<64, 64, 3>(pw, n0, pd, n1, st);
                else if (pw.bf16 == 2) launch_pair_one<64, 64, 2>(pw, n0, pd, n1, st);
                else launch_pair_one<64, 64, 0>(pw, n0, pd, n1, st);
            }
            ABN_CHECK_LAUNCH("gemm_bwd_pair");
            return ABN_OK;
        }
    }
    int rc = launch_gemm<false, false, EPI_WGRAD>(pw, splits, st);
    if (rc != ABN_OK) return rc;
    return launch_gemm<true, false, EPI_DGRAD>(pd, 1, st);
}

// ---------------------------------------------------------------------------
// small elementwise / column-reduction kernels
// ---------------------------------------------------------------------------
__global__ void act_bwd_kernel(const float* __restrict__ a, const float* __restrict__ da,
                               const float* __restrict__ mask, float* __restrict__ dz, int64_t n, int act)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        dz[i] = da[i] * act_grad(a[i], act) * (mask ? mask[i] : 1.0f);
}

struct ReduceTable {
    int n_layers;
    int splits[ABN_MAX_LAYERS];
    int64_t slab_stride;
    int64_t off[ABN_MAX_LAYERS];     // packed offset of [W_l | b_l]
    int64_t nW[ABN_MAX_LAYERS];
    int64_t nb[ABN_MAX_LAYERS];
    float* dW[ABN_MAX_LAYERS];
    float* db[ABN_MAX_LAYERS];
    int64_t total;
    int64_t begin;                   // first packed element this launch reduces (a layer boundary; 0: all of [0, total))
    // tensors whose gradient is already final in the flat gradient buffer (BatchNorm's gamma / beta): the fused
    // reduction + optimizer launch steps them too (float offsets into the flat buffers, element counts)
    int n_extra;
    int64_t extra_off[2 * ABN_MAX_LAYERS];
    int32_t extra_n[2 * ABN_MAX_LAYERS];
    // slab_reduce_kernel (the data-parallel step: an all-reduce follows): the failure word of the resident BatchNorm tower's
    // sync buffer and the tensors the tower's backward wrote itself (gamma / beta gradients).  Word set = the launches in
    // front of this one gave up: this rank hands the all-reduce a ZERO gradient instead of their poison.
    const unsigned* fail_word;
    int n_own;
    float* own[2 * ABN_MAX_LAYERS];
    int32_t own_n[2 * ABN_MAX_LAYERS];
};

// sum_s slab[s][0..3] in a FIXED order: four interleaved partial sums (s mod 4), combined as
// (p0 + p1) + (p2 + p3).  Sixteen slabs are loaded before any of them is added: the sum of 16-32
// slabs is then 1-2 memory round trips instead of 4-8 (the launch was latency, not bytes).
__device__ __forceinline__ f32x4 sum_slabs(const float* __restrict__ src, int S, int64_t stride)
{
    f32x4 p[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    int k = 0;
    for (; k + 15 < S; k += 16) {
        f32x4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (int64_t)(k + u) * stride);
#pragma unroll
        for (int u = 0; u < 16; ++u) p[u & 3] += v[u];
    }
    for (; k + 3 < S; k += 4) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (int64_t)(k + u) * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] += v[u];
    }
    for (; k < S; ++k) p[0] += *reinterpret_cast<const f32x4*>(src + (int64_t)k * stride);
    return (p[0] + p[1]) + (p[2] + p[3]);
}

// grad[i] = sum_s slab[s][i] in a FIXED order (deterministic, unlike atomics):
// four interleaved partial sums (s mod 4) so that four loads are in flight,
// combined as (p0 + p1) + (p2 + p3).  Layer boundaries are 64-float aligned, so
// a thread's 4 consecutive elements never straddle two layers.
__global__ void slab_reduce_kernel(const float* __restrict__ slabs, ReduceTable t)
{
    const bool dropped = t.fail_word && *t.fail_word != 0u;
    if (dropped && blockIdx.x == 0)
        for (int k = 0; k < t.n_own; ++k)
            for (int i = threadIdx.x; i < t.own_n[k]; i += blockDim.x) t.own[k][i] = 0.0f;
    const int64_t n4 = (t.total + 3) / 4;
    for (int64_t q = t.begin / 4 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n4;
         q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = 4 * q;
        int l = 0;
        while (l + 1 < t.n_layers && i >= t.off[l + 1]) ++l;
        const int S = t.splits[l];
        const f32x4 s = dropped ? f32x4{0.0f, 0.0f, 0.0f, 0.0f} : sum_slabs(slabs + i, S, t.slab_stride);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t j = i + e - t.off[l];
            if (j < t.nW[l]) t.dW[l][j] = s[e];
            else if (j - t.nW[l] < t.nb[l]) t.db[l][j - t.nW[l]] = s[e];
        }
    }
}

// The same reduction with the optimizer step applied to each element as soon as its gradient is
// known (single process, no gradient exchange in between): one launch instead of two, the
// gradient is written once (p.grad stays valid) and never read back.  Parameters, gradients and
// optimizer state share ONE flat layout: element j of layer l's dW lives at float offset
// (t.dW[l] - grads) + j of all four buffers.
// fail_word: the failure word of the resident BatchNorm tower's sync buffer (or null).  Set, the launches in front of this
// one gave up on a hand-over and their gradients are garbage: the step is DROPPED -- parameters, optimizer state and the
// gradient buffer stay as they are -- instead of carrying NaN into every parameter (the caller finds the word set, zeroes
// the buffer and goes on with the layer launches: INTEGRATION.md, "when the resident tower gives up").
__global__ void slab_reduce_step_kernel(const float* __restrict__ slabs, ReduceTable t, OptP o, float* __restrict__ params,
                                        float* __restrict__ grads, float* __restrict__ s1, float* __restrict__ s2,
                                        const unsigned* __restrict__ fail_word, int32_t* __restrict__ step_ctr)
{
    if (step_ctr && blockIdx.x == 0 && threadIdx.x == 0) *step_ctr += 1;      // abn_step_source: the step is over (every reader of its entry ran before)
    if (fail_word && *fail_word != 0u) return;
    const int64_t n4 = (t.total + 3) / 4;
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n4;
         q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = 4 * q;
        int l = 0;
        while (l + 1 < t.n_layers && i >= t.off[l + 1]) ++l;
        const int S = t.splits[l];
        const int64_t baseW = t.dW[l] - grads, baseb = t.db[l] - grads;
        const int64_t j0 = i - t.off[l];
        // the four elements inside one tensor, 16-byte aligned in the flat buffers (the usual case):
        // parameters and state come as float4, requested BEFORE the slabs are summed
        const bool inW = j0 + 3 < t.nW[l], inb = j0 >= t.nW[l] && (j0 - t.nW[l]) + 3 < t.nb[l];
        const int64_t idx0 = inW ? baseW + j0 : baseb + (j0 - t.nW[l]);
        if ((inW || inb) && (idx0 & 3) == 0) {
            f32x4 pv = *reinterpret_cast<const f32x4*>(params + idx0);
            f32x4 av = *reinterpret_cast<const f32x4*>(s1 + idx0);
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (opt_uses_s2(o)) bv = *reinterpret_cast<const f32x4*>(s2 + idx0);
            const f32x4 s = sum_slabs(slabs + i, S, t.slab_stride);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a1 = av[e], a2 = bv[e];
                pv[e] = opt_update_reg(o, pv[e], s[e], a1, a2);
                av[e] = a1; bv[e] = a2;
            }
            *reinterpret_cast<f32x4*>(grads + idx0) = s;
            *reinterpret_cast<f32x4*>(params + idx0) = pv;
            *reinterpret_cast<f32x4*>(s1 + idx0) = av;
            if (opt_uses_s2(o)) *reinterpret_cast<f32x4*>(s2 + idx0) = bv;
            continue;
        }
        const f32x4 s = sum_slabs(slabs + i, S, t.slab_stride);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t j = j0 + e;
            int64_t idx;
            if (j < t.nW[l]) idx = baseW + j;
            else if (j - t.nW[l] < t.nb[l]) idx = baseb + (j - t.nW[l]);
            else continue;
            grads[idx] = s[e];
            params[idx] = opt_update(o, params[idx], s[e], s1, s2, idx);
        }
    }
    for (int x = 0; x < t.n_extra; ++x)
        for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < t.extra_n[x]; j += (int64_t)gridDim.x * blockDim.x) {
            const int64_t idx = t.extra_off[x] + j;
            params[idx] = opt_update(o, params[idx], grads[idx], s1, s2, idx);
        }
}

// (BN_EPS, BN_MOMENTUM, BN_WG_GROUPS: tower_bn_persist.h)

// BatchNorm column reductions run in two deterministic stages over many workgroups
// (the first version used 32 workgroups for the whole matrix: 150-300 us per layer).
// Stage 1: a workgroup sums `chunk_rows` rows of one forward_once call for 64 columns
// (64 consecutive columns per row: 256-byte segments; 4 row lanes; fp64 accumulators)
// and writes its partial sums; stage 2 adds the chunks of a column in a fixed order.
constexpr int BN_MAX_CHUNKS = 128;
static inline int bn_chunks(int64_t rows_per_call)
{
    const int64_t c = (rows_per_call + 63) / 64;
    return (int)(c < 1 ? 1 : (c > BN_MAX_CHUNKS ? BN_MAX_CHUNKS : c));
}
static inline int64_t bn_chunk_rows(int64_t rows_per_call) { return (rows_per_call + bn_chunks(rows_per_call) - 1) / bn_chunks(rows_per_call); }

// partial layout: part[((g * nchunks + chunk) * 2 + which) * C + c]
template <bool BWD>
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ z /* fwd: z; bwd: da */,
                                                         const float* __restrict__ a, const float* __restrict__ xhat,
                                                         int64_t rows_per_call, int C, int64_t chunk_rows, int act,
                                                         double* __restrict__ part)
{
    __shared__ double s1[4][64], s2[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx, chunk = blockIdx.y, g = blockIdx.z;
    const int64_t r0 = chunk * chunk_rows, r1 = min(r0 + chunk_rows, rows_per_call);
    const int64_t base = (int64_t)g * rows_per_call * C;
    double u = 0.0, v = 0.0;
    if (c < C)
        for (int64_t r = r0 + ty; r < r1; r += 4) {
            const int64_t i = base + r * C + c;
            if (BWD) {     // s1 = sum dy, s2 = sum dy * xhat with dy = da * act'(a)
                const double dy = (double)(z[i] * act_grad(a[i], act));
                u += dy;
                v += dy * xhat[i];
            } else {       // s1 = sum z, s2 = sum z^2
                const double x = z[i];
                u += x;
                v += x * x;
            }
        }
    s1[ty][tx] = u;
    s2[ty][tx] = v;
    __syncthreads();
    if (ty == 0 && c < C) {
        for (int k = 1; k < 4; ++k) { u += s1[k][tx]; v += s2[k][tx]; }
        double* dst = part + ((int64_t)(g * gridDim.y + chunk) * 2) * C;
        dst[c] = u;
        dst[C + c] = v;
    }
}

// forward, one thread per column: batch mean, biased variance, 1/sqrt(var + eps) of
// every call, then the running statistics -- one momentum update per forward_once call,
// in call order (the reference updates twice per Siamese forward)
__global__ void bn_stats_finish_kernel(const double* __restrict__ part, int nchunks, int64_t rows_per_call, int C,
                                       int n_calls, float* __restrict__ mean, float* __restrict__ invstd,
                                       float* __restrict__ var_out, float* __restrict__ rm, float* __restrict__ rv)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float m_run = rm[c], v_run = rv[c];
    const float unb = rows_per_call > 1 ? (float)((double)rows_per_call / (double)(rows_per_call - 1)) : 1.0f;
    for (int g = 0; g < n_calls; ++g) {
        double a = 0.0, b = 0.0;
#pragma unroll 8
        for (int k = 0; k < nchunks; ++k) {       // fixed order; the loads are independent
            const double* src = part + ((int64_t)(g * nchunks + k) * 2) * C;
            a += src[c];
            b += src[C + c];
        }
        const double n = (double)rows_per_call;
        const double m = a / n;
        double var = b / n - m * m;
        if (var < 0.0) var = 0.0;
        const int64_t idx = (int64_t)g * C + c;
        mean[idx] = (float)m;
        var_out[idx] = (float)var;
        invstd[idx] = 1.0f / sqrtf((float)var + BN_EPS);
        m_run = (1.0f - BN_MOMENTUM) * m_run + BN_MOMENTUM * (float)m;
        v_run = (1.0f - BN_MOMENTUM) * v_run + BN_MOMENTUM * ((float)var * unb);
    }
    rm[c] = m_run;
    rv[c] = v_run;
}

// The same, from the per-workgroup statistics bn_fwd_layer_kernel leaves (tower_planes.h): 32 rows each,
// sums shifted by the workgroup's first row c -- sum z = sd + n c, sum z^2 = sq + 2 c sd + n c^2 in float64
// (n = the workgroup's rows: 32, fewer in the last one of a call).
// A block = 64 columns x 16 groups of workgroups: every group adds its share in order, thread group 0 the
// sixteen group sums in order (one thread per column walking 128 workgroups alone took 18 us).
// sums_out (cross-replica statistics, abn_tower_desc.bn_sync_world): the launch stops at [call][sum z | sum z^2][C] in
// float64 -- the caller all-reduces them and bn_stats_from_sums_kernel finishes.
__global__ __launch_bounds__(64 * BN_WG_GROUPS) void bn_stats_finish_wg_kernel(
    const float* __restrict__ part, int wgs_per_call, int64_t rows_per_call, int C, int n_calls, float* __restrict__ mean,
    float* __restrict__ invstd, float* __restrict__ var_out, float* __restrict__ rm, float* __restrict__ rv,
    double* __restrict__ sums_out, const int* __restrict__ n_valid)
{
    __shared__ double sa[BN_WG_GROUPS][64], sb[BN_WG_GROUPS][64];
    const int tx = threadIdx.x & 63, kg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    const bool ok = c < C;
    float m_run = 0.0f, v_run = 0.0f;
    if (ok && kg == 0) { m_run = rm[c]; v_run = rv[c]; }
    // (a padded batch, abn_tower_desc.n_valid: the calls end behind their real rows; the workgroups behind the end hold zero sums)
    const int64_t rows_alloc = rows_per_call;
    if (n_valid) { const int64_t nv = *n_valid; rows_per_call = nv < 1 ? 1 : (nv < rows_alloc ? nv : rows_alloc); }
    const float unb = rows_per_call > 1 ? (float)((double)rows_per_call / (double)(rows_per_call - 1)) : 1.0f;
    // the sixteen thread groups are dealt over the calls (two calls: eight groups each), so that every call's loads are in
    // flight together: one round trip for the launch instead of one per call (6.8 -> ~5 us at C2); each group adds its share
    // of a call's workgroups in order, thread group 0 the group sums of every call in order
    const int gpc = n_calls < BN_WG_GROUPS ? BN_WG_GROUPS / n_calls : 1;          // groups per call
    const int per = (wgs_per_call + gpc - 1) / gpc;
    for (int g0 = 0; g0 < n_calls; g0 += BN_WG_GROUPS / gpc) {
        const int g = g0 + kg / gpc, sub = kg % gpc;
        double a = 0.0, b = 0.0;
        if (ok && g < n_calls) {
            const int k0 = sub * per, k1 = min(k0 + per, wgs_per_call);
#pragma unroll 8
            for (int k = k0; k < k1; ++k) {           // fixed order; the loads are independent
                const float* src = part + (int64_t)(g * wgs_per_call + k) * (3 * PL_MAXW);
                const double sd = src[c], sq = src[PL_MAXW + c], cc = src[2 * PL_MAXW + c];
                const int64_t left = rows_per_call - (int64_t)k * PL_ROWS;
                const double nk = left < PL_ROWS ? (left > 0 ? (double)left : 0.0) : (double)PL_ROWS;      // rows of workgroup k
                a += sd + nk * cc;
                b += sq + 2.0 * cc * sd + nk * cc * cc;
            }
        }
        __syncthreads();                              // (the previous round's sums have been read)
        sa[kg][tx] = a;
        sb[kg][tx] = b;
        __syncthreads();
        if (ok && kg == 0) {
            for (int q = 0; q < BN_WG_GROUPS / gpc && g0 + q < n_calls; ++q) {
                const int gq = g0 + q;
                double ta = 0.0, tb = 0.0;
                for (int k = 0; k < gpc; ++k) { ta += sa[q * gpc + k][tx]; tb += sb[q * gpc + k][tx]; }
                if (sums_out) {
                    sums_out[((int64_t)gq * 2) * C + c] = ta;
                    sums_out[((int64_t)gq * 2 + 1) * C + c] = tb;
                    continue;
                }
                const double n = (double)rows_per_call;
                const double m = ta / n;
                double var = tb / n - m * m;
                if (var < 0.0) var = 0.0;
                const int64_t idx = (int64_t)gq * C + c;
                mean[idx] = (float)m;
                var_out[idx] = (float)var;
                invstd[idx] = 1.0f / sqrtf((float)var + BN_EPS);
                m_run = (1.0f - BN_MOMENTUM) * m_run + BN_MOMENTUM * (float)m;
                v_run = (1.0f - BN_MOMENTUM) * v_run + BN_MOMENTUM * ((float)var * unb);
            }
        }
    }
    if (ok && kg == 0 && !sums_out) {
        rm[c] = m_run;
        rv[c] = v_run;
    }
    // (cross-replica statistics: this replica's rows per call travel with the sums -- the replicas' batches differ in size)
    if (sums_out && blockIdx.x == 0 && (int)threadIdx.x < n_calls) sums_out[(int64_t)n_calls * 2 * C + threadIdx.x] = (double)rows_per_call;
}

// ... from the (all-reduced) sums over n_stat rows per call: mean, biased variance, invstd, the running statistics
// (one momentum update per call, in call order, unbiased variance), as above
// (n: the all-reduced row count of the call, behind the sums -- the replicas' batches may differ in size; nstat_out: the same
// as floats, for the backward's 1 / n)
__global__ void bn_stats_from_sums_kernel(const double* __restrict__ sums, int C, int n_calls, float* __restrict__ mean,
                                          float* __restrict__ invstd, float* __restrict__ var_out, float* __restrict__ rm,
                                          float* __restrict__ rv, float* __restrict__ nstat_out)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float m_run = rm[c], v_run = rv[c];
    for (int g = 0; g < n_calls; ++g) {
        const double n = sums[(int64_t)n_calls * 2 * C + g];
        const float unb = n > 1.0 ? (float)(n / (n - 1.0)) : 1.0f;
        if (c == 0) nstat_out[g] = (float)n;
        const double m = sums[((int64_t)g * 2) * C + c] / n;
        double var = sums[((int64_t)g * 2 + 1) * C + c] / n - m * m;
        if (var < 0.0) var = 0.0;
        const int64_t idx = (int64_t)g * C + c;
        mean[idx] = (float)m;
        var_out[idx] = (float)var;
        invstd[idx] = 1.0f / sqrtf((float)var + BN_EPS);
        m_run = (1.0f - BN_MOMENTUM) * m_run + BN_MOMENTUM * (float)m;
        v_run = (1.0f - BN_MOMENTUM) * v_run + BN_MOMENTUM * ((float)var * unb);
    }
    rm[c] = m_run;
    rv[c] = v_run;
}
// ... and the backward's s1 = sum dy, s2 = sum dy xhat from theirs ([call][s1 | s2][C] float64)
__global__ void bn_bwd_from_sums_kernel(const double* __restrict__ sums, int C, int n_calls, float* __restrict__ s1o, float* __restrict__ s2o)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_calls * C) return;
    const int g = idx / C, c = idx % C;
    s1o[idx] = (float)sums[((int64_t)g * 2) * C + c];
    s2o[idx] = (float)sums[((int64_t)g * 2 + 1) * C + c];
}

// backward: s1 = sum dy, s2 = sum dy * xhat per (call, column)
__global__ void bn_bwd_finish_kernel(const double* __restrict__ part, int nchunks, int C, int n_calls,
                                     float* __restrict__ s1o, float* __restrict__ s2o)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_calls * C) return;
    const int g = idx / C, c = idx % C;
    double u = 0.0, v = 0.0;
#pragma unroll 8
    for (int k = 0; k < nchunks; ++k) {
        const double* src = part + ((int64_t)(g * nchunks + k) * 2) * C;
        u += src[c];
        v += src[C + c];
    }
    s1o[idx] = (float)u;
    s2o[idx] = (float)v;
}

// xhat = (z - mean) * invstd ; a = act(gamma * xhat + beta).  train: per-call
// batch stats from ws; eval: running stats.
__global__ void bn_apply_kernel(const float* z /* may alias xhat */, int64_t rows, int64_t rows_per_call, int C,
                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                const float* __restrict__ rm, const float* __restrict__ rv, int train,
                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                int act, float* xhat, float* __restrict__ a, const int* __restrict__ n_valid = nullptr)
{
    const int64_t n = rows * C;
    const int64_t nv = n_valid ? *n_valid : rows_per_call;      // (a padded batch: the rows behind every call's real ones come out as zeros)
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t r = i / C;
        if (r % rows_per_call >= nv) { a[i] = 0.0f; continue; }
        float mu, is;
        if (train) {
            const int64_t g = r / rows_per_call;
            mu = mean[g * C + c];
            is = invstd[g * C + c];
        } else {
            mu = rm[c];
            is = 1.0f / sqrtf(rv[c] + BN_EPS);
        }
        const float xh = (z[i] - mu) * is;
        if (xhat) xhat[i] = xh;
        a[i] = act_apply(xh * gamma[c] + beta[c], act);
    }
}

// dz = gamma*invstd/n * (n*dy - s1 - xhat*s2); also dgamma = sum_g s2, dbeta = sum_g s1
__global__ void bn_bwd_apply_kernel(const float* __restrict__ a, const float* da,   // da may alias dz
                                    const float* __restrict__ xhat, int64_t rows, int64_t rows_per_call,
                                    int C, int act, const float* __restrict__ gamma,
                                    const float* __restrict__ invstd, const float* __restrict__ s1,
                                    const float* __restrict__ s2, int n_calls, const float* __restrict__ mask,
                                    float* dz, float* __restrict__ dgamma, float* __restrict__ dbeta)
{
    const int64_t n = rows * C;
    const float nf = (float)rows_per_call;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t g = (i / C) / rows_per_call;
        const float dy = da[i] * act_grad(a[i], act);
        const float k = gamma[c] * invstd[g * C + c] / nf;
        dz[i] = k * (nf * dy - s1[g * C + c] - xhat[i] * s2[g * C + c]) * (mask ? mask[i] : 1.0f);
        if (i < C && dgamma) {
            float sg = 0.0f, sb = 0.0f;
            for (int q = 0; q < n_calls; ++q) { sg += s2[(int64_t)q * C + c]; sb += s1[(int64_t)q * C + c]; }
            dgamma[c] = sg;
            dbeta[c] = sb;
        }
    }
}

// bn_planes_backward's small kernels.  The output layer's sums of dy and dy * xhat per workgroup of 32 rows
// (further down bn_bwd_layer_kernel leaves them itself): one thread per column, [workgroup][2][PL_MAXW].
// LOSS (abn_tower_backward_loss on a BatchNorm tower): the launch also IS the pair loss -- its first phase computes, per
// pair, the loss term and the two coefficients of d loss / d e = partner * inv - self * kself from the embeddings the
// forward left (the arithmetic of loss.hip, fp64 per pair, as the data-gradient chain's first phase does: tower_planes.h),
// the main loop forms d loss / d a from them instead of reading it, and stores it for the top layer's launch.
// Rows are [call 0 = tower 1: pairs 0 .. B-1 | call 1 = tower 2]; a workgroup's 32 rows belong to one call.
struct BnLossP {
    const void* y;
    int y_dtype, kind, B;
    double margin, scale;
    const int* n_valid;
    double* loss_partial;          // one per workgroup
    unsigned* loss_counter;        // ticket (zero before, zero after)
    float* loss_out;
    double* loss_accum;
    const float* a_top;            // [2 B][C] the embeddings
    float* da_out;                 // [2 B][C] d loss / d a, written here
};
template <bool LOSS>
__global__ __launch_bounds__(512) void bn_bwd_sums_wg_kernel(const float* __restrict__ da, const float* __restrict__ z,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             int act, int C, int64_t rows_per_call, float* __restrict__ part, BnLossP lp)
{
    __shared__ float su[8][4][64], sv[8][4][64];
    __shared__ double coef[64], term_s[32];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;         // 64 columns x 8 groups of 4 rows
    const int64_t wpc = (rows_per_call + PL_ROWS - 1) / PL_ROWS;    // (bn_fwd_layer_kernel's workgroup -> rows map)
    const int64_t g = blockIdx.x / wpc;
    const int64_t nv_rows = lp.n_valid ? (*lp.n_valid < 0 ? 0 : (*lp.n_valid < rows_per_call ? *lp.n_valid : rows_per_call)) : rows_per_call;
    const int64_t row0 = g * rows_per_call + (blockIdx.x - g * wpc) * PL_ROWS, r_end = g * rows_per_call + nv_rows;      // (a padded batch: abn_tower_desc.n_valid)
    const int64_t r0 = row0 + 4 * ty;
    if (LOSS) {
        const int B = lp.B;
        const int Bv = lp.n_valid ? *lp.n_valid : B;
        const double lscale = lp.n_valid && lp.scale != 1.0 ? 1.0 / (double)(Bv > 0 ? Bv : 1) : lp.scale;
        const int wave = ty, lane = tx;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int lr = 4 * wave + 2 * it + (lane >> 5), l = lane & 31;
            const int64_t gr = row0 + lr;
            double inv = 0.0, kself = 0.0, term = 0.0;
            const int tower = (int)g;
            const int64_t pi_ = gr < r_end ? gr - (int64_t)tower * B : 0;
            const bool ok = gr < r_end && pi_ < Bv;
            const float* a = lp.a_top + pi_ * C;                    // e1[pair], e2[pair]: the order loss.hip sums in
            const float* b = lp.a_top + ((int64_t)B + pi_) * C;
            double dot = 0.0, s11 = 0.0, s22 = 0.0;
            for (int c = l; c < C / 4; c += 32) {
                const float4 u = reinterpret_cast<const float4*>(a)[c];
                const float4 v = reinterpret_cast<const float4*>(b)[c];
                dot += (double)u.x * v.x + (double)u.y * v.y + (double)u.z * v.z + (double)u.w * v.w;
                s11 += (double)u.x * u.x + (double)u.y * u.y + (double)u.z * u.z + (double)u.w * u.w;
                s22 += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
            }
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) {
                dot += __shfl_xor(dot, o, 64);
                s11 += __shfl_xor(s11, o, 64);
                s22 += __shfl_xor(s22, o, 64);
            }
            if (ok) {
                constexpr double EPS = 1e-6;
                const double n1 = sqrt(s11), n2 = sqrt(s22);
                const double c1 = n1 > EPS ? n1 : EPS, c2 = n2 > EPS ? n2 : EPS;
                const double cs = dot / (c1 * c2);
                double v = 0.0;
                switch (lp.y_dtype) {
                    case ABN_Y_I8: v = ((const int8_t*)lp.y)[pi_]; break;
                    case ABN_Y_I32: v = ((const int32_t*)lp.y)[pi_]; break;
                    case ABN_Y_I64: v = (double)((const int64_t*)lp.y)[pi_]; break;
                    case ABN_Y_F32: v = ((const float*)lp.y)[pi_]; break;
                    default: v = ((const double*)lp.y)[pi_]; break;
                }
                const int code = v == 1.0 ? 1 : (v == -1.0 ? -1 : 0);
                double dcos;
                if (lp.kind == ABN_LOSS_COSCOS2) {
                    if (code == 1) { term = (1.0 - cs) * 0.5; dcos = -0.5; }
                    else if (code == -1) { term = cs * cs; dcos = 2.0 * cs; }
                    else { term = cs; dcos = 1.0; }
                } else {
                    if (code == 1) { term = 1.0 - cs; dcos = -1.0; }
                    else if (code == -1) { const double hh = cs - lp.margin; term = hh > 0.0 ? hh : 0.0; dcos = hh >= 0.0 ? 1.0 : 0.0; }
                    else { term = cs; dcos = 1.0; }
                }
                dcos *= lscale;
                inv = dcos / (c1 * c2);
                const double k1 = n1 > 0.0 ? dcos * cs / (c1 * n1) : 0.0;
                const double k2 = n2 > 0.0 ? dcos * cs / (c2 * n2) : 0.0;
                kself = tower ? k2 : k1;
                if (tower) term = 0.0;                               // a pair's term counts once
            }
            if (l == 0) { coef[2 * lr] = inv; coef[2 * lr + 1] = kself; term_s[lr] = term; }
        }
        __syncthreads();
    }
    // four chunks of 64 columns per round (C <= 256: one round): every load of a round is in flight before the first sum,
    // one pair of barriers per round
    for (int c00 = 0; c00 < C; c00 += 256) {
        float u[4], v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c00 + 64 * j + tx;
            u[j] = v[j] = 0.0f;
            if (c < C) {
                const float ga = gamma[c], be = beta[c], mu = mean[g * C + c], is = invstd[g * C + c];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (r0 + r >= r_end) break;
                    const int64_t i = (r0 + r) * C + c;
                    float dav;
                    if (LOSS) {
                        const int lr = 4 * ty + r;
                        const int64_t prow = g ? r0 + r - lp.B : r0 + r + lp.B;
                        dav = (float)(lp.a_top[prow * C + c] * coef[2 * lr] - lp.a_top[i] * coef[2 * lr + 1]);
                        lp.da_out[i] = dav;
                    } else {
                        dav = da[i];
                    }
                    const float xh = (z[i] - mu) * is;
                    const float dy = dav * act_grad(act_apply(xh * ga + be, act), act);
                    u[j] += dy;
                    v[j] += dy * xh;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) { su[ty][j][tx] = u[j]; sv[ty][j][tx] = v[j]; }
        __syncthreads();
        if (ty < 4) {                         // wave j finishes chunk j: the eight row groups in order
            const int c = c00 + 64 * ty + tx;
            if (c < C) {
                float a = su[0][ty][tx], b = sv[0][ty][tx];
                for (int k = 1; k < 8; ++k) { a += su[k][ty][tx]; b += sv[k][ty][tx]; }
                part[(int64_t)blockIdx.x * (2 * PL_MAXW) + c] = a;
                part[(int64_t)blockIdx.x * (2 * PL_MAXW) + PL_MAXW + c] = b;
            }
        }
    }
    if (LOSS) {
        // the loss terms: a ticket per workgroup, drawn at the END of the launch (nobody waits for its round trip but the one wave
        // that sums when it was the last), the last workgroup to arrive adds all partials in a fixed order
        const int Bv = lp.n_valid ? *lp.n_valid : lp.B;
        const double lscale = lp.n_valid && lp.scale != 1.0 ? 1.0 / (double)(Bv > 0 ? Bv : 1) : lp.scale;
        if (ty == 0) {
            int last = 0;
            if (tx == 0) {
                double sum = 0.0;
                for (int i = 0; i < 32; ++i) sum += term_s[i];
                last = abn_ticket_publish(&lp.loss_partial[blockIdx.x], sum, lp.loss_counter, gridDim.x);      // (common.h)
            }
            last = __shfl(last, 0, 64);
            if (last) {
                double sum = 0.0;
                for (int i = tx; i < (int)gridDim.x; i += 64) sum += abn_ticket_partial(&lp.loss_partial[i]);
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 64);
                if (tx == 0) {
                    const float lv = (float)(sum * lscale);
                    *lp.loss_out = lv;
                    if (lp.loss_accum) *lp.loss_accum += (double)lv;      // (one thread of one workgroup per call, calls in stream order)
                    __hip_atomic_store(lp.loss_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
}

// s1 = sum dy, s2 = sum dy * xhat per (call, column) from the workgroups' sums, added in a fixed order in
// float64 (64 columns x 16 groups of workgroups per block, as bn_stats_finish_wg_kernel); d gamma, d beta
__global__ __launch_bounds__(64 * BN_WG_GROUPS) void bn_bwd_finish_wg_kernel(const float* __restrict__ part, int wgs_per_call,
                                                                             int C, int n_calls, float* __restrict__ s1o,
                                                                             float* __restrict__ s2o, float* __restrict__ dgamma,
                                                                             float* __restrict__ dbeta, double* __restrict__ sums_out)
{
    __shared__ double sa[BN_WG_GROUPS][64], sb[BN_WG_GROUPS][64];
    const int tx = threadIdx.x & 63, kg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    const bool ok = c < C;
    const int gpc = n_calls < BN_WG_GROUPS ? BN_WG_GROUPS / n_calls : 1;          // thread groups per call (bn_stats_finish_wg_kernel)
    const int per = (wgs_per_call + gpc - 1) / gpc;
    float sg = 0.0f, sbeta = 0.0f;
    for (int g0 = 0; g0 < n_calls; g0 += BN_WG_GROUPS / gpc) {
        const int g = g0 + kg / gpc, sub = kg % gpc;
        double a = 0.0, b = 0.0;
        if (ok && g < n_calls) {
            const int k0 = sub * per, k1 = min(k0 + per, wgs_per_call);
#pragma unroll 8
            for (int k = k0; k < k1; ++k) {
                const float* src = part + (int64_t)(g * wgs_per_call + k) * (2 * PL_MAXW);
                a += (double)src[c];
                b += (double)src[PL_MAXW + c];
            }
        }
        __syncthreads();
        sa[kg][tx] = a;
        sb[kg][tx] = b;
        __syncthreads();
        if (ok && kg == 0) {
            for (int q = 0; q < BN_WG_GROUPS / gpc && g0 + q < n_calls; ++q) {
                const int gq = g0 + q;
                double ta = 0.0, tb = 0.0;
                for (int k = 0; k < gpc; ++k) { ta += sa[q * gpc + k][tx]; tb += sb[q * gpc + k][tx]; }
                if (sums_out) {        // (cross-replica statistics: s1 / s2 follow from the all-reduced sums; d gamma, d beta stay this replica's)
                    sums_out[((int64_t)gq * 2) * C + c] = ta;
                    sums_out[((int64_t)gq * 2 + 1) * C + c] = tb;
                } else {
                    s1o[(int64_t)gq * C + c] = (float)ta;
                    s2o[(int64_t)gq * C + c] = (float)tb;
                }
                sg += (float)tb;
                sbeta += (float)ta;
            }
        }
    }
    if (ok && kg == 0) {
        dgamma[c] = sg;
        dbeta[c] = sbeta;
    }
}

// ---------------------------------------------------------------------------
// workspace layout
// ---------------------------------------------------------------------------
struct Layout {
    int64_t x;                           // [rows, dims[0]] copy of the inputs
    int64_t a[ABN_MAX_LAYERS];           // post-activation outputs
    int64_t xhat[ABN_MAX_LAYERS];        // BN only (z is produced here, then normalised in place)
    int64_t mean[ABN_MAX_LAYERS], invstd[ABN_MAX_LAYERS], var[ABN_MAX_LAYERS];
    int64_t bn_part;                     // BN only: stage-1 partial sums (doubles), shared by the layers
    // precision 1 / 2 without BN (tower_planes.h): the weights and their transposes as MFMA operand
    // fragments, and the weight-gradient operands [x | 1], [a_l | 1] as transposed planes
    int64_t wpack, tp[ABN_MAX_LAYERS];   // wpack: the PackLayout image (unless the caller keeps a persistent one)
    int64_t amax[ABN_MAX_LAYERS];        // fp16 x 2: tp[l]'s maxima per 32-row block (the weight-gradient launch's scales)
    int64_t bn_wg;                       // per-workgroup column statistics of bn_fwd_layer_kernel ([rows / 32][3][PL_MAXW])
    int64_t bn_nstat;                    // cross-replica statistics: the rows a call's statistics span, per layer and call ([layer][8] floats)
    int64_t total;
};

// operand planes of a tower's arithmetic: bf16 one, bf16 x 3 three, fp16 x 2 two
static inline int planes_of(const abn_tower_desc* t) { return t->precision == 3 ? 2 : (t->precision == 2 ? 3 : 1); }
// one of a kernel's three instantiations (operand planes)
#define PL_LAUNCH(np, KERNEL, grid, block, lds, st, ...)                                             \
    do {                                                                                             \
        if ((np) == 3) hipLaunchKernelGGL(KERNEL<3>, grid, block, lds, st, __VA_ARGS__);             \
        else if ((np) == 2) hipLaunchKernelGGL(KERNEL<2>, grid, block, lds, st, __VA_ARGS__);        \
        else hipLaunchKernelGGL(KERNEL<1>, grid, block, lds, st, __VA_ARGS__);                       \
    } while (0)
#define PL_LDS_ATTR(KERNEL, BYTES)                                                                                                       \
    do {                                                                                                                                 \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES(1))); \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES(2))); \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES(3))); \
    } while (0)
static inline size_t wgrad_lds_of(int np) { return np == 3 ? wgrad_lds_bytes<3>() : (np == 2 ? wgrad_lds_bytes<2>() : wgrad_lds_bytes<1>()); }
// the launch's dynamic LDS: the largest stage ring of the table's shapes (shape 3, 8 operand blocks per row step, needs
// 4 x 8 x 2 KB: two workgroups fit a CU)
static inline bool wgrad_all128(int np, const WgradP& w)
{
    bool all3 = np == 2 && w.n_layers > 0;
    for (int i = 0; i < w.n_layers; ++i) all3 = all3 && w.L[i].shape == 3;
    return all3;
}
constexpr size_t WGRAD128_LDS = (size_t)4 * 8 * 2048 + 128;
// every layer on 128 x 128 tiles: the shape's own kernel (fewer registers, 64 KB of LDS: two workgroups per CU); else the general one
static void launch_wgrad(int np, const WgradP& w, int n_wg, hipStream_t st)
{
#ifndef WG128_GENERAL
    if (wgrad_all128(np, w)) {
        static bool attr_set[16] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        dev = (dev >= 0 && dev < 16) ? dev : 0;
        if (!attr_set[dev]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_planes128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WGRAD128_LDS);
            attr_set[dev] = true;
        }
        hipLaunchKernelGGL(wgrad_planes128_kernel, dim3((unsigned)n_wg), dim3(PL_NT), WGRAD128_LDS, st, w);
        return;
    }
#endif
    PL_LAUNCH(np, wgrad_planes_kernel, dim3((unsigned)n_wg), dim3(PL_NT), wgrad_all128(np, w) ? WGRAD128_LDS : wgrad_lds_of(np), st, w);
}
// the arithmetic the GEMM kernels (gemm_f32.h, tower_fused.h) run a precision in: fp16 x 2 exists on the operand planes only
static inline int gemm_prec(int precision) { return precision > 2 ? 2 : precision; }
// cross-replica BatchNorm statistics are on: a function to sum them with, for a group of bn_sync_world >= 1 replicas
// (a group of one: the function's reduction is the identity, the launches are the group's)
static inline bool bn_sync_on(const abn_tower_desc* t) { return t->bn_sync_fn != nullptr && t->bn_sync_world >= 1; }
// BatchNorm launches: workgroups of 32 rows never straddle two forward_once calls, so the row axis of the
// transposed images is padded per call
static inline int64_t bn_wgs_per_call(int64_t rows, int64_t n_calls) { return (rows / n_calls + PL_ROWS - 1) / PL_ROWS; }
static inline int64_t bn_vrows(int64_t rows, int64_t n_calls) { return n_calls * bn_wgs_per_call(rows, n_calls) * PL_ROWS; }
// the operand-plane kernels (tower_planes.h) take this tower's arithmetic and widths; with BatchNorm only
// its inference forward (running statistics: a per-feature affine map in the epilogue)
static bool planes_dims_ok(const abn_tower_desc* t)
{
    if (t->precision < 1) return false;
    for (int l = 0; l <= t->n_layers; ++l)
        if (t->dims[l] < 4 || t->dims[l] > PL_MAXW || t->dims[l] % 4 != 0) return false;
    return true;
}
static bool planes_shape_ok(const abn_tower_desc* t) { return !t->batch_norm && planes_dims_ok(t); }

// The weights as operand fragments: W_l and W_l^T for every layer, in a caller-owned persistent buffer
// (abn_tower_desc.wpack) or inside the forward's workspace.  Byte offsets from the image's base.
struct PackLayout { int64_t wp[ABN_MAX_LAYERS], wpt[ABN_MAX_LAYERS], bytes; };
static PackLayout make_pack_layout(const abn_tower_desc* t)
{
    PackLayout P = {};
    if (!planes_dims_ok(t)) return P;
    const int np = planes_of(t);
    int64_t o = 0;
    for (int l = 0; l < t->n_layers; ++l) {
        P.wp[l] = o; o += pl_image_bytes(t->dims[l + 1], t->dims[l], np);
        P.wpt[l] = o; o += pl_image_bytes(t->dims[l], t->dims[l + 1], np);
    }
    P.bytes = o;
    return P;
}

static Layout make_layout(const abn_tower_desc* t, int64_t rows, int64_t n_calls)
{
    Layout L;
    int64_t o = 0;
    auto take = [&](int64_t n) { int64_t r = o; o += align_up(n, 64); return r; };
    L.x = take(rows * t->dims[0]);
    const int64_t vrows = bn_vrows(rows, n_calls);       // every call padded to whole 32-row workgroups (tower_wide.h, BatchNorm launches)
    for (int l = 0; l < t->n_layers; ++l) {
        const int64_t w = t->dims[l + 1];
        L.a[l] = take(vrows * w);
        if (t->batch_norm) {
            L.xhat[l] = take(rows * w);
            L.mean[l] = take(n_calls * w);
            L.invstd[l] = take(n_calls * w);
            L.var[l] = take(n_calls * w);
        } else {
            L.xhat[l] = L.mean[l] = L.invstd[l] = L.var[l] = -1;
        }
    }
    L.bn_part = -1;
    if (t->batch_norm) {
        int64_t maxw = 0;
        for (int l = 1; l <= t->n_layers; ++l) maxw = t->dims[l] > maxw ? t->dims[l] : maxw;
        L.bn_part = take(2 * n_calls * bn_chunks(rows / n_calls) * 2 * maxw + 2 * n_calls);      // doubles = 2 floats each (+ the calls' row counts)
    }
    for (int l = 0; l < t->n_layers; ++l) L.tp[l] = L.amax[l] = -1;
    L.wpack = -1;
    L.bn_wg = -1;
    L.bn_nstat = -1;
    if (planes_dims_ok(t) && t->batch_norm && !t->forward_only) {
        L.bn_nstat = take(ABN_MAX_LAYERS * 8);
        L.bn_wg = take((n_calls * bn_wgs_per_call(rows, n_calls) + 1) * 3 * PL_MAXW);

    }
    if (planes_dims_ok(t)) {
        const int np = planes_of(t);
        L.wpack = take(make_pack_layout(t).bytes / 4);
        if (!t->forward_only)                                      // (last in the workspace: an inference call simply asks for less)
            for (int l = 0; l < t->n_layers; ++l) {
                L.tp[l] = take(pl_timage_bytes(t->dims[l] + 1, vrows, np) / 4);
                if (np == 2) L.amax[l] = take(pl_amax_floats(vrows));
            }
    }
    L.total = o;
    return L;
}

// Whether the planes kernels (tower_planes.h) take a call.  Forward and backward must agree (the forward then
// leaves the hidden activations in the transposed images only): both ask here.
enum { PLANES_TRAIN = 0, PLANES_EVAL_FORWARD = 1 };      // a forward in train mode or any backward | a forward with train == 0
enum { PLANES_NONE = 0, PLANES_CHAIN = 1, PLANES_WIDE = 2 };

// Workgroups per 32-row block of the layer-per-launch kernels (tower_wide.h) for a batch of `vrows` virtual
// rows: as many as keep the grid within one wave of workgroups over the 256 CUs; 0 = too many rows, the
// single-launch chains take the batch (measured, 280 -> 500 x 2 -> 100: layer per launch 151 / 172 / 180 / 202 us at
// 1024 / 1280 / 1536 / 2048 pairs against the chains' 175 / 182 / 187 / 195).
static int wide_groups(int64_t vrows, int max_blocks = 2 * WD_MAXG)
{
    const int64_t cap = switches().wide_max_rows >= 0 ? switches().wide_max_rows : 3072;
    if (!switches().wide || vrows > cap || vrows < PL_ROWS) return 0;
    const int64_t nrb = vrows / PL_ROWS;
    int64_t G = 256 / nrb;
    G = G < 1 ? 1 : (G > switches().wide_max_groups ? switches().wide_max_groups : G);
    // a workgroup's eight waves take at most one output block each (WideShare): the widest layer's blocks
    // must fit 8 x G
    if ((max_blocks + G - 1) / G > PL_WAVES) return 0;
    return (int)G;
}
static int wide_groups_for(const abn_tower_desc* t, int64_t vrows)
{
    int64_t maxw = 0;
    for (int l = 0; l <= t->n_layers; ++l) maxw = t->dims[l] > maxw ? t->dims[l] : maxw;
    return wide_groups(vrows, pl_blocks(maxw));
}

static int planes_kind(const abn_tower_desc* t, int64_t rows, int64_t n_calls, const float* x1, const float* x2, const float* ws,
                       int mode = PLANES_TRAIN, bool allow_wide = true)
{
    if (!switches().planes || !switches().fused) return PLANES_NONE;
    // BatchNorm: only the inference forward (running statistics), and only when no backward will follow
    if (!(mode == PLANES_EVAL_FORWARD && t->forward_only ? planes_dims_ok(t) : planes_shape_ok(t))) return PLANES_NONE;
    if (!aligned16(x1) || (x2 && !aligned16(x2)) || !aligned16(ws)) return PLANES_NONE;
    bool any_mask = false;
    for (int l = 0; l < t->n_layers; ++l) {
        if (!aligned16(t->W[l]) || !aligned16(t->b[l]) || (t->drop_mask[l] && !aligned16(t->drop_mask[l]))) return PLANES_NONE;
        any_mask = any_mask || t->drop_mask[l] != nullptr;
    }
    // small batches: one launch per layer, the output blocks dealt over several workgroups per 32 rows
    // (dropout there comes from the per-forward seed only; mask tensors stay with the chains)
    if (allow_wide && !t->batch_norm && !any_mask && n_calls >= 1 && rows % n_calls == 0 && wide_groups_for(t, bn_vrows(rows, n_calls)) > 0) return PLANES_WIDE;
    // a workgroup walks its 32 rows through every layer in ~50 us whatever the batch; from a few workgroups
    // up that beats the per-layer GEMMs (tools/rows_sweep.py, tools/fwd_rows_sweep.py: C2 train step 0.148 vs
    // 0.180 ms at 512 rows, 0.234 vs 0.331 at 8192; forward alone 62 vs 94 us at 5000 rows)
    const int64_t min_rows = switches().fused_min_rows >= 0 ? switches().fused_min_rows : 256;
    if (rows < min_rows || rows > (1LL << 20)) return PLANES_NONE;                          // (32-bit byte offsets inside one image)
    return PLANES_CHAIN;
}
static bool planes_path(const abn_tower_desc* t, int64_t rows, const float* x1, const float* x2, const float* ws,
                        int mode = PLANES_TRAIN, int64_t n_calls = 1)
{
    return planes_kind(t, rows, n_calls, x1, x2, ws, mode) != PLANES_NONE;
}

// The training forward and the backward of a BatchNorm tower, one operand-plane launch per layer
// (bn_fwd_layer_kernel, bn_bwd_layer_kernel): same conditions.  Forward and backward must agree: both ask here.
// ABN_BN_PLANES=0: the per-layer kernels (A/B measurements).
static bool bn_train_planes_path(const abn_tower_desc* t, int64_t rows, int64_t n_calls, const float* x1, const float* x2,
                                 const float* ws)
{
    if (!t->batch_norm || t->forward_only) return false;
    if (!switches().bn_planes) return false;
    (void)n_calls;                                  // (any split of the rows into calls: workgroups are cut per call)
    abn_tower_desc u = *t;
    u.batch_norm = 0;
    return planes_kind(&u, rows, 1, x1, x2, ws, PLANES_TRAIN, false) != PLANES_NONE;
}

// The resident BatchNorm tower (tower_bn_persist.h: one launch per direction, grid barriers between the layers) takes a
// training call of the BatchNorm layer launches when its grid -- one workgroup per 32 rows of a call -- is resident at once:
// at most as many workgroups as the device has CUs (the kernels need more than half a CU's LDS and registers: one
// workgroup per CU), at least 8 (the finishing step's share of the columns fits its LDS scratch), per-replica
// statistics (a cross-replica exchange happens on the host, between launches), and the caller keeps a sync buffer
// (abn_tower_desc.sync_ws: the workgroups' hand-overs and the launch counter their tags come from).  ABN_BN_PERSIST=0: never.
static int device_cus()
{
    static int cus[16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev = (dev >= 0 && dev < 16) ? dev : 0;
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 1;
        cus[dev] = n;
    }
    return cus[dev];
}
static bool bn_persist_path(const abn_tower_desc* t, int64_t rows, int64_t n_calls)
{
    if (!switches().bn_persist || bn_sync_on(t) || !t->sync_ws || !aligned16(t->sync_ws)) return false;
    const int64_t grid = n_calls * bn_wgs_per_call(rows, n_calls);
    return grid >= 8 && grid <= device_cus() && grid <= BNP_MAX_WGS && n_calls <= BNP_MAX_CALLS;
}
static inline size_t bnp_lds_bytes(int np) { return pl_lds_bytes_bn(np) + 16; }      // + the barrier's LDS word

// BatchNorm1d.num_batches_tracked (abnet3/model.py:139: torch's BatchNorm1d counts its training calls): += n_calls per layer
struct NbtP { long long* p[ABN_MAX_LAYERS]; int n, add; };
__global__ void bn_nbt_kernel(NbtP a)
{
    if ((int)threadIdx.x < a.n && a.p[threadIdx.x]) *a.p[threadIdx.x] += a.add;
}
static void launch_nbt(const abn_tower_desc* t, int64_t n_calls, hipStream_t st)
{
    NbtP a = {};
    bool any = false;
    a.n = t->n_layers; a.add = (int)n_calls;
    for (int l = 0; l < t->n_layers; ++l) { a.p[l] = reinterpret_cast<long long*>(t->bn_nbt[l]); any = any || a.p[l]; }
    if (any) hipLaunchKernelGGL(bn_nbt_kernel, dim3(1), dim3(64), 0, st, a);
}

static int check_desc(const abn_tower_desc* t, int64_t rows, int64_t n_calls)
{
    ABN_REQUIRE(t != nullptr, "tower: null descriptor");
    ABN_REQUIRE(t->n_layers >= 1 && t->n_layers <= ABN_MAX_LAYERS, "tower: n_layers=%d out of range", t->n_layers);
    ABN_REQUIRE(t->precision >= 0 && t->precision <= 3, "tower: precision=%d (0 = fp32, 1 = bf16 operands, 2 = bf16 x 3, 3 = fp16 x 2)", t->precision);
    ABN_REQUIRE(rows >= 0 && n_calls >= 1 && rows % n_calls == 0, "tower: rows=%lld not divisible by n_calls=%lld",
                (long long)rows, (long long)n_calls);
    for (int l = 0; l <= t->n_layers; ++l)
        ABN_REQUIRE(t->dims[l] >= 1 && t->dims[l] < (1 << 24), "tower: dims[%d]=%lld invalid", l, (long long)t->dims[l]);
    ABN_REQUIRE(rows < (1LL << 30), "tower: too many rows");
    for (int a : {t->act, t->last_act})
        if (a < ABN_ACT_NONE || a > ABN_ACT_TANH) { set_error("tower: unsupported activation %d", a); return ABN_E_UNSUPPORTED; }
    for (int l = 0; l < t->n_layers; ++l) {
        ABN_REQUIRE(t->W[l] && t->b[l], "tower: layer %d has null weight/bias", l);
        if (t->batch_norm)
            ABN_REQUIRE(t->bn_w[l] && t->bn_b[l] && t->bn_rm[l] && t->bn_rv[l], "tower: layer %d has null BN tensors", l);
    }
    return ABN_OK;
}

// Split-K factor of one wgrad GEMM (reduction over `rows`): enough slices that
// the launch has ~2 workgroups per CU, each slice at least 128 rows deep.
constexpr int MAX_SPLITS = 32;
static int split_count(int64_t rows, int64_t out_dim, int64_t in_dim)
{
    const int64_t n = in_dim + 1;                       // + the bias column
    const int64_t tm128 = (out_dim + 127) / 128, tm64 = (out_dim + 63) / 64;
    const int64_t tn128 = (n + 127) / 128, tn64 = (n + 63) / 64;
    const int64_t est = out_dim > 64 ? tm128 * tn64 : (n > 64 ? tm64 * tn128 : tm64 * tn64);
    int64_t s = (512 + est - 1) / est;
    const int64_t by_rows = rows / 128 < 1 ? 1 : rows / 128;
    if (s > by_rows) s = by_rows;
    if (s > MAX_SPLITS) s = MAX_SPLITS;
    return (int)(s < 1 ? 1 : s);
}

// The same for a tower the planes kernels can take (tower_planes.h, whichever path runs in the end):
// their weight-gradient tiles are up to 256 x 256, so a layer needs more slices to spread over the CUs.
// 128 x 128 weight-gradient tiles (shape 3, fp16 x 2): from 2048 tower rows on (rows rounded up to 64 like the slices below, so
// that a batch and its padded form agree; tools/tile128_sweep.py: 1024 rows 0.128 against 0.122 ms / step on the wide tiles, 2048 rows
// 0.119 against 0.129, 6144 rows 0.147 against 0.158); ABN_WGRAD_TILE128 = 0 / 1 forces never / always
static inline bool wgrad_tile128(int64_t rows, bool np2)
{
    const int mode = switches().wgrad_tile128;
    return np2 && mode != 0 && (mode > 0 || (rows + 63) / 64 * 64 >= 2048);
}
static int planes_split_count(int64_t rows, int64_t out_dim, int64_t in_dim, bool np2 = false)
{
    int shape, bn, bk;
    const int nblk = pl_blocks(out_dim), kblk = pl_blocks(in_dim + 1);
    wgrad_shape(nblk, kblk, &shape, &bn, &bk, wgrad_tile128(rows, np2));
    const int64_t tiles = (int64_t)((nblk + bn - 1) / bn) * ((kblk + bk - 1) / bk);
    // ~128 workgroups per layer.  The light shapes (first / output layer: half-empty tiles, small slabs) may be cut
    // twice as fine: at C2 every CU then gets one heavy workgroup (32 row steps) and one light one (8) -- with
    // 16-step light ones half the CUs idled through the launch's tail
    int64_t s = (128 + tiles - 1) / tiles;
    if (shape == 3) { const int64_t want = tiles >= 16 ? switches().wgrad_wgs_heavy : switches().wgrad_wgs_light; s = (want + tiles - 1) / tiles; }
    // (rows rounded up to 64: a Siamese batch of n pairs and the same batch padded to ceil32(n) pairs -- the
    // captured steps of the trainer's planned passes -- get the same slices, hence bit-identical gradients)
    const int64_t r64 = (rows + 63) / 64 * 64;
    const int64_t per = switches().wgrad_rows_per_slab > 0 ? switches().wgrad_rows_per_slab : 128;
    const int64_t by_rows = r64 / per < 1 ? 1 : r64 / per;
    const int64_t cap = shape == 0 ? MAX_SPLITS : 2 * MAX_SPLITS;
    if (s > by_rows) s = by_rows;
    if (s > cap) s = cap;
    return (int)(s < 1 ? 1 : s);
}

struct BwdLayout {
    int64_t dz[2];
    int64_t bn_s1, bn_s2;
    int64_t bn_part;                 // stage-1 partial sums of the BatchNorm backward (doubles)
    int64_t slabs;
    int64_t slab_stride;
    int64_t off[ABN_MAX_LAYERS];
    int splits[ABN_MAX_LAYERS];
    int psplits[ABN_MAX_LAYERS];     // BatchNorm: the slabs of the operand-plane weight-gradient launch (bn_planes_backward)
    int64_t dzp[ABN_MAX_LAYERS];     // tower_planes.h: transposed planes of dZ_l
    int64_t amax_dz[ABN_MAX_LAYERS]; // fp16 x 2: their maxima per 32-row block
    int64_t bn_wg;                   // bn_bwd_layer_kernel's per-workgroup sums ([rows / 32][2][PL_MAXW])
    int64_t total;
};

static BwdLayout make_bwd_layout(const abn_tower_desc* t, int64_t rows)
{
    BwdLayout B;
    int64_t maxw = 0, o = 0, packed = 0;
    for (int l = 0; l <= t->n_layers; ++l) maxw = t->dims[l] > maxw ? t->dims[l] : maxw;
    auto take = [&](int64_t n) { int64_t r = o; o += align_up(n, 64); return r; };
    B.dz[0] = take((rows + 8 * PL_ROWS) * maxw);         // (virtual rows: up to 8 calls, each padded to 32 rows)
    B.dz[1] = take((rows + 8 * PL_ROWS) * maxw);
    B.bn_s1 = take(8 * maxw);
    B.bn_s2 = take(8 * maxw);
    B.bn_part = take(2 * 8 * BN_MAX_CHUNKS * 2 * maxw);
    for (int l = 0; l < t->n_layers; ++l) {
        B.off[l] = packed;
        packed += align_up(t->dims[l + 1] * t->dims[l] + t->dims[l + 1], 64);
    }
    B.slab_stride = align_up(packed, 64);
    int smax = 1;
    for (int l = 0; l < t->n_layers; ++l) {
        // (one count whichever kernels run the backward of a tower the operand-plane kernels could take: abn_tower_reduce_step
        // is told the descriptor and the row count only)
        B.splits[l] = planes_dims_ok(t) ? planes_split_count(rows, t->dims[l + 1], t->dims[l], planes_of(t) == 2) : split_count(rows, t->dims[l + 1], t->dims[l]);
        B.psplits[l] = planes_dims_ok(t) ? planes_split_count(rows, t->dims[l + 1], t->dims[l], planes_of(t) == 2) : B.splits[l];
        smax = B.splits[l] > smax ? B.splits[l] : smax;
        smax = B.psplits[l] > smax ? B.psplits[l] : smax;
    }
    B.slabs = take(B.slab_stride * smax);
    for (int l = 0; l < t->n_layers; ++l) {
        B.dzp[l] = planes_dims_ok(t) ? take(pl_timage_bytes(t->dims[l + 1], rows + 8 * PL_ROWS, planes_of(t)) / 4) : -1;
        B.amax_dz[l] = planes_dims_ok(t) && planes_of(t) == 2 ? take(pl_amax_floats(rows + 8 * PL_ROWS)) : -1;
    }
    B.bn_wg = planes_dims_ok(t) && t->batch_norm ? take((rows / PL_ROWS + 9) * 2 * PL_MAXW) : -1;     // (up to 8 calls, each padded)
    B.total = o;
    return B;
}

static inline int grid_for(int64_t n) { int64_t g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g)); }

static ReduceTable make_reduce_table(const abn_tower_desc* t, const BwdLayout& B)
{
    ReduceTable rt = {};
    const int nl = t->n_layers;
    rt.n_layers = nl;
    rt.slab_stride = B.slab_stride;
    for (int l = 0; l < nl; ++l) {
        rt.splits[l] = B.splits[l];
        rt.off[l] = B.off[l];
        rt.nW[l] = t->dims[l + 1] * t->dims[l];
        rt.nb[l] = t->dims[l + 1];
        rt.dW[l] = t->dW[l];
        rt.db[l] = t->db[l];
    }
    rt.total = B.off[nl - 1] + rt.nW[nl - 1] + rt.nb[nl - 1];
    if (t->batch_norm && t->sync_ws) {
        rt.fail_word = reinterpret_cast<const unsigned*>(t->sync_ws) + 16;
        for (int l = 0; l < nl; ++l) {
            if (t->dbn_w[l]) { rt.own[rt.n_own] = t->dbn_w[l]; rt.own_n[rt.n_own++] = (int32_t)rt.nb[l]; }
            if (t->dbn_b[l]) { rt.own[rt.n_own] = t->dbn_b[l]; rt.own_n[rt.n_own++] = (int32_t)rt.nb[l]; }
        }
    }
    return rt;
}

// The pair loss riding in the chain kernel's first phase (abn_tower_backward_loss)
struct LossArgs {
    const void* y;
    int y_dtype, kind, avg;
    float margin;
    float* loss_out;
    void* ws;              // 8 bytes of ticket counter (zero before the first call, left zero) + one double per workgroup
    const int32_t* n_valid;
    double* loss_accum;
};

// Every layer's weight-gradient tiles for one wgrad_planes_kernel launch: operands = the transposed images
// the forward (ws + L.tp[l]) and the data-gradient launches (scratch + B.dzp[l]) left.
static WgradP make_wgrad(const abn_tower_desc* t, int64_t rows, const Layout& L, const BwdLayout& B, const float* ws,
                         float* scratch, int* n_wg_out, int l_first = 0, int l_end = ABN_MAX_LAYERS)
{
    const int nl = t->n_layers;
    WgradP w = {};
    w.slabs = scratch + B.slabs;
    w.slab_stride = B.slab_stride;
    w.tp_steps = pl_row_steps(rows);
    // launch order: most row steps per workgroup first
    int order[ABN_MAX_LAYERS];
    for (int l = 0; l < nl; ++l) order[l] = l;
    for (int i = 1; i < nl; ++i)
        for (int j = i; j > 0 && B.splits[order[j]] < B.splits[order[j - 1]]; --j) { int tmp = order[j]; order[j] = order[j - 1]; order[j - 1] = tmp; }
    int n_wg = 0;
    for (int i = 0; i < nl; ++i) {
        const int l = order[i];
        if (l < l_first || l >= l_end) continue;        // (abn_tower_desc.wgrad_part: one half of the layers)
        WgradLayer& W = w.L[w.n_layers++];
        W.dzp = reinterpret_cast<const char*>(scratch + B.dzp[l]);
        W.ap = reinterpret_cast<const char*>(ws + L.tp[l]);
        W.amax_dz = B.amax_dz[l] >= 0 ? scratch + B.amax_dz[l] : nullptr;
        W.amax_a = L.amax[l] >= 0 ? ws + L.amax[l] : nullptr;
        W.N = (int)t->dims[l + 1]; W.K = (int)t->dims[l];
        W.nblk = pl_blocks(W.N); W.kblk = pl_blocks(W.K + 1);
        int bn, bk;
        wgrad_shape(W.nblk, W.kblk, &W.shape, &bn, &bk, wgrad_tile128(rows, planes_of(t) == 2));
        W.tiles_n = (W.nblk + bn - 1) / bn; W.tiles_k = (W.kblk + bk - 1) / bk;
        W.splits = B.splits[l];
        W.first_wg = n_wg;
        W.slab_off = B.off[l];
        n_wg += W.tiles_n * W.tiles_k * W.splits;
    }
    // (ABN_WGRAD_XCD=0: workgroups in launch order, A/B measurements)
    w.xcd_groups = switches().wgrad_xcd;
#ifdef ABN_STAMPS
    w.stamps = getenv("ABN_WSTAMP_BUF") ? (unsigned long long*)strtoull(getenv("ABN_WSTAMP_BUF"), nullptr, 0) : nullptr;
#endif
    if (w.xcd_groups) {
        int most = 0;
        for (int x = 0; x < 8; ++x) most = wgrad_slots(w, x) > most ? wgrad_slots(w, x) : most;
        n_wg = 8 * most;
    }
    *n_wg_out = n_wg;
    return w;
}

// abn_tower_desc.wgrad_part / wgrad_split -> [first, end) layers whose weight gradients this call computes and reduces
static int wgrad_range(const abn_tower_desc* t, int* l_first, int* l_end)
{
    *l_first = 0; *l_end = t->n_layers;
    if (t->wgrad_part == 0) return ABN_OK;
    ABN_REQUIRE(t->wgrad_part == 1 || t->wgrad_part == 2, "tower_backward: wgrad_part=%d", t->wgrad_part);
    ABN_REQUIRE(t->wgrad_split >= 0 && t->wgrad_split <= t->n_layers, "tower_backward: wgrad_split=%d", t->wgrad_split);
    ABN_REQUIRE(!t->defer_reduce, "tower_backward: wgrad_part cannot be combined with defer_reduce");
    if (t->wgrad_part == 1) *l_first = t->wgrad_split;
    else *l_end = t->wgrad_split;
    return ABN_OK;
}
static void reduce_range(ReduceTable& rt, const abn_tower_desc* t, const BwdLayout& B, int l_first, int l_end)
{
    rt.begin = B.off[l_first < t->n_layers ? l_first : t->n_layers - 1];
    if (l_first >= t->n_layers) rt.begin = rt.total;
    if (l_end < t->n_layers) rt.total = B.off[l_end];
}

// Backward of a BatchNorm tower whose forward went through bn_fwd_layer_kernel (same predicate): per layer,
// top down, [bn_bwd_layer_kernel, bn_bwd_finish_wg_kernel]; then every layer's weight gradient in one launch
// and the slab reduction.
static int bn_planes_backward(const abn_tower_desc* t, const float* d_out, const LossArgs* loss, int64_t rows, int64_t n_calls, const Layout& L,
                              const BwdLayout& B0, const float* ws, float* scratch, float* dx, hipStream_t st)
{
    const int nl = t->n_layers;
    const int np = planes_of(t);
    const int64_t rpc = rows / n_calls;
    ABN_REQUIRE(aligned16(d_out) && aligned16(scratch) && (!dx || aligned16(dx)), "tower_backward: d_out / scratch / dx must be 16-byte aligned");
    // (with the pair loss riding along there is no d_out at all: the flag, which the caller sets for the chains, says nothing here)
    ABN_REQUIRE(!t->d_out_is_dz || loss, "tower_backward: d_out_is_dz cannot be combined with batch_norm");
    ABN_REQUIRE(!loss || n_calls == 2, "tower_backward_loss: two forward_once calls");
    BwdLayout B = B0;
    for (int l = 0; l < nl; ++l) B.splits[l] = B0.psplits[l];
    const PackLayout PL = make_pack_layout(t);
    const char* const image = t->wpack ? reinterpret_cast<const char*>(t->wpack) : reinterpret_cast<const char*>(ws + L.wpack);
    static bool attr_set[16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev = (dev >= 0 && dev < 16) ? dev : 0;
    if (!attr_set[dev]) {
        PL_LDS_ATTR(bn_bwd_layer_kernel, pl_lds_bytes);
        PL_LDS_ATTR(wgrad_planes_kernel, wgrad_lds_of);
        attr_set[dev] = true;
    }
    const int wgs_per_call = (int)bn_wgs_per_call(rows, n_calls);
    const dim3 cgrid((unsigned)(n_calls * wgs_per_call));
    const int64_t tp_steps = 2 * n_calls * wgs_per_call;      // (the images' row axis is padded per call)
    float* const part = scratch + B.bn_wg;
    float* const s1 = scratch + B.bn_s1, * const s2 = scratch + B.bn_s2;
    // cross-replica statistics: the sums leave as float64, are all-reduced by the caller's function, and come back as s1 / s2
    const bool sync = bn_sync_on(t);
    ABN_REQUIRE(!sync || t->bn_sync_fn, "tower_backward: bn_sync_world = %d without bn_sync_fn", t->bn_sync_world);
    // a padded batch: the descriptor's real-row count (the forward was given the same), or the loss call's
    ABN_REQUIRE(!loss || !loss->n_valid || !t->n_valid || loss->n_valid == t->n_valid, "tower_backward_loss: two different n_valid");
    const int32_t* const nvp = t->n_valid ? t->n_valid : (loss ? loss->n_valid : nullptr);
    if (sync && nvp) { set_error("tower_backward: n_valid (a padded batch) cannot be combined with cross-replica BatchNorm statistics"); return ABN_E_UNSUPPORTED; }
    double* const sums = reinterpret_cast<double*>(scratch + B.bn_part);
    auto finish_sums = [&](int C, float* dgamma, float* dbeta) -> int {
        hipLaunchKernelGGL(bn_bwd_finish_wg_kernel, dim3((unsigned)((C + 63) / 64)), dim3(64 * BN_WG_GROUPS), 0, st, part, wgs_per_call, C,
                           (int)n_calls, s1, s2, dgamma, dbeta, sync ? sums : static_cast<double*>(nullptr));
        if (!sync) return ABN_OK;
        if (t->bn_sync_fn(t->bn_sync_ctx, sums, n_calls * 2 * C, st) != 0) { set_error("tower_backward: bn_sync_fn failed"); return ABN_E_LAUNCH; }
        hipLaunchKernelGGL(bn_bwd_from_sums_kernel, dim3(grid_for(n_calls * C)), dim3(256), 0, st, sums, C, (int)n_calls, s1, s2);
        return ABN_OK;
    };
    if (bn_persist_path(t, rows, n_calls)) {
        // every layer in ONE resident launch, grid barriers in between (tower_bn_persist.h)
        static bool bt_attr_set[16] = {};
        if (!bt_attr_set[dev]) {
            PL_LDS_ATTR(bn_bwd_tower_kernel, bnp_lds_bytes);
            bt_attr_set[dev] = true;
        }
        BnBwdTowerP q = {};
        q.n_layers = nl; q.rows = (int)rows; q.rows_call = (int)rpc; q.n_calls = (int)n_calls;
        for (int l = 0; l <= nl; ++l) q.dims[l] = (int)t->dims[l];
        for (int l = 0; l < nl; ++l) {
            q.act[l] = (l == nl - 1) ? t->last_act : t->act;
            q.z[l] = ws + L.xhat[l];
            q.mean[l] = ws + L.mean[l]; q.invstd[l] = ws + L.invstd[l];
            q.gamma[l] = t->bn_w[l]; q.beta[l] = t->bn_b[l];
            q.dgamma[l] = t->dbn_w[l]; q.dbeta[l] = t->dbn_b[l];
            q.mask[l] = t->drop_mask[l];
            q.wpt[l] = image + PL.wpt[l];
            q.dzp[l] = reinterpret_cast<char*>(scratch + B.dzp[l]);
            q.amax_dz[l] = B.amax_dz[l] >= 0 ? scratch + B.amax_dz[l] : nullptr;
        }
        q.drop_seed = reinterpret_cast<const unsigned long long*>(t->drop_seed);
        q.drop_p = t->drop_p;
        q.wbase = image; q.wbytes = PL.bytes;
        q.tp_steps = tp_steps;
        q.dx = dx;
        q.d_out = loss ? nullptr : d_out;
        q.a_top = ws + L.a[nl - 1];
        q.sync_ws = t->sync_ws;
        q.sync_bytes = bnp_sync_bytes(BNP_MAX_WGS);
        q.n_valid = nvp;
        q.n_stat = (float)rpc;
        q.loss_kind = -1;
        if (loss) {
            q.loss_kind = loss->kind; q.y = loss->y; q.y_dtype = loss->y_dtype;
            q.margin = (double)loss->margin;
            q.scale = loss->avg ? 1.0 / (double)rpc : 1.0;
            q.loss_counter = reinterpret_cast<unsigned*>(loss->ws);
            q.loss_partial = reinterpret_cast<double*>(reinterpret_cast<char*>(loss->ws) + 8);
            q.loss_out = loss->loss_out;
            q.loss_accum = loss->loss_accum;
        }
        PL_LAUNCH(np, bn_bwd_tower_kernel, cgrid, dim3(PL_NT), bnp_lds_bytes(np), st, q);
    } else {
    // d loss / d a of the output layer: the caller's, or (the pair loss riding along) formed by the first launch from the embeddings
    const float* da_top = d_out;
    {
        const int N = (int)t->dims[nl];
        BnLossP lp = {};
        lp.n_valid = nvp;
        if (loss) {
            lp.y = loss->y; lp.y_dtype = loss->y_dtype; lp.kind = loss->kind; lp.B = (int)rpc;
            lp.margin = (double)loss->margin;
            lp.scale = loss->avg ? 1.0 / (double)rpc : 1.0;
            lp.loss_counter = reinterpret_cast<unsigned*>(loss->ws);
            lp.loss_partial = reinterpret_cast<double*>(reinterpret_cast<char*>(loss->ws) + 8);
            lp.loss_out = loss->loss_out;
            lp.loss_accum = loss->loss_accum;
            lp.a_top = ws + L.a[nl - 1];
            lp.da_out = scratch + B.dz[0];            // (idle until the top layer's launch has read it: that one writes dz[1])
            da_top = lp.da_out;
            hipLaunchKernelGGL(bn_bwd_sums_wg_kernel<true>, cgrid, dim3(512), 0, st, static_cast<const float*>(nullptr), ws + L.xhat[nl - 1],
                               ws + L.mean[nl - 1], ws + L.invstd[nl - 1], t->bn_w[nl - 1], t->bn_b[nl - 1], t->last_act, N, rpc, part, lp);
        } else {
            hipLaunchKernelGGL(bn_bwd_sums_wg_kernel<false>, cgrid, dim3(512), 0, st, d_out, ws + L.xhat[nl - 1], ws + L.mean[nl - 1],
                               ws + L.invstd[nl - 1], t->bn_w[nl - 1], t->bn_b[nl - 1], t->last_act, N, rpc, part, lp);
        }
        const int rc = finish_sums(N, t->dbn_w[nl - 1], t->dbn_b[nl - 1]);
        if (rc != ABN_OK) return rc;
    }
    int cur = 0;
    for (int l = nl - 1; l >= 0; --l) {
        BnBwdP q = {};
        q.l = l; q.rows = (int)rows; q.rows_call = (int)rpc;
        q.n_stat = (float)rpc;
        q.n_stat_dev = sync ? ws + L.bn_nstat + 8 * l : nullptr;       // (cross-replica statistics: the forward's all-reduced row counts)
        q.N = (int)t->dims[l + 1]; q.K = (int)t->dims[l];
        q.act_l = (l == nl - 1) ? t->last_act : t->act;
        q.act_prev = t->act;
        q.da = (l == nl - 1) ? da_top : scratch + B.dz[cur];
        q.z = ws + L.xhat[l];                         // (the forward left z there, un-normalised)
        q.mean = ws + L.mean[l];
        q.invstd = ws + L.invstd[l];
        q.gamma = t->bn_w[l]; q.beta = t->bn_b[l];
        q.s1 = s1; q.s2 = s2;
        q.mask = t->drop_mask[l];
        q.drop_seed = reinterpret_cast<const unsigned long long*>(t->drop_seed);
        q.drop_p = t->drop_p;
        q.dzp = reinterpret_cast<char*>(scratch + B.dzp[l]);
        q.amax_dz = B.amax_dz[l] >= 0 ? scratch + B.amax_dz[l] : nullptr;
        q.tp_steps = tp_steps;
        q.wpt = (l >= 1 || dx) ? image + PL.wpt[l] : nullptr;
        q.da_prev = l >= 1 ? scratch + B.dz[cur ^ 1] : dx;
        if (l >= 1) { q.z_prev = ws + L.xhat[l - 1]; q.mean_prev = ws + L.mean[l - 1]; q.invstd_prev = ws + L.invstd[l - 1];
                      q.gamma_prev = t->bn_w[l - 1]; q.beta_prev = t->bn_b[l - 1]; q.part_out = part; }
        q.n_valid = nvp;
        PL_LAUNCH(np, bn_bwd_layer_kernel, cgrid, dim3(PL_NT), pl_lds_bytes(np), st, q);
        if (l >= 1) {
            const int rc = finish_sums(q.K, t->dbn_w[l - 1], t->dbn_b[l - 1]);
            if (rc != ABN_OK) return rc;
            cur ^= 1;
        }
    }
    }
    int n_wg = 0;
    WgradP w = make_wgrad(t, rows, L, B, ws, scratch, &n_wg);
    w.tp_steps = tp_steps;
    launch_wgrad(np, w, n_wg, st);
    ABN_CHECK_LAUNCH("tower_backward (BatchNorm, planes)");
    if (t->defer_reduce) return ABN_OK;               // abn_tower_reduce_step finishes the job (the slabs of THIS launch: psplits)
    const ReduceTable rt = make_reduce_table(t, B);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(grid_for((rt.total + 3) / 4)), dim3(256), 0, st, scratch + B.slabs, rt);
    ABN_CHECK_LAUNCH("slab_reduce");
    return ABN_OK;
}

// part: PLANES_BWD_ALL for the product entries; the measurement entry abn_tower_backward_launch
// issues one of the two launches (the other's output being in place from an earlier complete call).
enum { PLANES_BWD_ALL = 0, PLANES_BWD_DGRAD = 1, PLANES_BWD_WGRAD = 2 };
static int planes_backward(const abn_tower_desc* t, const float* d_out, const LossArgs* loss, int64_t rows, const Layout& L,
                           const BwdLayout& B, const float* ws, float* scratch, float* dx, hipStream_t st, int part = PLANES_BWD_ALL)
{
    const int nl = t->n_layers;
    const int np = planes_of(t);
    PlanesBwdP b = {};
#ifdef ABN_STAMPS
    b.stamps = getenv("ABN_DSTAMP_BUF") ? (unsigned long long*)strtoull(getenv("ABN_DSTAMP_BUF"), nullptr, 0) : nullptr;
#endif
    b.n_layers = nl;
    b.rows = (int)rows;
    b.d_out = d_out;
    b.d_out_is_dz = t->d_out_is_dz;
    b.tp_steps = pl_row_steps(rows);
    ABN_REQUIRE((loss || t->wgrad_part == 2 || aligned16(d_out)) && aligned16(scratch) && (!dx || aligned16(dx)),
                "tower_backward: d_out / scratch / dx must be 16-byte aligned");
    b.loss_kind = -1;
    b.B = (int)rows;                             // (no tower boundary inside the rows unless the loss rides along)
    if (loss) {
        b.loss_kind = loss->kind;
        b.y = loss->y; b.y_dtype = loss->y_dtype;
        b.B = (int)(rows / 2);
        b.margin = (double)loss->margin;
        b.scale = loss->avg ? 1.0 / (double)(rows / 2) : 1.0;
        b.loss_counter = reinterpret_cast<unsigned*>(loss->ws);
        b.loss_partial = reinterpret_cast<double*>(reinterpret_cast<char*>(loss->ws) + 8);
        b.loss_out = loss->loss_out;
        b.n_valid = loss->n_valid;
        b.loss_accum = loss->loss_accum;
    }
    b.a_top = ws + L.a[nl - 1];
    b.dx = dx;
    b.drop_seed = reinterpret_cast<const unsigned long long*>(t->drop_seed);
    b.drop_p = t->drop_p;
    const PackLayout PL = make_pack_layout(t);
    const char* const image = t->wpack ? reinterpret_cast<const char*>(t->wpack) : reinterpret_cast<const char*>(ws + L.wpack);
    for (int l = 0; l <= nl; ++l) b.dims[l] = (int)t->dims[l];
    for (int l = 0; l < nl; ++l) {
        b.act[l] = (l == nl - 1) ? t->last_act : t->act;
        b.tp[l] = reinterpret_cast<const char*>(ws + L.tp[l]);
        b.mask[l] = t->drop_mask[l];
        b.wpt[l] = image + PL.wpt[l];
        b.dzp[l] = reinterpret_cast<char*>(scratch + B.dzp[l]);
        b.amax_dz[l] = B.amax_dz[l] >= 0 ? scratch + B.amax_dz[l] : nullptr;
    }
    b.wbase = image; b.wbytes = PL.bytes;
    int l_first, l_end;
    { const int rc = wgrad_range(t, &l_first, &l_end); if (rc != ABN_OK) return rc; }
    if (t->wgrad_part == 2) part = PLANES_BWD_WGRAD;      // (the data-gradient launch ran with part 1)
    int n_wg = 0;
    const WgradP w = make_wgrad(t, rows, L, B, ws, scratch, &n_wg, l_first, l_end);
    static bool bw_attr_set[16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev = (dev >= 0 && dev < 16) ? dev : 0;
    if (!bw_attr_set[dev]) {
        PL_LDS_ATTR(tower_dgrad_planes_kernel, pl_lds_bytes);
        PL_LDS_ATTR(wgrad_planes_kernel, wgrad_lds_of);
        bw_attr_set[dev] = true;
    }
    const dim3 cgrid((unsigned)((rows + PL_ROWS - 1) / PL_ROWS));
    const bool do_dgrad = part != PLANES_BWD_WGRAD, do_wgrad = part != PLANES_BWD_DGRAD;
    if (do_dgrad) PL_LAUNCH(np, tower_dgrad_planes_kernel, cgrid, dim3(PL_NT), pl_lds_bytes(np), st, b);
    if (do_wgrad && n_wg > 0) launch_wgrad(np, w, n_wg, st);
    ABN_CHECK_LAUNCH("tower_backward (planes)");
    if (t->defer_reduce) return ABN_OK;      // abn_tower_reduce_step finishes the job
    ReduceTable rt = make_reduce_table(t, B);
    reduce_range(rt, t, B, l_first, l_end);
    if (rt.total > rt.begin)
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(grid_for((rt.total - rt.begin + 3) / 4)), dim3(256), 0, st, scratch + B.slabs, rt);
    ABN_CHECK_LAUNCH("slab_reduce");
    return ABN_OK;
}

// Backward of a forward that went through the layer-per-launch kernels (same predicate): per layer, top down,
// one wide_dgrad_layer_kernel (the top one forms dZ_top -- from d_out or from the pair loss); then every
// layer's weight gradient in the shared launch and the slab reduction.
// Small batches, one process: the weight gradients over all rows + the optimizer's rule as ONE launch of
// abn_tower_reduce_step's (tower_wgrad_step.h) instead of slabs in the backward and their sum there.  A pure function of
// the descriptor and the row count: the backward (which then launches no weight-gradient kernel) and the step ask here.
static bool wgrad_step_small_ok(const abn_tower_desc* t, int64_t rows, int64_t n_calls)
{
    if (!switches().wgrad_step || !t->defer_reduce || !t->fwd_ws || t->batch_norm || t->wgrad_part != 0 || planes_of(t) != 2) return false;
    if (n_calls < 1 || rows % n_calls != 0 || !aligned16(t->fwd_ws)) return false;
    const int64_t vrows = bn_vrows(rows, n_calls);
    if (wide_groups_for(t, vrows) <= 0 || !planes_shape_ok(t)) return false;
    return vrows <= 2048;                 // (a workgroup sums every row step itself: 128 of them at most)
}

static int wide_backward(const abn_tower_desc* t, const float* d_out, const LossArgs* loss, int64_t rows, int64_t n_calls,
                         const Layout& L, const BwdLayout& B, const float* ws, float* scratch, float* dx, hipStream_t st)
{
    const int nl = t->n_layers, top = nl - 1;
    const int np = planes_of(t);
    ABN_REQUIRE((loss || t->wgrad_part == 2 || aligned16(d_out)) && aligned16(scratch) && (!dx || aligned16(dx)),
                "tower_backward: d_out / scratch / dx must be 16-byte aligned");
    ABN_REQUIRE(!loss || n_calls == 2, "tower_backward_loss: two forward_once calls");
    const PackLayout PL = make_pack_layout(t);
    const char* const image = t->wpack ? reinterpret_cast<const char*>(t->wpack) : reinterpret_cast<const char*>(ws + L.wpack);
    static bool attr_set[16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev = (dev >= 0 && dev < 16) ? dev : 0;
    if (!attr_set[dev]) {
        PL_LDS_ATTR(wide_dgrad_layer_kernel, wd_lds_bytes);
        PL_LDS_ATTR(wgrad_planes_kernel, wgrad_lds_of);
        attr_set[dev] = true;
    }
    const int64_t rpc = rows / n_calls;
    const int64_t wpc = bn_wgs_per_call(rows, n_calls), nrb = n_calls * wpc;
    const int G = wide_groups_for(t, nrb * PL_ROWS);
    ABN_REQUIRE(G > 0, "tower_backward: too many rows for the layer-per-launch kernels");
    int l_first, l_end;
    { const int rc = wgrad_range(t, &l_first, &l_end); if (rc != ABN_OK) return rc; }
    const int last = dx ? 0 : 1;                      // the lowest layer a data-gradient launch runs for
    int cur = 0;
    for (int l = top; (l >= last || l == top) && t->wgrad_part != 2; --l) {
        WideBwdP q = {};
        q.l = l; q.top = top;
        q.N = (int)t->dims[l + 1]; q.K = (int)t->dims[l];
        q.act_prev = t->act;
        q.rows_call = (int)rpc; q.n_calls = (int)n_calls; q.wpc = (int)wpc;
        const int nblk = pl_blocks(q.K);
        q.G = G < nblk ? G : nblk;
        q.dz_in = l < top ? scratch + B.dz[cur] : nullptr;
        q.wpt = (l >= 1 || dx) ? image + PL.wpt[l] : nullptr;
        q.a_prev = l >= 1 ? ws + L.a[l - 1] : nullptr;
        q.dz_out = (l >= 1 && l - 1 >= last) ? scratch + B.dz[cur ^ 1] : nullptr;
        q.dzp_top = reinterpret_cast<char*>(scratch + B.dzp[top]);
        q.dzp_out = l >= 1 ? reinterpret_cast<char*>(scratch + B.dzp[l - 1]) : nullptr;
        if (np == 2) { q.amax_top = scratch + B.amax_dz[top]; q.amax_out = l >= 1 ? scratch + B.amax_dz[l - 1] : nullptr; }
        q.tp_steps = 2 * nrb;
        q.dx = dx;
        q.drop_seed = reinterpret_cast<const unsigned long long*>(t->drop_seed);
        q.drop_p = t->drop_p;
        if (l == top) {
            q.d_out = loss ? nullptr : d_out;
            q.d_out_is_dz = t->d_out_is_dz;
            q.a_top = ws + L.a[top];
            q.act_top = t->last_act;
            if (loss) {
                q.loss_kind = loss->kind; q.y_dtype = loss->y_dtype; q.y = loss->y;
                q.margin = (double)loss->margin;
                q.scale = loss->avg ? 1.0 / (double)rpc : 1.0;
                q.loss_counter = reinterpret_cast<unsigned*>(loss->ws);
                q.loss_partial = reinterpret_cast<double*>(reinterpret_cast<char*>(loss->ws) + 8);
                q.loss_out = loss->loss_out;
                q.n_valid = loss->n_valid;
                q.loss_accum = loss->loss_accum;
                if (t->source) {
                    q.y = t->source->labels;
                    q.g_steps = t->source->steps; q.g_ctr = t->source->step_ctr;
                }
            }
        }
        const dim3 grid((unsigned)(nrb * q.G));
        PL_LAUNCH(np, wide_dgrad_layer_kernel, grid, dim3(PL_NT), wd_lds_bytes(np), st, q);
        if (l < top || q.dz_out) cur ^= (q.dz_out ? 1 : 0);
        if (l == 0) break;
    }
    if (wgrad_step_small_ok(t, rows, n_calls)) {      // the weight gradients are abn_tower_reduce_step's, with the optimizer's rule
        ABN_REQUIRE(t->fwd_ws == ws && t->fwd_calls == n_calls, "tower_backward: abn_tower_desc.fwd_ws / fwd_calls are not this call's");
        ABN_CHECK_LAUNCH("tower_backward (layer per launch)");
        return ABN_OK;
    }
    int n_wg = 0;
    WgradP w = make_wgrad(t, rows, L, B, ws, scratch, &n_wg, l_first, l_end);
    w.tp_steps = 2 * nrb;
    if (n_wg > 0) launch_wgrad(np, w, n_wg, st);
    ABN_CHECK_LAUNCH("tower_backward (layer per launch)");
    if (t->defer_reduce) return ABN_OK;               // abn_tower_reduce_step finishes the job
    ReduceTable rt = make_reduce_table(t, B);
    reduce_range(rt, t, B, l_first, l_end);
    if (rt.total > rt.begin)
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(grid_for((rt.total - rt.begin + 3) / 4)), dim3(256), 0, st, scratch + B.slabs, rt);
    ABN_CHECK_LAUNCH("slab_reduce");
    return ABN_OK;
}

}  // namespace abn

using namespace abn;

extern "C" {

int abn_abi_version(void) { return ABN_ABI_VERSION; }
const char* abn_last_error(void) { return abn::g_err; }

int64_t abn_tower_ws_floats(const abn_tower_desc* t, int64_t rows, int64_t n_calls)
{
    if (check_desc(t, rows, n_calls) != ABN_OK) return -1;
    return make_layout(t, rows, n_calls).total;
}

int64_t abn_tower_out_offset(const abn_tower_desc* t, int64_t rows, int64_t n_calls)
{
    if (check_desc(t, rows, n_calls) != ABN_OK) return -1;
    return make_layout(t, rows, n_calls).a[t->n_layers - 1];
}

// Float offset of one of the operand-fragment images -- which = 0 packed W_l, 1 packed W_l^T, 2 transposed
// planes [input of layer l | 1] (all in the forward workspace), 3 transposed planes of dZ_l (in the backward
// scratch), 4 a BatchNorm layer's z_l, 5 the row-major output of layer l -- or -1 (see the header).
int64_t abn_tower_image_offset(const abn_tower_desc* t, int64_t rows, int64_t n_calls, int which, int l)
{
    if (check_desc(t, rows, n_calls) != ABN_OK || l < 0 || l >= t->n_layers) return -1;
    if (which == 3) return make_bwd_layout(t, rows).dzp[l];
    const Layout L = make_layout(t, rows, n_calls);
    if (which == 2) return L.tp[l];
    if (which == 4) return L.xhat[l];            // BatchNorm: z_l (bn_fwd_layer_kernel) / xhat_l (per-layer kernels)
    if (which == 5) return L.a[l];               // row-major output of layer l (layer-per-launch kernels: virtual rows; per-layer GEMMs)
    if (which != 0 && which != 1) return -1;
    const PackLayout P = make_pack_layout(t);
    // relative to abn_tower_desc.wpack when the caller keeps one, else to the workspace
    return (t->wpack ? 0 : L.wpack) + (which == 0 ? P.wp[l] : P.wpt[l]) / 4;
}

// Which kernels a call takes is a pure function of the descriptor, the row count, the pointers' alignment
// and the environment switches: the dispatchers below branch on these, and abn_tower_path reports them
// (ABN_PATH_* in the header).  No state is kept about past calls.
static bool fused_f32_ok(const abn_tower_desc* t, const float* x1, const float* x2, int64_t rows, int train, const float* ws)
{
    // Whole tower in one launch when it fits the fused kernel's LDS image (tower_fused.h): no BatchNorm
    // (its statistics span all rows), widths <= 512 and multiples of 4, 16-byte aligned tensors.
    // ABN_FUSED=0 forces the per-layer path (A/B measurements).  A fused workgroup walks its 32 rows
    // through every layer in ~110 us whatever the batch: it only pays once there are workgroups for most
    // CUs.  Below that the per-layer GEMMs (tiles over rows AND columns) are faster (measured: 4096 rows
    // 91 vs 116 us, 1024 rows 64 vs 107 us; 8192 rows 145 vs 131 us).
    const int64_t fused_min_rows = switches().fused_min_rows >= 0 ? switches().fused_min_rows : 6144;
    bool fusable = switches().fused && rows >= fused_min_rows && !t->batch_norm && aligned16(x1) && (!x2 || aligned16(x2)) &&
                   aligned16(ws);
    for (int l = 0; l <= t->n_layers && fusable; ++l)
        fusable = t->dims[l] >= 4 && t->dims[l] <= FUSED_MAXW && t->dims[l] % 4 == 0;
    for (int l = 0; l < t->n_layers && fusable; ++l)
        fusable = aligned16(t->W[l]) && (!t->drop_mask[l] || !train || aligned16(t->drop_mask[l]));
    return fusable;
}
static int forward_path(const abn_tower_desc* t, const float* x1, const float* x2, int64_t rows, int64_t n_calls, int train,
                        const float* ws)
{
    const bool bn_train = train && bn_train_planes_path(t, rows, n_calls, x1, x2, ws);
    const int kind = planes_kind(t, rows, n_calls, x1, x2, ws, train ? PLANES_TRAIN : PLANES_EVAL_FORWARD);
    if (!bn_train && kind == PLANES_WIDE) return ABN_PATH_WIDE;
    if (bn_train) return bn_persist_path(t, rows, n_calls) ? ABN_PATH_BN_TOWER : ABN_PATH_BN_LAYERS;
    if (kind != PLANES_NONE) {
        if (t->batch_norm) return ABN_PATH_PLANES_INFER_BN;
        return (t->forward_only && !(train && t->drop_seed) && !train) ? ABN_PATH_PLANES_INFER : ABN_PATH_PLANES;
    }
    return fused_f32_ok(t, x1, x2, rows, train, ws) ? ABN_PATH_FUSED_F32 : ABN_PATH_PER_LAYER;
}
static int backward_path(const abn_tower_desc* t, const float* x1, const float* x2, int64_t rows, int64_t n_calls, const float* ws)
{
    const int kind = planes_kind(t, rows, n_calls, x1, x2, ws);
    if (kind == PLANES_WIDE) return ABN_PATH_WIDE;
    if (kind == PLANES_CHAIN) return ABN_PATH_PLANES;
    if (bn_train_planes_path(t, rows, n_calls, x1, x2, ws)) return bn_persist_path(t, rows, n_calls) ? ABN_PATH_BN_TOWER : ABN_PATH_BN_LAYERS;
    return ABN_PATH_PER_LAYER;
}

int abn_tower_path(const abn_tower_desc* t, const float* x1, const float* x2, int64_t rows, int64_t n_calls, int train,
                   const float* ws, int backward, int32_t* precision_out)
{
    if (check_desc(t, rows, n_calls) != ABN_OK) return -1;
    const int path = backward ? backward_path(t, x1, x2, rows, n_calls, ws) : forward_path(t, x1, x2, rows, n_calls, train, ws);
    // fp16 x 2 exists on the operand planes only: the GEMM kernels run such a tower in bf16 x 3
    if (precision_out) *precision_out = (path == ABN_PATH_PER_LAYER || path == ABN_PATH_FUSED_F32) ? gemm_prec(t->precision) : t->precision;
    return path;
}

int abn_tower_uses_planes(const abn_tower_desc* t, int64_t rows, const float* x1, const float* x2, const float* ws, int train)
{
    if (check_desc(t, rows, 1) != ABN_OK) return -1;
    if (train && t->batch_norm) return bn_train_planes_path(t, rows, 1, x1, x2, ws) ? 1 : 0;
    return planes_path(t, rows, x1, x2, ws, train ? PLANES_TRAIN : PLANES_EVAL_FORWARD) ? 1 : 0;
}

int64_t abn_tower_sync_ws_bytes(void) { return bnp_sync_bytes(BNP_MAX_WGS); }

int64_t abn_tower_wpack_floats(const abn_tower_desc* t)
{
    if (check_desc(t, 0, 1) != ABN_OK) return -1;
    return make_pack_layout(t).bytes / 4;
}

int64_t abn_tower_bwd_scratch_floats(const abn_tower_desc* t, int64_t rows)
{
    if (check_desc(t, rows, 1) != ABN_OK) return -1;
    return make_bwd_layout(t, rows).total;
}

int abn_tower_forward(const abn_tower_desc* t, const float* x1, const float* x2, int64_t rows,
                      int64_t n_calls, int train, float* ws, void* stream)
{
    int rc = check_desc(t, rows, n_calls);
    if (rc != ABN_OK) return rc;
    ABN_REQUIRE(x1 && ws, "tower_forward: null input/workspace");
    ABN_REQUIRE(x2 == nullptr || n_calls == 2, "tower_forward: x2 given but n_calls=%lld", (long long)n_calls);
    ABN_REQUIRE(n_calls <= 8, "tower_forward: at most 8 calls per launch");
    if (rows == 0) return ABN_OK;
    hipStream_t st = (hipStream_t)stream;
    const Layout L = make_layout(t, rows, n_calls);
    const int64_t rpc = rows / n_calls;

    const int path = forward_path(t, x1, x2, rows, n_calls, train, ws);
    const bool fusable = path == ABN_PATH_FUSED_F32;
    const int pmode = train ? PLANES_TRAIN : PLANES_EVAL_FORWARD;
    const bool bn_train = train && bn_train_planes_path(t, rows, n_calls, x1, x2, ws);
    if (train && t->batch_norm && bn_sync_on(t) && !bn_train) {
        set_error("tower_forward: cross-replica BatchNorm statistics (bn_sync_world) need the operand-plane launches");
        return ABN_E_UNSUPPORTED;
    }
    if (train && t->batch_norm && t->n_valid && !bn_train) {
        set_error("tower_forward: a padded batch (n_valid) through a BatchNorm tower in training needs the BatchNorm layer launches");
        return ABN_E_UNSUPPORTED;
    }
    const int kind = planes_kind(t, rows, n_calls, x1, x2, ws, pmode);
    if (t->source && !(train && !bn_train && kind == PLANES_WIDE && n_calls == 2 && x2 && !t->forward_only)) {
        set_error("tower_forward: a step source (abn_tower_desc.source) needs the layer-per-launch kernels in training, two calls");
        return ABN_E_UNSUPPORTED;
    }
    if (t->source) ABN_REQUIRE(t->source->table && t->source->idx1 && t->source->idx2 && t->source->steps && t->source->step_ctr &&
                                   aligned16(t->source->table) && t->dims[0] % 4 == 0,
                               "tower_forward: abn_step_source: null or misaligned array");
    if (t->drop_seed && train && !bn_train && kind == PLANES_NONE) {
        for (int l = 0; l < t->n_layers; ++l)
            if (!t->drop_mask[l]) { set_error("tower_forward: in-kernel dropout (drop_seed) needs the operand-plane kernels: pass drop_mask tensors"); return ABN_E_UNSUPPORTED; }
    }
    if (bn_train || kind != PLANES_NONE) {
        const int np = planes_of(t);
        PackTable pk = {};
        PlanesFwdP f = {};
        f.n_layers = t->n_layers;
        f.rows = (int)rows;
        f.rows_call = (int)rpc;
        f.x1 = x1; f.x2 = x2;
        f.x_copy = nullptr;                      // the planes backward reads the transposed images only
        const PackLayout PL = make_pack_layout(t);
        char* const image = t->wpack ? reinterpret_cast<char*>(t->wpack) : reinterpret_cast<char*>(ws + L.wpack);
        const bool repack = !(t->wpack && t->wpack_valid);
        ABN_REQUIRE(aligned16(image), "tower_forward: wpack must be 16-byte aligned");
        pk.base = image;
        f.wbase = image; f.wbytes = PL.bytes;
        for (int l = 0; l <= t->n_layers; ++l) f.dims[l] = (int)t->dims[l];
        for (int l = 0; l < t->n_layers; ++l) {
            f.act[l] = (l == t->n_layers - 1) ? t->last_act : t->act;
            f.b[l] = t->b[l];
            f.mask[l] = train ? t->drop_mask[l] : nullptr;
            f.out[l] = l == t->n_layers - 1 ? ws + L.a[l] : nullptr;     // hidden activations live in tp[l + 1] only
            f.wp[l] = image + PL.wp[l];
            PackJob& J = pk.job[pk.n_jobs++];
            J.W = t->W[l]; J.N = (int)t->dims[l + 1]; J.K = (int)t->dims[l]; J.transposed = 0;
            J.nblk = pl_blocks(J.N); J.nsteps = pl_steps(J.K);
            J.tile0 = pk.n_tiles; J.dst = PL.wp[l];
            J.blk0 = pk.n_blocks;
            pk.n_tiles += J.nblk * J.nsteps;
            pk.n_blocks += J.nblk;
            {                                    // W_l^T for the backward's data-gradient chain (l = 0: d loss / d input)
                PackJob& T = pk.job[pk.n_jobs++];
                T.W = t->W[l]; T.N = J.N; T.K = J.K; T.transposed = 1;
                T.nblk = pl_blocks(T.K); T.nsteps = pl_steps(T.N);
                T.tile0 = pk.n_tiles; T.dst = PL.wpt[l];
                T.blk0 = pk.n_blocks;
                pk.n_tiles += T.nblk * T.nsteps;
                pk.n_blocks += T.nblk;
            }
            f.tp[l] = (t->forward_only || t->batch_norm) ? nullptr : reinterpret_cast<char*>(ws + L.tp[l]);     // (inference: nothing kept for a backward)
            f.amax[l] = f.tp[l] && L.amax[l] >= 0 ? ws + L.amax[l] : nullptr;
            if (t->batch_norm) { f.bn_rm[l] = t->bn_rm[l]; f.bn_rv[l] = t->bn_rv[l]; f.bn_w[l] = t->bn_w[l]; f.bn_b[l] = t->bn_b[l]; }
        }
        f.tp_steps = pl_row_steps(rows);
        f.drop_seed = train ? reinterpret_cast<const unsigned long long*>(t->drop_seed) : nullptr;
        f.drop_p = t->drop_p;
        f.bn_eps = BN_EPS;
#ifdef ABN_STAMPS
        f.stamps = getenv("ABN_STAMP_BUF") ? (unsigned long long*)strtoull(getenv("ABN_STAMP_BUF"), nullptr, 0) : nullptr;
#endif
        static bool pl_attr_set[16] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        dev = (dev >= 0 && dev < 16) ? dev : 0;
        const dim3 pgrid((unsigned)((pk.n_tiles + 3) / 4));
        const dim3 fgrid((unsigned)((rows + PL_ROWS - 1) / PL_ROWS));
        if (repack) {
            if (np == 3) hipLaunchKernelGGL(pack_planes_kernel<3>, pgrid, dim3(256), 0, st, pk);
            else if (np == 2) hipLaunchKernelGGL(pack_planes_scaled_kernel, dim3((unsigned)pk.n_blocks), dim3(PL_NT), 0, st, pk);
            else hipLaunchKernelGGL(pack_planes_kernel<1>, pgrid, dim3(256), 0, st, pk);
        }
        if (!bn_train && kind == PLANES_WIDE) {
            static bool wd_attr_set[16] = {};
            if (!wd_attr_set[dev]) {
                PL_LDS_ATTR(wide_fwd_layer_kernel, wd_lds_bytes);
                wd_attr_set[dev] = true;
            }
            const int nl = t->n_layers;
            const int64_t wpc = bn_wgs_per_call(rows, n_calls), nrb = n_calls * wpc;
            const int G = wide_groups_for(t, nrb * PL_ROWS);
            const bool keep = !t->forward_only;              // transposed images for a backward
            for (int l = 0; l < nl; ++l) {
                WideFwdP q = {};
                q.l = l; q.last = l == nl - 1;
                q.K = (int)t->dims[l]; q.N = (int)t->dims[l + 1];
                q.act = f.act[l];
                q.rows_call = (int)rpc; q.n_calls = (int)n_calls; q.wpc = (int)wpc;
                const int nblk = pl_blocks(q.N);
                q.G = G < nblk ? G : nblk;
                q.x1 = x1; q.x2 = x2;
                q.a_prev = l >= 1 ? ws + L.a[l - 1] : nullptr;
                q.wp = f.wp[l];
                q.b = t->b[l];
                q.a_out = q.last ? nullptr : ws + L.a[l];
                q.out = q.last ? ws + L.a[l] : nullptr;
                q.tp_in = keep && l == 0 ? reinterpret_cast<char*>(ws + L.tp[0]) : nullptr;
                q.tp_out = keep && !q.last ? reinterpret_cast<char*>(ws + L.tp[l + 1]) : nullptr;
                if (np == 2 && keep) { q.amax_in = ws + L.amax[0]; q.amax_out = q.last ? nullptr : ws + L.amax[l + 1]; }
                q.tp_steps = 2 * nrb;
                q.drop_seed = f.drop_seed; q.drop_p = f.drop_p;
                if (t->source && l == 0) {
                    q.g_table = t->source->table; q.g_rows = t->source->table_rows;
                    q.g_idx1 = t->source->idx1; q.g_idx2 = t->source->idx2;
                    q.g_steps = t->source->steps; q.g_ctr = t->source->step_ctr;
                }
#ifdef ABN_STAMPS
                q.stamps = getenv("ABN_STAMP_BUF") ? (unsigned long long*)strtoull(getenv("ABN_STAMP_BUF"), nullptr, 0) : nullptr;
#endif
                const dim3 wgrid((unsigned)(nrb * q.G));
                PL_LAUNCH(np, wide_fwd_layer_kernel, wgrid, dim3(PL_NT), wd_lds_bytes(np), st, q);
            }
            ABN_CHECK_LAUNCH("tower_forward (layer per launch)");
            return ABN_OK;
        }
        if (bn_train) {
            static bool bn_attr_set[16] = {};
            if (!bn_attr_set[dev]) {
                PL_LDS_ATTR(bn_fwd_layer_kernel, pl_lds_bytes);
                bn_attr_set[dev] = true;
            }
            f.bn_part = ws + L.bn_wg;
            const int nl = t->n_layers;
            const int64_t wpc = bn_wgs_per_call(rows, n_calls);
            const dim3 bgrid((unsigned)(n_calls * wpc));
            f.tp_steps = 2 * n_calls * wpc;                    // (the images' row axis is padded per call)
            if (bn_persist_path(t, rows, n_calls)) {
                // every layer in ONE resident launch, grid barriers in between (tower_bn_persist.h)
                static bool bt_attr_set[16] = {};
                if (!bt_attr_set[dev]) {
                    PL_LDS_ATTR(bn_fwd_tower_kernel, bnp_lds_bytes);
                    bt_attr_set[dev] = true;
                }
                PlanesFwdP fl = f;
                BnPersistP q = {};
                for (int l = 0; l < nl; ++l) {
                    fl.tp[l] = reinterpret_cast<char*>(ws + L.tp[l]);        // [a_{l-1} | 1] transposed: the weight gradient's operand
                    fl.amax[l] = L.amax[l] >= 0 ? ws + L.amax[l] : nullptr;
                    fl.out[l] = nullptr;
                    q.z[l] = ws + L.xhat[l];
                    q.mean[l] = ws + L.mean[l]; q.invstd[l] = ws + L.invstd[l]; q.var[l] = ws + L.var[l];
                    q.rm[l] = t->bn_rm[l]; q.rv[l] = t->bn_rv[l];
                    q.nbt[l] = reinterpret_cast<long long*>(t->bn_nbt[l]);
                }
                q.sync_ws = t->sync_ws;
                q.sync_bytes = bnp_sync_bytes(BNP_MAX_WGS);
                q.n_valid = t->n_valid;
                q.n_calls = (int)n_calls;
                q.a_top = ws + L.a[nl - 1];
                PL_LAUNCH(np, bn_fwd_tower_kernel, bgrid, dim3(PL_NT), bnp_lds_bytes(np), st, fl, q);
                ABN_CHECK_LAUNCH("tower_forward (BatchNorm, resident tower)");
                return ABN_OK;
            }
            for (int l = 0; l < nl; ++l) {
                PlanesFwdP fl = f;
                for (int i = 0; i < nl; ++i) { fl.tp[i] = nullptr; fl.out[i] = nullptr; }
                fl.tp[l] = reinterpret_cast<char*>(ws + L.tp[l]);        // [a_{l-1} | 1] transposed: the weight gradient's operand
                fl.amax[l] = L.amax[l] >= 0 ? ws + L.amax[l] : nullptr;
                fl.act[l] = ACT_NONE;                          // z_l leaves the launch as it is; act[l - 1] is applied on the way in
                fl.out[l] = ws + L.xhat[l];                    // (z lands where xhat will live)
                BnTrainP q = {};
                q.l = l;
                if (l > 0) { q.mean = ws + L.mean[l - 1]; q.invstd = ws + L.invstd[l - 1]; q.z_prev = ws + L.xhat[l - 1]; q.a_prev = nullptr; }
                q.n_valid = t->n_valid;
                PL_LAUNCH(np, bn_fwd_layer_kernel, bgrid, dim3(PL_NT), pl_lds_bytes(np), st, fl, q);
                const int N = (int)t->dims[l + 1];
                const bool sync = bn_sync_on(t);
                ABN_REQUIRE(!sync || t->bn_sync_fn, "tower_forward: bn_sync_world = %d without bn_sync_fn", t->bn_sync_world);
                if (sync && t->n_valid) { set_error("tower_forward: n_valid (a padded batch) cannot be combined with cross-replica BatchNorm statistics"); return ABN_E_UNSUPPORTED; }
                double* const sums = reinterpret_cast<double*>(ws + L.bn_part);       // (the per-layer kernels' partials: idle here)
                hipLaunchKernelGGL(bn_stats_finish_wg_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64 * BN_WG_GROUPS), 0, st, ws + L.bn_wg,
                                   (int)wpc, rpc, N, (int)n_calls, ws + L.mean[l], ws + L.invstd[l], ws + L.var[l],
                                   t->bn_rm[l], t->bn_rv[l], sync ? sums : static_cast<double*>(nullptr), t->n_valid);
                if (sync) {
                    if (t->bn_sync_fn(t->bn_sync_ctx, sums, n_calls * 2 * N + n_calls, st) != 0) { set_error("tower_forward: bn_sync_fn failed"); return ABN_E_LAUNCH; }
                    hipLaunchKernelGGL(bn_stats_from_sums_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, sums,
                                       N, (int)n_calls, ws + L.mean[l], ws + L.invstd[l], ws + L.var[l],
                                       t->bn_rm[l], t->bn_rv[l], ws + L.bn_nstat + 8 * l);
                }
            }
            const int N = (int)t->dims[nl];
            const float* z = ws + L.xhat[nl - 1];      // (stays un-normalised, like every layer's: the backward normalises again)
            hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(rows * N)), dim3(256), 0, st, z, rows, rpc, N, ws + L.mean[nl - 1],
                               ws + L.invstd[nl - 1], t->bn_rm[nl - 1], t->bn_rv[nl - 1], 1, t->bn_w[nl - 1], t->bn_b[nl - 1],
                               t->last_act, static_cast<float*>(nullptr), ws + L.a[nl - 1], t->n_valid);
            launch_nbt(t, n_calls, st);
            ABN_CHECK_LAUNCH("tower_forward (BatchNorm, planes)");
            return ABN_OK;
        }
        // inference (no mask, no seed, nothing kept for a backward) has its own, lighter instantiations
        const bool infer = t->batch_norm || (t->forward_only && !f.drop_seed && !train);
        const int mode = t->batch_norm ? PL_INFER_BN : infer ? PL_INFER : PL_TRAIN;
        const size_t lds = t->batch_norm ? pl_lds_bytes_bn(np) : pl_lds_bytes_stag(np);      // (the layers may stagger: two operand images)
        const void* kernels[3][3] = {
            {reinterpret_cast<const void*>(tower_fwd_planes_kernel<1, PL_TRAIN>), reinterpret_cast<const void*>(tower_fwd_planes_kernel<1, PL_INFER>),
             reinterpret_cast<const void*>(tower_fwd_planes_kernel<1, PL_INFER_BN>)},
            {reinterpret_cast<const void*>(tower_fwd_planes_kernel<2, PL_TRAIN>), reinterpret_cast<const void*>(tower_fwd_planes_kernel<2, PL_INFER>),
             reinterpret_cast<const void*>(tower_fwd_planes_kernel<2, PL_INFER_BN>)},
            {reinterpret_cast<const void*>(tower_fwd_planes_kernel<3, PL_TRAIN>), reinterpret_cast<const void*>(tower_fwd_planes_kernel<3, PL_INFER>),
             reinterpret_cast<const void*>(tower_fwd_planes_kernel<3, PL_INFER_BN>)}};
        if (!pl_attr_set[dev]) {
            for (int a = 0; a < 3; ++a)
                for (int m = 0; m < 3; ++m)
                    if (kernels[a][m])
                        (void)hipFuncSetAttribute(kernels[a][m], hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  (int)(m == PL_INFER_BN ? pl_lds_bytes_bn(a + 1) : pl_lds_bytes_stag(a + 1)));
            pl_attr_set[dev] = true;
        }
        void* kargs[] = {&f};
        (void)hipLaunchKernel(kernels[np - 1][mode], fgrid, dim3(PL_NT), kargs, lds, st);
        ABN_CHECK_LAUNCH("tower_fwd_planes");
        return ABN_OK;
    }
    if (fusable) {
        FusedFwdP f = {};
        f.n_layers = t->n_layers;
        f.rows = (int)rows;
        f.rows_call = (int)rpc;
        f.bf16 = gemm_prec(t->precision);
        f.x1 = x1; f.x2 = x2;
        f.x_copy = x2 ? ws + L.x : nullptr;
        for (int l = 0; l <= t->n_layers; ++l) f.dims[l] = (int)t->dims[l];
        for (int l = 0; l < t->n_layers; ++l) {
            f.act[l] = (l == t->n_layers - 1) ? t->last_act : t->act;
            f.W[l] = t->W[l]; f.b[l] = t->b[l];
            f.mask[l] = train ? t->drop_mask[l] : nullptr;
            f.out[l] = ws + L.a[l];
        }
#ifdef ABN_STAMPS
        f.stamps = getenv("ABN_STAMP_BUF") ? (unsigned long long*)strtoull(getenv("ABN_STAMP_BUF"), nullptr, 0) : nullptr;
#endif
        static bool attr_set[16] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        dev = (dev >= 0 && dev < 16) ? dev : 0;
        if (!attr_set[dev]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tower_fwd_fused_kernel<0>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)FUSED_LDS_BYTES);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tower_fwd_fused_kernel<1>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)FUSED_LDS_BYTES);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tower_fwd_fused_kernel<2>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)FUSED_LDS_BYTES);
            attr_set[dev] = true;
        }
        const dim3 fgrid((unsigned)((rows + FUSED_ROWS - 1) / FUSED_ROWS));
        if (f.bf16 == 1) hipLaunchKernelGGL(tower_fwd_fused_kernel<1>, fgrid, dim3(FUSED_NT), FUSED_LDS_BYTES, st, f);
        else if (f.bf16 == 2) hipLaunchKernelGGL(tower_fwd_fused_kernel<2>, fgrid, dim3(FUSED_NT), FUSED_LDS_BYTES, st, f);
        else hipLaunchKernelGGL(tower_fwd_fused_kernel<0>, fgrid, dim3(FUSED_NT), FUSED_LDS_BYTES, st, f);
        ABN_CHECK_LAUNCH("tower_fwd_fused");
        return ABN_OK;
    }

    const float* in = x1;
    if (x2) {    // the two towers' inputs become one [2B, D] operand
        const size_t half = (size_t)rpc * t->dims[0] * sizeof(float);
        if (hipMemcpyAsync(ws + L.x, x1, half, hipMemcpyDeviceToDevice, st) != hipSuccess ||
            hipMemcpyAsync(ws + L.x + rpc * t->dims[0], x2, half, hipMemcpyDeviceToDevice, st) != hipSuccess) {
            set_error("tower_forward: input copy failed");
            return ABN_E_LAUNCH;
        }
        in = ws + L.x;
    }
    for (int l = 0; l < t->n_layers; ++l) {
        const int K = (int)t->dims[l], N = (int)t->dims[l + 1];
        const int act = (l == t->n_layers - 1) ? t->last_act : t->act;
        float* a = ws + L.a[l];
        GemmP p = {};
        p.A = in; p.lda = K;
        p.B = t->W[l]; p.ldb = K;
        p.M = (int)rows; p.N = N; p.K = K; p.k_chunk = K;
        p.bias = t->b[l];
        p.mask = train ? t->drop_mask[l] : nullptr;
        p.a_vec = aligned16(in) && (K % 4 == 0);
        p.b_vec = aligned16(t->W[l]) && (K % 4 == 0);
        p.ones_col = -1;
        p.bf16 = gemm_prec(t->precision);
        if (!t->batch_norm) {
            p.C = a; p.ldc = N; p.act = act;
            rc = launch_gemm<true, true, EPI_FWD>(p, 1, st);
            if (rc != ABN_OK) return rc;
        } else {
            float* z = ws + L.xhat[l];          // z lands where xhat will live
            p.C = z; p.ldc = N; p.act = ACT_NONE;
            rc = launch_gemm<true, true, EPI_FWD>(p, 1, st);
            if (rc != ABN_OK) return rc;
            if (train) {
                const int nch = bn_chunks(rpc);
                double* part = reinterpret_cast<double*>(ws + L.bn_part);
                hipLaunchKernelGGL(bn_partial_kernel<false>, dim3((N + 63) / 64, nch, (int)n_calls), dim3(256), 0, st, z,
                                   nullptr, nullptr, rpc, N, bn_chunk_rows(rpc), 0, part);
                hipLaunchKernelGGL(bn_stats_finish_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, st, part, nch, rpc,
                                   N, (int)n_calls, ws + L.mean[l], ws + L.invstd[l], ws + L.var[l], t->bn_rm[l],
                                   t->bn_rv[l]);
            }
            hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(rows * N)), dim3(256), 0, st, z, rows, rpc, N,
                               ws + L.mean[l], ws + L.invstd[l], t->bn_rm[l], t->bn_rv[l], train, t->bn_w[l],
                               t->bn_b[l], act, z, a);
            ABN_CHECK_LAUNCH("batch_norm forward");
        }
        in = a;
    }
    if (train && t->batch_norm) { launch_nbt(t, n_calls, st); ABN_CHECK_LAUNCH("batch_norm counters"); }
    return ABN_OK;
}

int abn_tower_backward(const abn_tower_desc* t, const float* x1, const float* x2, const float* d_out,
                       int64_t rows, int64_t n_calls, const float* ws, float* scratch,
                       int64_t scratch_floats, float* dx, void* stream)
{
    int rc = check_desc(t, rows, n_calls);
    if (rc != ABN_OK) return rc;
    ABN_REQUIRE(x1 && (d_out || t->wgrad_part == 2) && ws && scratch, "tower_backward: null pointer");      // (part 2 reads no d_out)
    for (int l = 0; l < t->n_layers; ++l) {
        ABN_REQUIRE(t->dW[l] && t->db[l], "tower_backward: layer %d has null gradient buffers", l);
        if (t->batch_norm) ABN_REQUIRE(t->dbn_w[l] && t->dbn_b[l], "tower_backward: layer %d has null BN gradient buffers", l);
    }
    if (rows == 0) return ABN_OK;
    hipStream_t st = (hipStream_t)stream;
    const Layout L = make_layout(t, rows, n_calls);
    const BwdLayout B = make_bwd_layout(t, rows);
    if (scratch_floats < B.total) {
        set_error("tower_backward: scratch too small (%lld < %lld floats)", (long long)scratch_floats, (long long)B.total);
        return ABN_E_WORKSPACE;
    }
    const int64_t rpc = rows / n_calls;
    const float* xin = x2 ? ws + L.x : x1;
    const int nl = t->n_layers;

    // The forward that filled `ws` went through the planes kernels (same predicate): its workspace
    // holds W^T and the weight-gradient operands as operand fragments.  Two launches: the data
    // gradient chain (one workgroup per 32 rows, all layers), then every layer's weight gradient.
    ABN_REQUIRE(!t->source, "tower_backward: a step source (abn_tower_desc.source) goes with abn_tower_backward_loss (the labels are the plan's)");
    {
        const int kind = planes_kind(t, rows, n_calls, x1, x2, ws);
        if (kind == PLANES_WIDE) { return wide_backward(t, d_out, nullptr, rows, n_calls, L, B, ws, scratch, dx, st); }
        if (kind == PLANES_CHAIN) { return planes_backward(t, d_out, nullptr, rows, L, B, ws, scratch, dx, st); }
    }
    if (t->wgrad_part != 0) {
        set_error("tower_backward: wgrad_part needs the operand-plane launches of a tower without BatchNorm");
        return ABN_E_UNSUPPORTED;
    }
    if (bn_train_planes_path(t, rows, n_calls, x1, x2, ws)) { return bn_planes_backward(t, d_out, nullptr, rows, n_calls, L, B, ws, scratch, dx, st); }
    if (t->batch_norm && bn_sync_on(t)) {
        set_error("tower_backward: cross-replica BatchNorm statistics (bn_sync_world) need the operand-plane launches");
        return ABN_E_UNSUPPORTED;
    }
    if (t->batch_norm && t->n_valid) {
        set_error("tower_backward: a padded batch (n_valid) through a BatchNorm tower needs the BatchNorm layer launches");
        return ABN_E_UNSUPPORTED;
    }

    int cur = 0;
    const float* dz_in = nullptr;                // the output layer's dz when the caller supplied it

    // dz of the output layer from d_out
    {
        const int N = (int)t->dims[nl];
        const float* a = ws + L.a[nl - 1];
        float* dz = scratch + B.dz[cur];
        if (t->d_out_is_dz) {
            ABN_REQUIRE(!t->batch_norm, "tower_backward: d_out_is_dz cannot be combined with batch_norm");
            dz_in = d_out;                       // abn_pair_loss_dz already applied act' and the dropout mask
        } else if (!t->batch_norm) {
            hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(rows * N)), dim3(256), 0, st, a, d_out,
                               t->drop_mask[nl - 1], dz, rows * N, t->last_act);
        } else {
            const int nch = bn_chunks(rpc);
            double* part = reinterpret_cast<double*>(scratch + B.bn_part);
            hipLaunchKernelGGL(bn_partial_kernel<true>, dim3((N + 63) / 64, nch, (int)n_calls), dim3(256), 0, st, d_out, a,
                               ws + L.xhat[nl - 1], rpc, N, bn_chunk_rows(rpc), t->last_act, part);
            hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3((unsigned)((n_calls * N + 63) / 64)), dim3(64), 0, st, part,
                               nch, N, (int)n_calls, scratch + B.bn_s1, scratch + B.bn_s2);
            hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(rows * N)), dim3(256), 0, st, a, d_out,
                               ws + L.xhat[nl - 1], rows, rpc, N, t->last_act, t->bn_w[nl - 1],
                               ws + L.invstd[nl - 1], scratch + B.bn_s1, scratch + B.bn_s2, (int)n_calls,
                               t->drop_mask[nl - 1], dz, t->dbn_w[nl - 1], t->dbn_b[nl - 1]);
        }
        ABN_CHECK_LAUNCH("output-layer dz");
    }

    float* slabs = scratch + B.slabs;
    for (int l = nl - 1; l >= 0; --l) {
        const int Kin = (int)t->dims[l], Nout = (int)t->dims[l + 1];
        const float* dz = (l == nl - 1 && dz_in) ? dz_in : scratch + B.dz[cur];
        const float* a_in = (l == 0) ? xin : ws + L.a[l - 1];
        // wgrad: dW[Nout, Kin] (+ db via the ones column) = dz^T a_in, split over rows
        GemmP pw = {};
        pw.A = dz; pw.lda = Nout;
        pw.B = a_in; pw.ldb = Kin;
        pw.C = slabs + B.off[l]; pw.ldc = Kin;
        pw.C2 = slabs + B.off[l] + (int64_t)Nout * Kin;
        pw.slab_stride = B.slab_stride;
        pw.M = Nout; pw.N = Kin + 1; pw.K = (int)rows;
        pw.k_chunk = (int)align_up((rows + B.splits[l] - 1) / B.splits[l], BK);
        pw.ones_col = Kin;
        pw.bf16 = gemm_prec(t->precision);
        pw.a_vec = aligned16(dz) && (Nout % 4 == 0);
        pw.b_vec = aligned16(a_in) && (Kin % 4 == 0);
        // slices past the end of the reduction write zero slabs (k range empty)
        if (!(l > 0 || dx)) {
            rc = launch_gemm<false, false, EPI_WGRAD>(pw, B.splits[l], st);
            if (rc != ABN_OK) return rc;
        }
        // dgrad: d a_{l-1} = dz W_l, times act'(a_{l-1}) when no BN sits in between.  Both
        // GEMMs read dz only: they go out as one grid (launch_bwd_pair).
        if (l > 0 || dx) {
            float* dst = (l == 0) ? dx : scratch + B.dz[cur ^ 1];
            GemmP p = {};
            p.A = dz; p.lda = Nout;
            p.B = t->W[l]; p.ldb = Kin;
            p.C = dst; p.ldc = Kin;
            p.M = (int)rows; p.N = Kin; p.K = Nout; p.k_chunk = Nout;
            p.a_vec = aligned16(dz) && (Nout % 4 == 0);
            p.b_vec = aligned16(t->W[l]) && (Kin % 4 == 0);
            p.ones_col = -1;
            p.bf16 = gemm_prec(t->precision);
            if (l > 0 && !t->batch_norm) { p.aux = ws + L.a[l - 1]; p.ldaux = Kin; p.act = t->act; p.mask = t->drop_mask[l - 1]; }
            rc = launch_bwd_pair(pw, B.splits[l], p, st);
            if (rc != ABN_OK) return rc;
            if (l > 0 && t->batch_norm) {
                // dst holds d a_{l-1}; turn it into d z_{l-1} through act' and BN
                const float* a = ws + L.a[l - 1];
                const int nch = bn_chunks(rpc);
                double* part = reinterpret_cast<double*>(scratch + B.bn_part);
                hipLaunchKernelGGL(bn_partial_kernel<true>, dim3((Kin + 63) / 64, nch, (int)n_calls), dim3(256), 0, st,
                                   dst, a, ws + L.xhat[l - 1], rpc, Kin, bn_chunk_rows(rpc), t->act, part);
                hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3((unsigned)((n_calls * Kin + 63) / 64)), dim3(64), 0, st,
                                   part, nch, Kin, (int)n_calls, scratch + B.bn_s1, scratch + B.bn_s2);
                hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(rows * Kin)), dim3(256), 0, st, a, dst,
                                   ws + L.xhat[l - 1], rows, rpc, Kin, t->act, t->bn_w[l - 1], ws + L.invstd[l - 1],
                                   scratch + B.bn_s1, scratch + B.bn_s2, (int)n_calls, t->drop_mask[l - 1], dst,
                                   t->dbn_w[l - 1], t->dbn_b[l - 1]);
                ABN_CHECK_LAUNCH("batch_norm backward");
            }
            cur ^= 1;
        }
    }
    if (t->defer_reduce) return ABN_OK;          // abn_tower_reduce_step finishes the job
    const ReduceTable rt = make_reduce_table(t, B);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(grid_for((rt.total + 3) / 4)), dim3(256), 0, st, slabs, rt);
    ABN_CHECK_LAUNCH("slab_reduce");
    return ABN_OK;
}

// ONE of the two launches of the operand-plane backward -- part 1 the data-gradient chain, 2 the weight gradients --
// after a complete backward with the same arguments has left the other launch's output in place.  Never reduces
// the slabs (bench.py's per-launch timings; see the header).
int abn_tower_backward_launch(const abn_tower_desc* t, const float* x1, const float* x2, const float* d_out,
                                  int64_t rows, int64_t n_calls, const float* ws, float* scratch,
                                  int64_t scratch_floats, int part, void* stream)
{
    int rc = check_desc(t, rows, n_calls);
    if (rc != ABN_OK) return rc;
    ABN_REQUIRE(x1 && d_out && ws && scratch && (part == PLANES_BWD_DGRAD || part == PLANES_BWD_WGRAD), "tower_backward_launch: bad argument");
    if (rows == 0 || planes_kind(t, rows, n_calls, x1, x2, ws) != PLANES_CHAIN) { set_error("tower_backward_launch: single-launch operand-plane towers only"); return ABN_E_UNSUPPORTED; }
    const Layout L = make_layout(t, rows, n_calls);
    const BwdLayout B = make_bwd_layout(t, rows);
    if (scratch_floats < B.total) { set_error("tower_backward_launch: scratch too small"); return ABN_E_WORKSPACE; }
    abn_tower_desc u = *t;
    u.defer_reduce = 1;
    return planes_backward(&u, d_out, nullptr, rows, L, B, ws, scratch, nullptr, (hipStream_t)stream, part);
}

void abn_reload_switches(void) { abn::reload_switches(); }

// abn_tower_desc.bn_sync_fn over RCCL without a host language in between (include/abnet3_hip.h): the caller's communicator and
// the address of ITS ncclAllReduce -- this library links no collective library
int abn_rccl_allreduce_f64(void* ctx, void* device_doubles, int64_t n, void* stream)
{
    typedef int (*all_reduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
    abn_rccl_ctx* c = static_cast<abn_rccl_ctx*>(ctx);
    if (!c || !c->comm || !c->all_reduce || !device_doubles || n < 0) { set_error("rccl_allreduce_f64: null context / buffer"); return 1; }
    constexpr int NCCL_FLOAT64 = 8, NCCL_SUM = 0;          // (ncclDataType_t / ncclRedOp_t: nccl.h)
    const int rc = reinterpret_cast<all_reduce_t>(c->all_reduce)(device_doubles, device_doubles, (size_t)n, NCCL_FLOAT64, NCCL_SUM, c->comm,
                                                                  (hipStream_t)stream);
    if (rc != 0) { set_error("rccl_allreduce_f64: ncclAllReduce returned %d", rc); return 1; }
    ++c->calls;
    return 0;
}

// (one double per workgroup: a BatchNorm tower's launches cut the rows per forward_once call -- up to one workgroup more)
int64_t abn_tower_backward_loss_ws_bytes(int64_t rows) { return 8 + ((rows + PL_ROWS - 1) / PL_ROWS + 2) * (int64_t)sizeof(double); }

int abn_tower_backward_loss(const abn_tower_desc* t, const float* x1, const float* x2, const void* y, int y_dtype,
                            int loss_kind, float margin, int avg, int64_t rows, const float* ws, float* scratch,
                            int64_t scratch_floats, float* loss_out, void* loss_ws, const int32_t* n_valid, double* loss_accum,
                            void* stream)
{
    int rc = check_desc(t, rows, 2);
    if (rc != ABN_OK) return rc;
    ABN_REQUIRE(x1 && y && ws && scratch && loss_out && loss_ws, "tower_backward_loss: null pointer");
    ABN_REQUIRE(loss_kind == ABN_LOSS_COSCOS2 || loss_kind == ABN_LOSS_COSMARGIN, "tower_backward_loss: unknown loss kind %d", loss_kind);
    ABN_REQUIRE(y_dtype >= ABN_Y_I8 && y_dtype <= ABN_Y_F64, "tower_backward_loss: unknown label dtype %d", y_dtype);
    ABN_REQUIRE(loss_kind != ABN_LOSS_COSMARGIN || (margin >= 0.0f && margin <= 1.0f), "tower_backward_loss: margin outside [0,1]");
    for (int l = 0; l < t->n_layers; ++l) ABN_REQUIRE(t->dW[l] && t->db[l], "tower_backward_loss: layer %d has null gradient buffers", l);
    // a BatchNorm tower on its layer launches (per-replica statistics, all parts of the backward in this call): the loss rides in
    // the launch that sums the output layer's dy and dy xhat
    const bool bn = rows > 0 && t->batch_norm && bn_train_planes_path(t, rows, 2, x1, x2, ws) && !bn_sync_on(t) && t->wgrad_part == 0 &&
                    t->dims[t->n_layers] % 4 == 0;
    const int kind = rows == 0 || t->batch_norm ? PLANES_NONE : planes_kind(t, rows, 2, x1, x2, ws);
    if (kind == PLANES_NONE && !bn) {
        set_error("tower_backward_loss: only for towers the operand-plane kernels take (the split arithmetics, "
                  "widths <= 512 and multiples of 4, enough rows; BatchNorm without cross-replica statistics): use abn_pair_loss_dz + abn_tower_backward");
        return ABN_E_UNSUPPORTED;
    }
    if (t->source && (kind != PLANES_WIDE || bn || !t->source->labels)) {
        set_error("tower_backward_loss: a step source (abn_tower_desc.source) needs the layer-per-launch kernels and its labels");
        return ABN_E_UNSUPPORTED;
    }
    const Layout L = make_layout(t, rows, 2);
    const BwdLayout B = make_bwd_layout(t, rows);
    if (scratch_floats < B.total) { set_error("tower_backward_loss: scratch too small"); return ABN_E_WORKSPACE; }
    LossArgs la = {y, y_dtype, loss_kind, avg, margin, loss_out, loss_ws, n_valid, loss_accum};
    if (bn) return bn_planes_backward(t, nullptr, &la, rows, 2, L, B, ws, scratch, nullptr, (hipStream_t)stream);
    if (kind == PLANES_WIDE) return wide_backward(t, nullptr, &la, rows, 2, L, B, ws, scratch, nullptr, (hipStream_t)stream);
    return planes_backward(t, nullptr, &la, rows, L, B, ws, scratch, nullptr, (hipStream_t)stream);
}

int abn_tower_reduce_step(const abn_tower_desc* t, int64_t rows, const float* scratch, int64_t scratch_floats, int kind,
                          float* params, float* grads, float* state1, float* state2, int64_t n, float lr, float hp0,
                          float hp1, float eps, int64_t step, float grad_scale, void* stream)
{
    int rc = check_desc(t, rows, 1);
    if (rc != ABN_OK) return rc;
    ABN_REQUIRE(kind >= ABN_OPT_SGD && kind <= ABN_OPT_RMSPROP, "tower_reduce_step: unknown optimizer %d", kind);
    ABN_REQUIRE(scratch && params && grads && state1, "tower_reduce_step: null pointer");
    ABN_REQUIRE(state2 || (kind != ABN_OPT_ADADELTA && kind != ABN_OPT_ADAM), "tower_reduce_step: state2 required");
    ABN_REQUIRE(step >= 1 && n >= 1, "tower_reduce_step: bad n/step");
    const BwdLayout B = make_bwd_layout(t, rows);
    if (scratch_floats < B.total) { set_error("tower_reduce_step: scratch too small"); return ABN_E_WORKSPACE; }
    if (t->defer_reduce && wgrad_step_small_ok(t, rows, t->fwd_calls)) {
        // the backward left the transposed images and no slabs: every layer's weight gradient + the rule, one launch
        const Layout L = make_layout(t, rows, t->fwd_calls);
        const int64_t nrb = t->fwd_calls * bn_wgs_per_call(rows, t->fwd_calls);
        WgsP w = {};
        w.n_layers = t->n_layers;
        w.tp_steps = (int)(2 * nrb);
        int n_wg = 0;
        for (int l = 0; l < t->n_layers; ++l) {
            WgsLayer& W = w.L[l];
            W.N = (int)t->dims[l + 1]; W.K = (int)t->dims[l];
            W.nblk = pl_blocks(W.N); W.kblk = pl_blocks(W.K + 1);
            ABN_REQUIRE(t->dW[l] && t->db[l] && t->dW[l] >= grads && t->dW[l] + (int64_t)W.N * W.K <= grads + n && t->db[l] >= grads &&
                            t->db[l] + W.N <= grads + n,
                        "tower_reduce_step: layer %d's gradients are not inside the flat buffer", l);
            ABN_REQUIRE(B.amax_dz[l] >= 0 && L.amax[l] >= 0, "tower_reduce_step: the images' maxima are missing");
            W.dzp = reinterpret_cast<const char*>(scratch + B.dzp[l]);
            W.ap = reinterpret_cast<const char*>(t->fwd_ws + L.tp[l]);
            W.amax_dz = scratch + B.amax_dz[l];
            W.amax_a = t->fwd_ws + L.amax[l];
            W.tiles_k = (W.kblk + 1) / 2;
            W.first_wg = n_wg;
            W.w_off = t->dW[l] - grads; W.b_off = t->db[l] - grads;
            n_wg += ((W.nblk + 1) / 2) * W.tiles_k;
        }
        w.n_tiles = n_wg;
        w.per_xcd = (n_wg + 7) / 8;
        n_wg = 8 * w.per_xcd;
        w.o = make_optp(kind, lr, hp0, hp1, eps, step, grad_scale);
        w.params = params; w.grads = grads; w.s1 = state1; w.s2 = state2;
        w.fail_word = nullptr;
        w.step_ctr = t->source ? t->source->step_ctr : nullptr;
        static bool wgs_attr[16] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        dev = (dev >= 0 && dev < 16) ? dev : 0;
        if (!wgs_attr[dev]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_step_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WGS_LDS_BYTES) != hipSuccess) {
                set_error("tower_reduce_step: cannot reserve %zu bytes of LDS", WGS_LDS_BYTES);
                return ABN_E_LAUNCH;
            }
            wgs_attr[dev] = true;
        }
        hipLaunchKernelGGL(wgrad_step_small_kernel, dim3((unsigned)n_wg), dim3(WGS_NT), WGS_LDS_BYTES, (hipStream_t)stream, w);
        ABN_CHECK_LAUNCH("tower_reduce_step (weight gradients + step)");
        return ABN_OK;
    }
    ReduceTable rt = make_reduce_table(t, B);
    for (int l = 0; l < t->n_layers; ++l) {      // every gradient tensor must lie inside the flat buffers
        ABN_REQUIRE(t->dW[l] && t->db[l] && t->dW[l] >= grads && t->dW[l] + rt.nW[l] <= grads + n && t->db[l] >= grads &&
                        t->db[l] + rt.nb[l] <= grads + n,
                    "tower_reduce_step: layer %d's gradients are not inside the flat buffer", l);
        if (t->batch_norm) {                     // gamma / beta: their gradients are final already, the same launch steps them
            ABN_REQUIRE(t->dbn_w[l] && t->dbn_b[l] && t->dbn_w[l] >= grads && t->dbn_w[l] + rt.nb[l] <= grads + n &&
                            t->dbn_b[l] >= grads && t->dbn_b[l] + rt.nb[l] <= grads + n,
                        "tower_reduce_step: layer %d's BatchNorm gradients are not inside the flat buffer", l);
            rt.extra_off[rt.n_extra] = t->dbn_w[l] - grads; rt.extra_n[rt.n_extra++] = (int32_t)rt.nb[l];
            rt.extra_off[rt.n_extra] = t->dbn_b[l] - grads; rt.extra_n[rt.n_extra++] = (int32_t)rt.nb[l];
        }
    }
    hipLaunchKernelGGL(slab_reduce_step_kernel, dim3(grid_for((rt.total + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       scratch + B.slabs, rt, make_optp(kind, lr, hp0, hp1, eps, step, grad_scale), params, grads, state1,
                       state2, t->batch_norm && t->sync_ws ? reinterpret_cast<const unsigned*>(t->sync_ws) + 16 : nullptr,
                       t->source ? t->source->step_ctr : nullptr);
    ABN_CHECK_LAUNCH("tower_reduce_step");
    return ABN_OK;
}


// ---- single-layer entry points (the same kernels, one Linear at a time) -------
int abn_linear_forward(const float* x, const float* W, const float* b, int64_t rows, int64_t in_dim,
                       int64_t out_dim, int act, float* y, void* stream)
{
    ABN_REQUIRE(x && W && y, "linear_forward: null pointer");
    ABN_REQUIRE(rows >= 0 && rows < (1LL << 30) && in_dim >= 1 && out_dim >= 1 && in_dim < (1 << 24) && out_dim < (1 << 24),
                "linear_forward: bad shape");
    ABN_REQUIRE(act >= ABN_ACT_NONE && act <= ABN_ACT_TANH, "linear_forward: unsupported activation %d", act);
    GemmP p = {};
    p.A = x; p.lda = in_dim;
    p.B = W; p.ldb = in_dim;
    p.C = y; p.ldc = out_dim;
    p.M = (int)rows; p.N = (int)out_dim; p.K = (int)in_dim; p.k_chunk = (int)in_dim;
    p.bias = b; p.act = act; p.ones_col = -1;
    p.a_vec = aligned16(x) && (in_dim % 4 == 0);
    p.b_vec = aligned16(W) && (in_dim % 4 == 0);
    return launch_gemm<true, true, EPI_FWD>(p, 1, (hipStream_t)stream);
}

int abn_linear_dgrad(const float* dz, const float* W, int64_t rows, int64_t in_dim, int64_t out_dim,
                     const float* a_prev, int act_prev, float* dx, void* stream)
{
    ABN_REQUIRE(dz && W && dx, "linear_dgrad: null pointer");
    ABN_REQUIRE(rows >= 0 && rows < (1LL << 30) && in_dim >= 1 && out_dim >= 1 && in_dim < (1 << 24) && out_dim < (1 << 24),
                "linear_dgrad: bad shape");
    ABN_REQUIRE(act_prev >= ABN_ACT_NONE && act_prev <= ABN_ACT_TANH, "linear_dgrad: unsupported activation %d", act_prev);
    GemmP p = {};
    p.A = dz; p.lda = out_dim;
    p.B = W; p.ldb = in_dim;
    p.C = dx; p.ldc = in_dim;
    p.M = (int)rows; p.N = (int)in_dim; p.K = (int)out_dim; p.k_chunk = (int)out_dim;
    p.aux = a_prev; p.ldaux = in_dim; p.act = act_prev; p.ones_col = -1;
    p.a_vec = aligned16(dz) && (out_dim % 4 == 0);
    p.b_vec = aligned16(W) && (in_dim % 4 == 0);
    return launch_gemm<true, false, EPI_DGRAD>(p, 1, (hipStream_t)stream);
}

int64_t abn_linear_wgrad_scratch_floats(int64_t rows, int64_t in_dim, int64_t out_dim)
{
    if (rows < 0 || in_dim < 1 || out_dim < 1) return -1;
    return align_up(out_dim * in_dim + out_dim, 64) * split_count(rows, out_dim, in_dim);
}

int abn_linear_wgrad(const float* dz, const float* a_in, int64_t rows, int64_t in_dim, int64_t out_dim, float* dW,
                     float* db, float* scratch, int64_t scratch_floats, void* stream)
{
    ABN_REQUIRE(dz && a_in && dW && db && scratch, "linear_wgrad: null pointer");
    ABN_REQUIRE(rows >= 1 && rows < (1LL << 30) && in_dim >= 1 && out_dim >= 1 && in_dim < (1 << 24) && out_dim < (1 << 24),
                "linear_wgrad: bad shape");
    const int splits = split_count(rows, out_dim, in_dim);
    const int64_t stride = align_up(out_dim * in_dim + out_dim, 64);
    if (scratch_floats < stride * splits) { set_error("linear_wgrad: scratch too small"); return ABN_E_WORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    GemmP p = {};
    p.A = dz; p.lda = out_dim;
    p.B = a_in; p.ldb = in_dim;
    p.C = scratch; p.ldc = in_dim;
    p.C2 = scratch + out_dim * in_dim;
    p.slab_stride = stride;
    p.M = (int)out_dim; p.N = (int)in_dim + 1; p.K = (int)rows;
    p.k_chunk = (int)align_up((rows + splits - 1) / splits, BK);
    p.ones_col = (int)in_dim;
    p.a_vec = aligned16(dz) && (out_dim % 4 == 0);
    p.b_vec = aligned16(a_in) && (in_dim % 4 == 0);
    int rc = launch_gemm<false, false, EPI_WGRAD>(p, splits, st);
    if (rc != ABN_OK) return rc;
    ReduceTable rt = {};
    rt.n_layers = 1; rt.splits[0] = splits; rt.slab_stride = stride;
    rt.off[0] = 0; rt.nW[0] = out_dim * in_dim; rt.nb[0] = out_dim; rt.dW[0] = dW; rt.db[0] = db;
    rt.total = out_dim * in_dim + out_dim;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(grid_for((rt.total + 3) / 4)), dim3(256), 0, st, scratch, rt);
    ABN_CHECK_LAUNCH("linear_wgrad");
    return ABN_OK;
}

// dgrad + wgrad of one nn.Linear as the tower backward issues them: ONE grid
// (gemm_bwd_pair_kernel) when both take their vectorised instantiations, then the slab
// reduction.  a_in doubles as a_prev (the layer's input IS the previous activation).
int abn_linear_backward(const float* dz, const float* W, const float* a_in, int64_t rows, int64_t in_dim,
                        int64_t out_dim, int act_prev, float* dW, float* db, float* dx, float* scratch,
                        int64_t scratch_floats, void* stream)
{
    return abn_linear_backward_prec(dz, W, a_in, rows, in_dim, out_dim, act_prev, 0, dW, db, dx, scratch, scratch_floats, stream);
}

int abn_linear_backward_prec(const float* dz, const float* W, const float* a_in, int64_t rows, int64_t in_dim,
                             int64_t out_dim, int act_prev, int precision, float* dW, float* db, float* dx,
                             float* scratch, int64_t scratch_floats, void* stream)
{
    ABN_REQUIRE(precision >= 0 && precision <= 3, "linear_backward: precision=%d (0 = fp32, 1 = bf16, 2 = bf16 x 3, 3 = fp16 x 2: run as bf16 x 3 here)", precision);
    ABN_REQUIRE(dz && W && a_in && dx && scratch && ((dW == nullptr) == (db == nullptr)), "linear_backward: null pointer");
    ABN_REQUIRE(rows >= 1 && rows < (1LL << 30) && in_dim >= 1 && out_dim >= 1 && in_dim < (1 << 24) && out_dim < (1 << 24),
                "linear_backward: bad shape");
    ABN_REQUIRE(act_prev >= ABN_ACT_NONE && act_prev <= ABN_ACT_TANH, "linear_backward: unsupported activation %d", act_prev);
    const int splits = split_count(rows, out_dim, in_dim);
    const int64_t stride = align_up(out_dim * in_dim + out_dim, 64);
    if (scratch_floats < stride * splits) { set_error("linear_backward: scratch too small"); return ABN_E_WORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    GemmP pw = {};
    pw.A = dz; pw.lda = out_dim;
    pw.B = a_in; pw.ldb = in_dim;
    pw.C = scratch; pw.ldc = in_dim;
    pw.C2 = scratch + out_dim * in_dim;
    pw.slab_stride = stride;
    pw.M = (int)out_dim; pw.N = (int)in_dim + 1; pw.K = (int)rows;
    pw.k_chunk = (int)align_up((rows + splits - 1) / splits, BK);
    pw.ones_col = (int)in_dim;
    pw.a_vec = aligned16(dz) && (out_dim % 4 == 0);
    pw.b_vec = aligned16(a_in) && (in_dim % 4 == 0);
    GemmP pd = {};
    pd.A = dz; pd.lda = out_dim;
    pd.B = W; pd.ldb = in_dim;
    pd.C = dx; pd.ldc = in_dim;
    pd.M = (int)rows; pd.N = (int)in_dim; pd.K = (int)out_dim; pd.k_chunk = (int)out_dim;
    pd.aux = act_prev == ABN_ACT_NONE ? nullptr : a_in; pd.ldaux = in_dim; pd.act = act_prev; pd.ones_col = -1;
    pd.a_vec = aligned16(dz) && (out_dim % 4 == 0);
    pd.b_vec = aligned16(W) && (in_dim % 4 == 0);
    pw.bf16 = pd.bf16 = gemm_prec(precision);
    int rc = launch_bwd_pair(pw, splits, pd, st);
    if (rc != ABN_OK) return rc;
    if (!dW) return ABN_OK;                      // slabs left unreduced in scratch: the pair grid alone (kernel timing)
    ReduceTable rt = {};
    rt.n_layers = 1; rt.splits[0] = splits; rt.slab_stride = stride;
    rt.off[0] = 0; rt.nW[0] = out_dim * in_dim; rt.nb[0] = out_dim; rt.dW[0] = dW; rt.db[0] = db;
    rt.total = out_dim * in_dim + out_dim;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(grid_for((rt.total + 3) / 4)), dim3(256), 0, st, scratch, rt);
    ABN_CHECK_LAUNCH("linear_backward");
    return ABN_OK;
}

}  // extern "C"

#ifdef ABN_WGS_STAMPS
extern "C" int abn_debug_wgs_stamps(unsigned long long* out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(abn::g_wgs_stamps), sizeof(abn::g_wgs_stamps)) == hipSuccess ? 0 : -1;
}
#endif
