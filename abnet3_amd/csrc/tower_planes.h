// tower_planes.h -- the tower forward on bf16 MFMA operands that are split ONCE (bf16 x 3, or plain
// bf16), no BatchNorm, widths <= 512.  Replaces tower_fused.h's kernel for precision 1 and 2.
//
// What the stamps and the knock-out builds said about that kernel: its floor is
// the per-wave weight stream (LDS-DMA of 64-byte row pieces, 1.6 us latency under load: 80 us with
// every MFMA and fragment read removed), and in the bf16 x 3 arithmetic every wave re-split the same
// fp32 fragments on the VALU (activations 8 times per workgroup, weights once per workgroup = 256
// times per step).  Here nothing on the k-loop touches the VALU:
//
//   * pack_planes_kernel turns each weight matrix into MFMA operand fragments once per call:
//       image[block of 32 output features][step of 16 k][plane hi|mid|lo][lane][8 bf16]
//     -- 1 KB per (block, step, plane), consecutive steps contiguous: a wave streams its block's
//     weights with one 16-byte global load per lane, plane and step, STRAIGHT INTO REGISTERS
//     (full cache lines, no LDS round trip, the ring of PL_DEPTH steps lives in VGPRs);
//   * the products are transposed, Y^T = W X^T: the weights are the MFMA's A operand, the 32 rows of
//     the workgroup its B operand.  The accumulator then has the batch row on the lane and 16
//     output features in its registers -- which IS the B-operand layout of the next layer (guide:
//     "an accumulator tile as the next MFMA's operand", k order 16s + 8(j>>2) + 4h + (j&3) inside
//     a step; the packed weights carry the same permutation).  The epilogue splits each value once
//     and writes whole fragments, 16 bytes per lane and plane, into the LDS image
//       img[step][plane][lane][8 bf16]
//     that all eight waves read back lane-linearly (conflict free) as their B operand.
//
// Per step and 32-feature block a wave issues 3 global loads, (shared by its blocks) 3 LDS reads
// and 6 MFMAs -- no conversion, no address arithmetic beyond an add.
#pragma once
#include <type_traits>

#include "gemm_f32.h"

namespace abn {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int PL_ROWS = 32;
constexpr int PL_WAVES = 8;
constexpr int PL_NT = 64 * PL_WAVES;
constexpr int PL_MAXW = 512;
constexpr int PL_MAXSTEPS = PL_MAXW / 16;
#ifdef PL_DEPTH_OVERRIDE
constexpr int PL_DEPTH = PL_DEPTH_OVERRIDE;
#else
constexpr int PL_DEPTH = 4;                       // weight steps in flight per wave (registers)
#endif
constexpr int PL_PART_BYTES = 4 * 16 * 64 * 4;    // K-split hand-over: 4 waves x one accumulator block

__host__ __device__ inline int pl_steps(int64_t contraction) { return ((int)((contraction + 15) / 16) + PL_DEPTH - 1) / PL_DEPTH * PL_DEPTH; }
static inline int pl_blocks(int64_t features) { return (int)((features + 31) / 32); }
static inline int64_t pl_image_bytes(int64_t features, int64_t contraction, int np)
{
    return (int64_t)pl_blocks(features) * pl_steps(contraction) * np * 1024;
}
static inline size_t pl_lds_bytes(int np) { return (size_t)PL_MAXSTEPS * np * 1024 + PL_PART_BYTES; }

// ---------------------------------------------------------------------------------------------
// weights -> operand fragments
// ---------------------------------------------------------------------------------------------
struct PackJob {
    const float* W;        // [N][K] row-major (nn.Linear.weight)
    int N, K;
    int transposed;        // 0: operand rows are output features, sum over k (forward)
                           // 1: operand rows are input features, sum over n (backward, W^T)
    int nblk, nsteps;
    int tile0;             // first tile (block, step) of this job in the launch
    int64_t dst;           // byte offset of the image
};
struct PackTable {
    int n_jobs, n_tiles;
    char* base;
    PackJob job[2 * ABN_MAX_LAYERS];
};

template <int NP>
__device__ __forceinline__ void write_frag(char* dst, const f32x4& v0, const f32x4& v1)
{
    if constexpr (NP == 3) {
        const bf16x8x3 s = split_bf16x3(v0, v1);
        *reinterpret_cast<bf16x8*>(dst) = s.hi;
        *reinterpret_cast<bf16x8*>(dst + 1024) = s.mid;
        *reinterpret_cast<bf16x8*>(dst + 2048) = s.lo;
    } else {
        *reinterpret_cast<bf16x8*>(dst) = pack_bf16(v0, v1);
    }
}

template <int NP>
__global__ __launch_bounds__(256) void pack_planes_kernel(PackTable t)
{
    const int tile = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (tile >= t.n_tiles) return;
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    int jn = 0;
    while (jn + 1 < t.n_jobs && tile >= t.job[jn + 1].tile0) ++jn;
    const PackJob& J = t.job[jn];
    const int local = tile - J.tile0;
    const int nb = local / J.nsteps, s = local % J.nsteps;
    const int a = 32 * nb + r;                       // operand row
    const int rows_a = J.transposed ? J.K : J.N;     // operand rows in the matrix
    const int len_c = J.transposed ? J.N : J.K;      // length of the sum
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
    if (a < rows_a) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c0 = 16 * s + 4 * h + e, c1 = c0 + 8;
            if (c0 < len_c) v0[e] = J.transposed ? J.W[(int64_t)c0 * J.K + a] : J.W[(int64_t)a * J.K + c0];
            if (c1 < len_c) v1[e] = J.transposed ? J.W[(int64_t)c1 * J.K + a] : J.W[(int64_t)a * J.K + c1];
        }
    }
    write_frag<NP>(t.base + J.dst + ((int64_t)local * NP) * 1024 + lane * 16, v0, v1);
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
struct PlanesFwdP {
    int n_layers;
    int rows;                      // total rows (both towers)
    int rows_call;                 // rows per forward_once call (x1 | x2 split)
    int dims[ABN_MAX_LAYERS + 1];
    int act[ABN_MAX_LAYERS];
    const float* x1;
    const float* x2;               // may be null: all rows in x1
    float* x_copy;                 // [rows, dims[0]] concatenated copy for the backward (may be null)
    const char* wp[ABN_MAX_LAYERS];   // packed forward image of layer l
    const float* b[ABN_MAX_LAYERS];
    const float* mask[ABN_MAX_LAYERS];
    float* out[ABN_MAX_LAYERS];    // [rows, dims[l+1]] post-activation outputs
#ifdef ABN_STAMPS
    unsigned long long* stamps;
#endif
};

#ifdef ABN_STAMPS
#define PSTAMPF(slot) do { if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 128 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#define WSTAMP(k) do { if (p.stamps && lane == 0) p.stamps[(size_t)blockIdx.x * 128 + 32 + wave * 12 + 4 * l + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PSTAMPF(slot) do {} while (0)
#define WSTAMP(k) do {} while (0)
#endif

// One layer for one workgroup.  img holds the input fragments of all pl_steps(K) steps (zero in
// the padding); on return it holds this layer's output the same way, for pl_steps(N) steps.
// KS = 2 (at most 4 blocks): waves 4..7 sum the second half of the steps for the blocks of
// waves 0..3 and hand their accumulators over through LDS.
template <int NP, int BPW, int KS>
__device__ __forceinline__ void planes_layer(const PlanesFwdP& p, int l, char* __restrict__ img,
                                             float* __restrict__ part, int wave, int lane, int row0)
{
    const int K = p.dims[l], N = p.dims[l + 1];
    const int nsteps = pl_steps(K), nblk = (N + 31) / 32;
    const int r = lane & 31, h = lane >> 5;
    const int khalf = KS == 2 ? wave >> 2 : 0;
    const int blk0 = KS == 2 ? (wave & 3) : wave * BPW;
    const int s_first = KS == 2 ? khalf * (nsteps / 2) : 0;
    const int my_steps = KS == 2 ? nsteps / 2 : nsteps;          // a multiple of PL_DEPTH (host: KS = 2 only if nsteps % 8 == 0)
    const bool active = blk0 < nblk;
    PSTAMPF(2 + 5 * l);
    WSTAMP(0);

    f32x16 acc[BPW];
#pragma unroll
    for (int j = 0; j < BPW; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][q] = 0.0f;

    f32x4 bv[BPW][4];
    {
        const float* __restrict__ bias = p.b[l];
#pragma unroll
        for (int j = 0; j < BPW; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = 32 * (blk0 + j) + 4 * h + 8 * g;
                bv[j][g] = bias && active ? *reinterpret_cast<const f32x4*>(bias + (n < N ? n : N - 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
    }
    if (active) {
        // The weight ring is filled with raw buffer loads: the compiler counts their vmcnt itself, and --
        // unlike plain loads, which InstCombine sinks through the loop's phi right in front of their
        // MFMAs -- they stay where the pipeline puts them, PL_DEPTH steps ahead.
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(p.wp[l]), 0, nblk * nsteps * (NP * 1024), 0x00020000);
        int wv[BPW];
#pragma unroll
        for (int j = 0; j < BPW; ++j) {
            const int blk = blk0 + j < nblk ? blk0 + j : nblk - 1;        // an odd block count leaves the last wave half idle
            wv[j] = (blk * nsteps + s_first) * (NP * 1024) + lane * 16;
        }
        const char* ab = img + (int64_t)s_first * (NP * 1024) + lane * 16;
        v4i wq[PL_DEPTH][BPW][NP];
#pragma unroll
        for (int i = 0; i < PL_DEPTH; ++i)
#pragma unroll
            for (int j = 0; j < BPW; ++j)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                {   // (fenced one by one: the in-order vmcnt the compiler derives for the loop is the worst of
                    // the loop's own order and this one)
                    wq[i][j][pl] = __builtin_amdgcn_raw_buffer_load_b128(rs, wv[j], (i * NP + pl) * 1024, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
        // activation fragments: read one step ahead, into alternating register sets
        bf16x8 af[2][NP];
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) af[0][pl] = *reinterpret_cast<const bf16x8*>(ab + pl * 1024);
        for (int s0 = 0; s0 < my_steps; s0 += PL_DEPTH) {
            if (s0 == PL_DEPTH) WSTAMP(1);
#pragma unroll
            for (int i = 0; i < PL_DEPTH; ++i) {
                const int s = s0 + i;
                // (the three regions are fenced: left alone, the machine scheduler issues a refill before
                // the slot's last MFMA -- a second register set and copies that wait for the loads at the
                // loop's end -- or gathers all refills there)
                const int s1 = s + 1 < my_steps ? s + 1 : s;
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
#ifndef PEXP_NOA
                    af[(i + 1) & 1][pl] = *reinterpret_cast<const bf16x8*>(ab + (s1 * NP + pl) * 1024);
#else
                    af[(i + 1) & 1][pl][0] = (__bf16)(float)s1;
#endif
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8* a = af[i & 1];
#ifdef PEXP_NOMFMA
#pragma unroll
                for (int j = 0; j < BPW; ++j)
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) acc[j][pl] += (float)wq[i][j][pl][0] + (float)a[pl][1];
#else
                if constexpr (NP == 3) {
                    // smallest terms first (gemm_f32.h); the blocks alternate so that consecutive MFMAs are independent
                    constexpr int WP[6] = {2, 0, 1, 1, 0, 0}, AP[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                    for (int t = 0; t < 6; ++t)
#pragma unroll
                        for (int j = 0; j < BPW; ++j)
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wq[i][j][WP[t]]), a[AP[t]], acc[j], 0, 0, 0);
                } else {
#pragma unroll
                    for (int j = 0; j < BPW; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wq[i][j][0]), a[0], acc[j], 0, 0, 0);
                }
#endif
                // (pure MFMA nodes float across a sched_barrier at instruction selection: the empty asm
                // ties the accumulators, and with them every MFMA of the step, in front of the refills)
#pragma unroll
                for (int j = 0; j < BPW; ++j) asm volatile("" : "+v"(acc[j]) :: "memory");
                __builtin_amdgcn_sched_barrier(0);
                // refill the slot for step s + PL_DEPTH (clamped: the last PL_DEPTH loads are repeats nobody reads)
                const int sn = s + PL_DEPTH < my_steps ? s + PL_DEPTH : my_steps - 1;
#pragma unroll
                for (int j = 0; j < BPW; ++j)
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
#if defined(PEXP_SAMEW)
                        wq[i][j][pl] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, ((sn & 3) * NP + pl) * 1024, 0);
#elif !defined(PEXP_NOW)
                        wq[i][j][pl] = __builtin_amdgcn_raw_buffer_load_b128(rs, wv[j], (sn * NP + pl) * 1024, 0);
#else
                        wq[i][j][pl][0] += sn;
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    PSTAMPF(3 + 5 * l);
    WSTAMP(2);
    // epilogue.  Register q of block j is output feature 32 (blk0 + j) + (q & 3) + 8 (q >> 2) + 4 h of
    // batch row r: registers 4g .. 4g+3 are four consecutive features (one 16-byte piece of the
    // row-major output), registers 8t .. 8t+7 the lane's operand of step 2 blk + t of the next layer.
    // All loads of the epilogue are issued as one batch from clamped addresses (a load under a
    // per-group `if` waits for its own round trip: eight of them in a row were 8.5 k cycles per layer),
    // the bias before the k-loop, the dropout mask here; the activation is a template argument
    // (its runtime switch inside the unrolled loops was most of the kernel's code, run once, from a
    // cold instruction cache).
    const float* __restrict__ mask = p.mask[l];
    const int gr = row0 + r;
    const bool row_ok = gr < p.rows;
    f32x4 mv[BPW][4];
    if (mask && active && khalf == 0) {
        const float* mrow = mask + (int64_t)(row_ok ? gr : p.rows - 1) * N;
#pragma unroll
        for (int j = 0; j < BPW; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = 32 * (blk0 + j) + 4 * h + 8 * g;
                mv[j][g] = *reinterpret_cast<const f32x4*>(mrow + (n < N ? n : N - 4));
            }
    }
    auto finish_as = [&](auto act_tag) {
        constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
        for (int j = 0; j < BPW; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const bool live = 32 * (blk0 + j) + 4 * h + 8 * g < N;      // N % 4 == 0: four features in or out together
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[j][4 * g + e] + bv[j][g][e];
                    if (mask) v *= mv[j][g][e];
                    acc[j][4 * g + e] = live ? act_apply(v, ACT) : 0.0f;
                }
            }
    };
    auto finish = [&]() {
        switch (p.act[l]) {
            case ACT_SIGMOID: finish_as(std::integral_constant<int, ACT_SIGMOID>{}); break;
            case ACT_RELU: finish_as(std::integral_constant<int, ACT_RELU>{}); break;
            case ACT_TANH: finish_as(std::integral_constant<int, ACT_TANH>{}); break;
            default: finish_as(std::integral_constant<int, ACT_NONE>{}); break;
        }
    };
    if (KS == 1) {
        if (active) finish();
    } else if (active && khalf == 1) {
#pragma unroll
        for (int q = 0; q < 16; ++q) part[((wave & 3) * 16 + q) * 64 + lane] = acc[0][q];
    }
    PSTAMPF(4 + 5 * l);
    WSTAMP(3);
    __syncthreads();                               // every wave is done reading img
    PSTAMPF(5 + 5 * l);
    if (KS == 2 && active && khalf == 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[0][q] += part[(wave * 16 + q) * 64 + lane];
        finish();
    }
    float* __restrict__ out = p.out[l];
    if (active && khalf == 0) {
#pragma unroll
        for (int j = 0; j < BPW; ++j) {
            const int blk = blk0 + j;
            if (blk < nblk) {
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    const f32x4 v0 = {acc[j][8 * t2], acc[j][8 * t2 + 1], acc[j][8 * t2 + 2], acc[j][8 * t2 + 3]};
                    const f32x4 v1 = {acc[j][8 * t2 + 4], acc[j][8 * t2 + 5], acc[j][8 * t2 + 6], acc[j][8 * t2 + 7]};
                    write_frag<NP>(img + (int64_t)(2 * blk + t2) * (NP * 1024) + lane * 16, v0, v1);
                    const int n = 32 * blk + 16 * t2 + 4 * h;
#ifndef PEXP_NOOUT
                    if (row_ok && n < N) *reinterpret_cast<f32x4*>(out + (int64_t)gr * N + n) = v0;
                    if (row_ok && n + 8 < N) *reinterpret_cast<f32x4*>(out + (int64_t)gr * N + n + 8) = v1;
#else
                    if (row_ok && n < -N) *reinterpret_cast<f32x4*>(out + (int64_t)gr * N + n) = v0 + v1;
#endif
                }
            }
        }
    }
    PSTAMPF(6 + 5 * l);
    // steps of the next layer's padding that no block of this layer covers
    if (l + 1 < p.n_layers) {
        const int next_steps = pl_steps(N);
        const bf16x8 z = {};
        for (int s = 2 * nblk + wave; s < next_steps; s += PL_WAVES)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<bf16x8*>(img + ((int64_t)s * NP + pl) * 1024 + lane * 16) = z;
    }
    __syncthreads();
}

template <int NP>
__global__ __launch_bounds__(PL_NT) void tower_fwd_planes_kernel(PlanesFwdP p)
{
    extern __shared__ __attribute__((aligned(16))) char pl_smem[];
    char* const img = pl_smem;
    float* const part = reinterpret_cast<float*>(pl_smem + PL_MAXSTEPS * NP * 1024);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * PL_ROWS;
    const int D0 = p.dims[0];
    PSTAMPF(0);

    // input rows -> operand fragments (+ the concatenated copy for the backward): lane (r, h) of
    // step s holds x[row r][16 s + 4 h + 0..3] and x[row r][16 s + 8 + 4 h + 0..3]
    const int steps0 = pl_steps(D0);
    for (int s = wave; s < steps0; s += PL_WAVES) {
        const int gr = row0 + r;
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
        if (gr < p.rows) {
            const float* src = (p.x2 && gr >= p.rows_call) ? p.x2 + (int64_t)(gr - p.rows_call) * D0 : p.x1 + (int64_t)gr * D0;
            const int c0 = 16 * s + 4 * h, c1 = c0 + 8;
            if (c0 < D0) {
                v0 = *reinterpret_cast<const f32x4*>(src + c0);
                if (p.x_copy) *reinterpret_cast<f32x4*>(p.x_copy + (int64_t)gr * D0 + c0) = v0;
            }
            if (c1 < D0) {
                v1 = *reinterpret_cast<const f32x4*>(src + c1);
                if (p.x_copy) *reinterpret_cast<f32x4*>(p.x_copy + (int64_t)gr * D0 + c1) = v1;
            }
        }
        write_frag<NP>(img + (int64_t)s * (NP * 1024) + lane * 16, v0, v1);
    }
    PSTAMPF(1);
    __syncthreads();

    for (int l = 0; l < p.n_layers; ++l) {
        const int nblk = (p.dims[l + 1] + 31) / 32;
        if (nblk > PL_WAVES) planes_layer<NP, 2, 1>(p, l, img, part, wave, lane, row0);
        else if (nblk > PL_WAVES / 2 || pl_steps(p.dims[l]) % (2 * PL_DEPTH) != 0) planes_layer<NP, 1, 1>(p, l, img, part, wave, lane, row0);
        else planes_layer<NP, 1, 2>(p, l, img, part, wave, lane, row0);
    }
    PSTAMPF(2 + 5 * p.n_layers);
}

}  // namespace abn
