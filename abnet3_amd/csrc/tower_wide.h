// tower_wide.h -- the tower forward and data-gradient chain for SMALL batches (bf16 x 3 / bf16 operand planes).
//
// The reference's canonical loader yields 8 word pairs = a few hundred 280-d frame pairs per step
// (abnet3/dataloader.py:248-255).  The single-launch chains of tower_planes.h give such a batch ~20
// workgroups (one per 32 rows, every layer's whole weight image streamed by each): 64 + 51 us on 20 of the
// 256 CUs, whatever the batch.  Here a layer is its own launch and its OUTPUT BLOCKS are dealt over G
// workgroups per 32 rows (grid = row blocks x G, G <= 8): a workgroup stages its rows' input once
// (fp32 row-major -> operand fragments in LDS, the split done once), its eight waves share one or two
// 32-feature output blocks -- several waves per block, each summing a slice of the steps (planes_kloop, the
// same register ring) -- the slices are added through LDS in a fixed order, and the epilogue leaves the
// layer's output ROW-MAJOR fp32 for the next launch plus the transposed operand image the weight-gradient
// launch reads (emit_planes: the same images as the chains', so wgrad_planes_kernel is shared).
// A kernel boundary costs ~1.5 us here, against 3-5 us for an in-launch hand-off of a 96 KB operand image
// between workgroups (MI355X_MICROARCH.md, price list: "cut GEMM -> GEMM seams at these sizes").
//
// Rows are VIRTUAL: every forward_once call is padded to whole 32-row workgroups (call c starts at virtual
// row c * wpc * 32), so a batch of n pairs and the same batch padded with zero rows to ceil32(n) pairs (the
// captured steps of the trainer's planned passes) run the same arithmetic in the same order: bit-identical
// losses and gradients.  Rows past a call's end are zero inputs in the forward and zero rows of every dZ.
#pragma once
#include "tower_planes.h"

namespace abn {

constexpr int WD_MAXG = 8;
constexpr int WD_PART_BYTES = 7 * 16 * 64 * 4;          // up to 7 hand-over tiles (one block shared by 8 waves)
constexpr int WD_COEF_BYTES = 1024;                     // the pair loss's per-row coefficients (top dgrad launch)
static inline size_t wd_lds_bytes(int np) { return (size_t)PL_MAXSTEPS * np * 1024 + WD_PART_BYTES + WD_COEF_BYTES + PL_SC_BYTES; }
// fp16 x 2: the row maxima / inverse-scale tables (row_scales, tower_planes.h) behind the hand-over tiles
template <int NP> __device__ __forceinline__ float* wd_scales(char* smem) { return reinterpret_cast<float*>(smem + PL_MAXSTEPS * NP * 1024 + WD_PART_BYTES + WD_COEF_BYTES); }

// Which block a wave sums and which steps of it.  The workgroup owns blocks g, g + G, ... of the layer's
// nblk (nbw of them); its 8 waves are dealt `per` to a block, each taking a contiguous run of the
// nsteps / PL_DEPTH step groups.
struct WideShare {
    int j, blk, kpart, ks, per, s_first, my_steps;
    bool active;
    __device__ __forceinline__ WideShare(int wave, int g, int G, int nblk, int nsteps)
    {
        const int nbw = g < nblk ? (nblk - g + G - 1) / G : 0;
        const int nbp2 = nbw <= 1 ? 1 : (nbw <= 2 ? 2 : (nbw <= 4 ? 4 : 8));
        per = PL_WAVES / nbp2;
        j = wave / per;
        kpart = wave - j * per;
        const int groups = nsteps / PL_DEPTH;
        ks = groups < per ? groups : per;
        blk = g + G * j;
        active = j < nbw && kpart < ks;
        const int g0 = kpart * groups / ks, g1 = (kpart + 1) * groups / ks;
        s_first = g0 * PL_DEPTH;
        my_steps = (g1 - g0) * PL_DEPTH;
    }
    // hand-over tile of (block j, slice kpart >= 1)
    __device__ __forceinline__ int tile(int kp) const { return j * (per - 1) + kp - 1; }
};

// 32 rows of a row-major fp32 matrix (row pointer per lane, null = a zero row) -> operand fragments in img for
// pl_steps(K) steps.  emit(kb, f): called by the wave that built block kb's two fragments.  A wave's loads (up to
// two 32-feature blocks: K <= 512) are all issued before the first conversion: one round trip, not one per block.
// fp16 x 2: the workgroup first agrees on its rows' scales (one more barrier); returns the lane's row maximum, ainv its
// row's inverse scale.
template <int NP, class Emit>
__device__ __forceinline__ float wide_stage_rows(const float* __restrict__ src, int K, char* __restrict__ img, float* __restrict__ sc,
                                                 int wave, int lane, float& ainv, Emit&& emit)
{
    const int h = lane >> 5;
    const int blocks = pl_steps(K) / 2;
    f32x4 v[2][2][2];                                   // [block of this wave][step of the block][half]
    float m = 0.0f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int kb = wave + PL_WAVES * u;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const int c0 = 16 * (2 * kb + t2) + 4 * h, c1 = c0 + 8;
            v[u][t2][0] = v[u][t2][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (src && kb < blocks && c0 < K) v[u][t2][0] = *reinterpret_cast<const f32x4*>(src + c0);
            if (src && kb < blocks && c1 < K) v[u][t2][1] = *reinterpret_cast<const f32x4*>(src + c1);
            if constexpr (NP == 2) m = fmaxf(m, absmax8(v[u][t2][0], v[u][t2][1]));
        }
    }
    float osc = 1.0f;
    if constexpr (NP == 2) {
        sc[wave * 64 + lane] = m;
        __syncthreads();
        m = row_scales(sc, wave, lane, osc, ainv);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int kb = wave + PL_WAVES * u;
        if (kb < blocks) {
            Frag<NP> f[2];
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                f[t2] = make_frag<NP>(v[u][t2][0], v[u][t2][1], osc);
                store_frag<NP>(img + (int64_t)(2 * kb + t2) * (NP * 1024) + lane * 16, f[t2]);
            }
            emit(kb, f);
        }
    }
    return m;
}

// fp16 x 2, a finishing wave's 32 x 32 block on its way out transposed: the scale of each row from the block's own
// values (both lane halves of a row agree), the inverse in this wave's table for emit_planes; returns the block's maximum.
__device__ __forceinline__ float wide_block_scale(const f32x16& acc, float* __restrict__ sc, int wave, int lane, float& osc)
{
    float m = 0.0f;
#pragma unroll
    for (int q = 0; q < 16; ++q) m = fmaxf(m, fabsf(acc[q]));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float oinv;
    scale_of(m, osc, oinv);
    float* const tab = sc + PL_WAVES * 64 + wave * 32;
    tab[lane & 31] = oinv;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    return wave_max(m);
}

// planes_kloop (tower_planes.h) in two halves: the first PL_DEPTH steps' weight loads are issued BEFORE the
// workgroup stages its input rows (the weights do not depend on them), so the two round trips overlap
// (measured at 485 pairs of 280-d frames: forward launches 11.5 -> 9.8 us, data-gradient launches 11.4 -> 10.4;
// a ring of 8 steps, or the rows' loads in front of the ring's, were both slower: the launch is bound by what
// one CU can pull from its L2 -- 192 KB of weights + 64 KB of rows per workgroup -- not by a latency).
template <int NP>
struct WideRing {
    __amdgpu_buffer_rsrc_t rs;
    int wv;
    v4i wq[PL_DEPTH][NP];
    __device__ __forceinline__ void issue(const char* __restrict__ image, int nblk, int nsteps, int blk, int s_first, int lane)
    {
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(image), 0, nblk * nsteps * (NP * 1024), 0x00020000);
        wv = ((blk < nblk ? blk : nblk - 1) * nsteps + s_first) * (NP * 1024) + lane * 16;
#pragma unroll
        for (int i = 0; i < PL_DEPTH; ++i)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                wq[i][pl] = __builtin_amdgcn_raw_buffer_load_b128(rs, wv, (i * NP + pl) * 1024, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    __device__ __forceinline__ void run(f32x16& acc, const char* __restrict__ img, int s_first, int my_steps, int lane)
    {
        const char* ab = img + (int64_t)s_first * (NP * 1024) + lane * 16;
        bf16x8 af[2][NP];
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) af[0][pl] = *reinterpret_cast<const bf16x8*>(ab + pl * 1024);
        for (int s0 = 0; s0 < my_steps; s0 += PL_DEPTH) {
#pragma unroll
            for (int i = 0; i < PL_DEPTH; ++i) {
                const int s = s0 + i;
                const int s1 = s + 1 < my_steps ? s + 1 : s;
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) af[(i + 1) & 1][pl] = *reinterpret_cast<const bf16x8*>(ab + (s1 * NP + pl) * 1024);
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8* a = af[i & 1];
#pragma unroll
                for (int t = 0; t < Products<NP>::N; ++t)               // smallest terms first, as planes_kloop
                    acc = pl_mfma<NP>(__builtin_bit_cast(bf16x8, wq[i][Products<NP>::A[t]]), a[Products<NP>::B[t]], acc);
                asm volatile("" : "+v"(acc) :: "memory");
                __builtin_amdgcn_sched_barrier(0);
                const int sn = s + PL_DEPTH < my_steps ? s + PL_DEPTH : my_steps - 1;
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) wq[i][pl] = __builtin_amdgcn_raw_buffer_load_b128(rs, wv, (sn * NP + pl) * 1024, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
};

// The waves of one block add their slices: slices kpart >= 1 park their accumulators, slice 0 adds them in order.
__device__ __forceinline__ void wide_park(const WideShare& ws, const f32x16& acc, float* __restrict__ part, int lane)
{
    if (ws.active && ws.kpart > 0) {
        float* t = part + (int64_t)ws.tile(ws.kpart) * (16 * 64);
#pragma unroll
        for (int q = 0; q < 16; ++q) t[q * 64 + lane] = acc[q];
    }
}
__device__ __forceinline__ void wide_collect(const WideShare& ws, f32x16& acc, const float* __restrict__ part, int lane)
{
    for (int kp = 1; kp < ws.ks; ++kp) {
        const float* t = part + (int64_t)ws.tile(kp) * (16 * 64);
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] += t[q * 64 + lane];
    }
}

// ---------------------------------------------------------------------------------------------
// forward: one layer
// ---------------------------------------------------------------------------------------------
struct WideFwdP {
    int l, last;                   // layer index; 1 = the tower's last layer
    int K, N, act;
    int rows_call, n_calls, wpc;   // rows per forward_once call, calls, workgroup rows per call = ceil(rows_call / 32)
    int G;
    const float* x1;               // l == 0: the inputs (x2 null: all rows in x1)
    const float* x2;
    const float* a_prev;           // l >= 1: [virtual rows, K] the previous layer's output
    const char* wp;                // packed W_l
    const float* b;
    float* a_out;                  // [virtual rows, N] row-major (null for the last layer)
    float* out;                    // last layer: [rows, N] at the caller's row indices
    char* tp_in;                   // l == 0: transposed image of [x | 1] (null: nothing kept for a backward)
    char* tp_out;                  // l < last: transposed image of [a_l | 1] (null: inference)
    float* amax_in;                // fp16 x 2: the images' maxima (PL_AMAX floats per 32-row block, tower_planes.h)
    float* amax_out;
    int64_t tp_steps;
    const unsigned long long* drop_seed;
    float drop_p;
    // l == 0, abn_tower_desc.source: the rows come from the pass's plan (abn_step_source), not from x1 / x2
    const float* g_table;
    int64_t g_rows;
    const int64_t* g_idx1;
    const int64_t* g_idx2;
    const int64_t* g_steps;
    const int32_t* g_ctr;
#ifdef ABN_STAMPS
    unsigned long long* stamps;
#endif
};

#ifdef ABN_STAMPS
#define WSTAMP(slot) do { if (p.stamps && threadIdx.x == 0) p.stamps[((size_t)p.l * 1024 + blockIdx.x) * 16 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WSTAMP(slot) do {} while (0)
#endif

template <int NP>
__global__ __launch_bounds__(PL_NT) void wide_fwd_layer_kernel(WideFwdP p)
{
    extern __shared__ __attribute__((aligned(16))) char pl_smem[];
    char* const img = pl_smem;
    float* const part = reinterpret_cast<float*>(pl_smem + PL_MAXSTEPS * NP * 1024);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int rb = blockIdx.x / p.G, g = blockIdx.x - rb * p.G;
    const int call = rb / p.wpc, lb = rb - call * p.wpc;
    const int vrow = rb * PL_ROWS + r;                                 // virtual row of this lane
    const int rows_left = p.rows_call - lb * PL_ROWS;                  // real rows in this workgroup (>= 1)
    const bool row_ok = r < rows_left;
    const int64_t arow = (int64_t)call * p.rows_call + lb * PL_ROWS + r;   // the caller's row index
    const int K = p.K, N = p.N;
    bf16x8 idf[2];
    make_identity<NP>(idf, lane);
    float* const sc = wd_scales<NP>(pl_smem);
    const float* const inv_tab = sc + PL_WAVES * 64 + wave * 32;

    WSTAMP(0);
    const int nblk = pl_blocks(N), nsteps = pl_steps(K);
    const WideShare ws(wave, g, p.G, nblk, nsteps);
    WideRing<NP> ring;
    if (ws.active) ring.issue(p.wp, nblk, nsteps, ws.blk, ws.s_first, lane);      // the weights do not wait for the rows
    f32x4 b4[4] = {};
    if (ws.active && ws.kpart == 0 && p.b) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int n = 32 * ws.blk + 4 * h + 8 * gq;
            b4[gq] = *reinterpret_cast<const f32x4*>(p.b + (n < N ? n : N - 4));
        }
    }

    const float* src;
    if (p.l == 0 && p.g_table) {
        // the step's pairs [first, first + n) of the plan: pair lb * 32 + r of this call's tower, a zero row behind the last
        const int64_t st = *p.g_ctr;
        const int64_t first = p.g_steps[2 * st];
        const int n = (int)p.g_steps[2 * st + 1];
        const int pair = lb * PL_ROWS + r;
        src = nullptr;
        if (row_ok && pair < n) {
            const int64_t row = (call >= 1 ? p.g_idx2 : p.g_idx1)[first + pair];
            if ((uint64_t)row < (uint64_t)p.g_rows) src = p.g_table + row * K;      // (a row outside the table reads as zeros: abn_gather_pairs)
        }
    }
    else if (p.l == 0) src = !row_ok ? nullptr : (p.x2 && call >= 1 ? p.x2 + (arow - p.rows_call) * K : p.x1 + arow * K);
    else src = p.a_prev + (int64_t)vrow * K;
    char* const tp_in = p.l == 0 ? p.tp_in : nullptr;
    float ainv = 1.0f;
    const float in_max = wide_stage_rows<NP>(src, K, img, sc, wave, lane, ainv, [&](int kb, const Frag<NP>* f) {
        // the input's transposed image (weight gradient of layer 0): the row block's G workgroups share the blocks
        if (tp_in && kb < pl_blocks(K + 1) && kb % p.G == g)
            emit_planes<NP>(tp_in + ((int64_t)kb * p.tp_steps + 2 * rb) * tile_bytes<NP>(), f, idf, lane, kb == K / 32 ? K % 32 : -1, rows_left, inv_tab);
    });
    if (tp_in && pl_blocks(K + 1) > pl_steps(K) / 2 && wave == PL_WAVES - 1 && (K / 32) % p.G == g) {   // K % 32 == 0, no padding block for the ones
        Frag<NP> z[2] = {};
        emit_planes<NP>(tp_in + ((int64_t)(K / 32) * p.tp_steps + 2 * rb) * tile_bytes<NP>(), z, idf, lane, 0, rows_left, inv_tab);
    }
    if (NP == 2 && tp_in && g == 0 && wave == 0) store_amax_rows(p.amax_in + (int64_t)rb * PL_AMAX, in_max, 1.0f, lane);
    if (NP == 2 && p.tp_out && g == 0 && wave == PL_WAVES - 1) {           // the blocks this layer's image does not have
        const int first = pl_blocks(N + 1) + lane;
        if (first < PL_AMAX) p.amax_out[(int64_t)rb * PL_AMAX + first] = 0.0f;
    }
    WSTAMP(1);
    __syncthreads();
    WSTAMP(2);

    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
    if (ws.active) ring.run(acc, img, ws.s_first, ws.my_steps, lane);
    WSTAMP(3);
    wide_park(ws, acc, part, lane);
    __syncthreads();
    WSTAMP(4);
    if (ws.active && ws.kpart == 0) {
        wide_collect(ws, acc, part, lane);
        const DropGen drop = make_drop(p.drop_seed, p.drop_p, p.l);
        const int blk = ws.blk;
        if constexpr (NP == 2) {
            const float cinv = packed_inv(p.wp, nblk, nsteps, blk) * ainv;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] *= cinv;
        }
        with_act(p.act, [&](auto tag) {
            constexpr int ACT = decltype(tag)::value;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int n = 32 * blk + 4 * h + 8 * gq;
                const bool live = n < N;
                f32x4 m4 = {1.f, 1.f, 1.f, 1.f};
                if (drop.on) m4 = drop4(drop, vrow, n);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[4 * gq + e] + b4[gq][e];
                    if (drop.on) v *= m4[e];
                    acc[4 * gq + e] = live ? act_apply(v, ACT) : 0.0f;
                }
            }
        });
        Frag<NP> f[2];
        float osc = 1.0f;
        if (NP == 2 && p.tp_out) {
            const float bm = wide_block_scale(acc, sc, wave, lane, osc);
            if (lane == 0) p.amax_out[(int64_t)rb * PL_AMAX + blk] = blk == N / 32 ? fmaxf(bm, 1.0f) : bm;      // (the column of ones)
        }
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const f32x4 v0 = {acc[8 * t2], acc[8 * t2 + 1], acc[8 * t2 + 2], acc[8 * t2 + 3]};
            const f32x4 v1 = {acc[8 * t2 + 4], acc[8 * t2 + 5], acc[8 * t2 + 6], acc[8 * t2 + 7]};
            const int n = 32 * blk + 16 * t2 + 4 * h;
            if (p.a_out) {
                if (n < N) *reinterpret_cast<f32x4*>(p.a_out + (int64_t)vrow * N + n) = v0;
                if (n + 8 < N) *reinterpret_cast<f32x4*>(p.a_out + (int64_t)vrow * N + n + 8) = v1;
            }
            if (p.out && row_ok) {
                if (n < N) *reinterpret_cast<f32x4*>(p.out + arow * N + n) = v0;
                if (n + 8 < N) *reinterpret_cast<f32x4*>(p.out + arow * N + n + 8) = v1;
            }
            if (p.tp_out) f[t2] = make_frag<NP>(v0, v1, osc);
        }
        if (p.tp_out)
            emit_planes<NP>(p.tp_out + ((int64_t)blk * p.tp_steps + 2 * rb) * tile_bytes<NP>(), f, idf, lane, blk == N / 32 ? N % 32 : -1, rows_left, inv_tab);
    }
    if (p.tp_out && N % 32 == 0 && wave == PL_WAVES - 1 && nblk % p.G == g) {       // the column of ones opens a block of its own
        Frag<NP> z[2] = {};
        emit_planes<NP>(p.tp_out + ((int64_t)nblk * p.tp_steps + 2 * rb) * tile_bytes<NP>(), z, idf, lane, 0, rows_left, inv_tab);
        if (NP == 2 && lane == 0) p.amax_out[(int64_t)rb * PL_AMAX + nblk] = 1.0f;
    }
    WSTAMP(5);
}

// ---------------------------------------------------------------------------------------------
// backward: dZ_{l-1} = (dZ_l W_l) act'(a_{l-1}) mask_{l-1}, one launch per layer, top down; the top launch
// forms dZ_top first -- from d_out, or from the pair loss (the arithmetic of tower_dgrad_planes_kernel /
// loss.hip: fp64 per pair) -- and leaves it as the transposed image the weight-gradient launch reads.
// ---------------------------------------------------------------------------------------------
struct WideBwdP {
    int l, top;                    // this launch: dZ_l -> dZ_{l-1} (l >= 1) or -> dx (l == 0)
    int N, K;                      // dims[l + 1] (the sum), dims[l] (outputs)
    int act_prev;                  // activation behind layer l - 1
    int rows_call, n_calls, wpc, G;
    const float* dz_in;            // l < top: [virtual rows, N] row-major
    const char* wpt;               // packed W_l^T
    const float* a_prev;           // l >= 1: [virtual rows, K] the outputs of layer l - 1
    float* dz_out;                 // [virtual rows, K] (null: nobody reads it row-major)
    char* dzp_top;                 // l == top: out, transposed image of dZ_top
    char* dzp_out;                 // l >= 1: out, transposed image of dZ_{l-1}
    float* amax_top;               // fp16 x 2: the images' maxima (PL_AMAX floats per 32-row block)
    float* amax_out;
    int64_t tp_steps;
    float* dx;                     // l == 0: [rows, K] at the caller's row indices
    const unsigned long long* drop_seed;
    float drop_p;
    // the top launch
    const float* d_out;            // [rows, N] d loss / d output (or d loss / d z), caller's row indices; null: the pair loss
    int d_out_is_dz;
    const float* a_top;            // [rows, N] the tower's output, caller's row indices
    int act_top;
    int loss_kind, y_dtype;
    const void* y;
    double margin, scale;
    double* loss_partial;          // one per workgroup row of call 0
    unsigned* loss_counter;
    float* loss_out;
    const int* n_valid;
    double* loss_accum;
    const int64_t* g_steps;        // abn_tower_desc.source: the step's first pair (the labels' offset) and real-pair count
    const int32_t* g_ctr;
};

template <int NP>
__global__ __launch_bounds__(PL_NT) void wide_dgrad_layer_kernel(WideBwdP p)
{
    extern __shared__ __attribute__((aligned(16))) char pl_smem[];
    char* const img = pl_smem;
    float* const part = reinterpret_cast<float*>(pl_smem + PL_MAXSTEPS * NP * 1024);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int rb = blockIdx.x / p.G, g = blockIdx.x - rb * p.G;
    const int call = rb / p.wpc, lb = rb - call * p.wpc;
    const int vrow = rb * PL_ROWS + r;
    const int rows_left = p.rows_call - lb * PL_ROWS;
    const bool row_ok = r < rows_left;
    const int64_t arow = (int64_t)call * p.rows_call + lb * PL_ROWS + r;
    const int N = p.N, K = p.K;
    bf16x8 idf[2];
    make_identity<NP>(idf, lane);
    float* const sc = wd_scales<NP>(pl_smem);
    const float* const inv_tab = sc + PL_WAVES * 64 + wave * 32;
    float ainv = 1.0f;
    const int nblk = pl_blocks(K), nsteps = pl_steps(N);
    const WideShare ws(wave, g, p.G, nblk, nsteps);
    WideRing<NP> ring;
    if (p.wpt && ws.active) ring.issue(p.wpt, nblk, nsteps, ws.blk, ws.s_first, lane);      // the weights do not wait for dZ_l

    if (p.l == p.top) {
        // ---- dZ of the last layer for this workgroup's 32 rows (every one of the row block's G workgroups forms it)
        double* const coef = reinterpret_cast<double*>(reinterpret_cast<char*>(part) + WD_PART_BYTES);     // [32][2]
        double* const term_s = coef + 64;                                                                   // [32]
        int* const is_last_s = reinterpret_cast<int*>(term_s + 32);
        const bool with_loss = p.d_out == nullptr;
        const int B = p.rows_call;
        int64_t yoff = 0;
        int Bv = with_loss && p.n_valid ? *p.n_valid : B;
        const bool counted = with_loss && (p.n_valid || p.g_steps);
        if (with_loss && p.g_steps) {
            const int64_t st = *p.g_ctr;
            yoff = p.g_steps[2 * st];
            Bv = (int)p.g_steps[2 * st + 1];
        }
        const double lscale = counted && p.scale != 1.0 ? 1.0 / (double)(Bv > 0 ? Bv : 1) : p.scale;
        if (with_loss) {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int lr = 4 * wave + 2 * it + (lane >> 5), l = lane & 31;
                const int pi_ = lb * PL_ROWS + lr;                     // the pair of local row lr
                const bool ok = pi_ < B && pi_ < Bv;
                const int pc = pi_ < B ? pi_ : 0;
                double inv = 0.0, kself = 0.0, term = 0.0;
                const float* a = p.a_top + (int64_t)pc * N;            // e1[pair], e2[pair]: the order loss.hip sums in
                const float* b = p.a_top + (int64_t)(B + pc) * N;
                double dot = 0.0, s11 = 0.0, s22 = 0.0;
                for (int c = l; c < N / 4; c += 32) {
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
                    switch (p.y_dtype) {
                        case ABN_Y_I8: v = ((const int8_t*)p.y)[yoff + pi_]; break;
                        case ABN_Y_I32: v = ((const int32_t*)p.y)[yoff + pi_]; break;
                        case ABN_Y_I64: v = (double)((const int64_t*)p.y)[yoff + pi_]; break;
                        case ABN_Y_F32: v = ((const float*)p.y)[yoff + pi_]; break;
                        default: v = ((const double*)p.y)[yoff + pi_]; break;
                    }
                    const int code = v == 1.0 ? 1 : (v == -1.0 ? -1 : 0);
                    double dcos;
                    if (p.loss_kind == ABN_LOSS_COSCOS2) {
                        if (code == 1) { term = (1.0 - cs) * 0.5; dcos = -0.5; }
                        else if (code == -1) { term = cs * cs; dcos = 2.0 * cs; }
                        else { term = cs; dcos = 1.0; }
                    } else {
                        if (code == 1) { term = 1.0 - cs; dcos = -1.0; }
                        else if (code == -1) { const double hh = cs - p.margin; term = hh > 0.0 ? hh : 0.0; dcos = hh >= 0.0 ? 1.0 : 0.0; }
                        else { term = cs; dcos = 1.0; }
                    }
                    dcos *= lscale;
                    inv = dcos / (c1 * c2);
                    const double k1 = n1 > 0.0 ? dcos * cs / (c1 * n1) : 0.0;
                    const double k2 = n2 > 0.0 ? dcos * cs / (c2 * n2) : 0.0;
                    kself = call ? k2 : k1;
                }
                if (l == 0) { coef[2 * lr] = inv; coef[2 * lr + 1] = kself; term_s[lr] = term; }
            }
            __syncthreads();
            // a pair's term counts once: the workgroups of call 0 with g == 0 publish, the last of them adds up
            if (call == 0 && g == 0) {
                if (threadIdx.x == 0) {
                    double sum = 0.0;
                    for (int i = 0; i < 32; ++i) sum += term_s[i];
                    *is_last_s = abn_ticket_publish(&p.loss_partial[lb], sum, p.loss_counter, (unsigned)p.wpc);      // (common.h)
                }
                __syncthreads();
                if (*is_last_s && wave == 0) {
                    double sum = 0.0;
                    for (int i = lane; i < p.wpc; i += 64) sum += abn_ticket_partial(&p.loss_partial[i]);
#pragma unroll
                    for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 64);
                    if (lane == 0) {
                        const float lv = (float)(sum * lscale);
                        *p.loss_out = lv;
                        if (p.loss_accum) *p.loss_accum += (double)lv;
                        __hip_atomic_store(p.loss_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        }
        const int blocks_t = pl_steps(N) / 2;
        const DropGen drop_top = make_drop(p.drop_seed, p.drop_p, p.top);
        const float* const self_row = p.a_top + (row_ok ? arow : 0) * N;
        const float* const partner_row = p.a_top + (row_ok && with_loss ? (call ? arow - B : arow + B) : 0) * N;
        const double my_inv = with_loss ? coef[2 * r] : 0.0, my_k = with_loss ? coef[2 * r + 1] : 0.0;
        // one 16-feature step of dZ_top for this lane's row
        auto top_step = [&](int s, f32x4* v) {
            v[0] = v[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int c = 16 * s + 4 * h + 8 * u;
                if (row_ok && c < N) {
                    const f32x4 es = *reinterpret_cast<const f32x4*>(self_row + c);
                    if (with_loss) {
                        const f32x4 ep = *reinterpret_cast<const f32x4*>(partner_row + c);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float o = (float)(ep[e] * my_inv - es[e] * my_k);
                            if (p.act_top != ACT_NONE) o *= act_grad(es[e], p.act_top);
                            v[u][e] = o;
                        }
                        if (drop_top.on) {
                            const f32x4 m = drop4(drop_top, vrow, c);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[u][e] *= m[e];
                        }
                    } else {
                        v[u] = *reinterpret_cast<const f32x4*>(p.d_out + arow * N + c);
                        if (!p.d_out_is_dz) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[u][e] *= act_grad(es[e], p.act_top);
                            if (drop_top.on) {
                                const f32x4 m = drop4(drop_top, vrow, c);
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[u][e] *= m[e];
                            }
                        }
                    }
                }
            }
        };
        if constexpr (NP == 2) {
            f32x4 v[2][2][2];
            float m = 0.0f;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    const int kb = wave + PL_WAVES * u;
                    v[u][t2][0] = v[u][t2][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (kb < blocks_t) top_step(2 * kb + t2, v[u][t2]);
                    m = fmaxf(m, absmax8(v[u][t2][0], v[u][t2][1]));
                }
            sc[wave * 64 + lane] = m;
            __syncthreads();
            float osc;
            m = row_scales(sc, wave, lane, osc, ainv);
            if (g == 0 && wave == 0) store_amax_rows(p.amax_top + (int64_t)rb * PL_AMAX, m, 0.0f, lane);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int kb = wave + PL_WAVES * u;
                if (kb < blocks_t) {
                    Frag<NP> f[2];
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2) {
                        f[t2] = make_frag<NP>(v[u][t2][0], v[u][t2][1], osc);
                        store_frag<NP>(img + (int64_t)(2 * kb + t2) * (NP * 1024) + lane * 16, f[t2]);
                    }
                    if (kb < pl_blocks(N) && kb % p.G == g)
                        emit_planes<NP>(p.dzp_top + ((int64_t)kb * p.tp_steps + 2 * rb) * tile_bytes<NP>(), f, idf, lane, -1, 0, inv_tab);
                }
            }
        } else {
            for (int kb = wave; kb < blocks_t; kb += PL_WAVES) {
                Frag<NP> f[2];
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    f32x4 v[2];
                    top_step(2 * kb + t2, v);
                    f[t2] = make_frag<NP>(v[0], v[1]);
                    store_frag<NP>(img + (int64_t)(2 * kb + t2) * (NP * 1024) + lane * 16, f[t2]);
                }
                if (kb < pl_blocks(N) && kb % p.G == g)
                    emit_planes<NP>(p.dzp_top + ((int64_t)kb * p.tp_steps + 2 * rb) * tile_bytes<NP>(), f, idf, lane, -1, 0);
            }
        }
    } else {
        wide_stage_rows<NP>(p.dz_in + (int64_t)vrow * N, N, img, sc, wave, lane, ainv, [&](int, const Frag<NP>*) {});
    }
    __syncthreads();
    if (!p.wpt) return;                                // a one-layer tower without an input gradient: dZ_top was all there is
    if (NP == 2 && p.l >= 1 && g == 0 && wave == PL_WAVES - 1) {           // the blocks dZ_{l-1}'s image does not have
        const int first = nblk + lane;
        if (first < PL_AMAX) p.amax_out[(int64_t)rb * PL_AMAX + first] = 0.0f;
    }

    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
    f32x4 av[4] = {};
    if (ws.active && ws.kpart == 0 && p.l >= 1) {      // act'(a_{l-1}) wants the forward's output of layer l - 1
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int k = 32 * ws.blk + 4 * h + 8 * gq;
            av[gq] = *reinterpret_cast<const f32x4*>(p.a_prev + (int64_t)vrow * K + (k < K ? k : K - 4));
        }
    }
    if (ws.active) ring.run(acc, img, ws.s_first, ws.my_steps, lane);
    wide_park(ws, acc, part, lane);
    __syncthreads();
    if (!(ws.active && ws.kpart == 0)) return;
    wide_collect(ws, acc, part, lane);
    const int blk = ws.blk;
    if constexpr (NP == 2) {
        const float cinv = packed_inv(p.wpt, nblk, nsteps, blk) * ainv;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] *= cinv;
    }
    if (p.l >= 1) {
        const DropGen drop = make_drop(p.drop_seed, p.drop_p, p.l - 1);
        with_act(p.act_prev, [&](auto tag) {
            constexpr int ACT = decltype(tag)::value;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int k = 32 * blk + 4 * h + 8 * gq;
                const bool live = k < K && row_ok;
                f32x4 m4 = {1.f, 1.f, 1.f, 1.f};
                if (drop.on) m4 = drop4(drop, vrow, k);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[4 * gq + e] * act_grad(av[gq][e], ACT);
                    if (drop.on) v *= m4[e];
                    acc[4 * gq + e] = live ? v : 0.0f;
                }
            }
        });
    }
    Frag<NP> f[2];
    float osc = 1.0f;
    if (NP == 2 && p.l >= 1) {
        const float bm = wide_block_scale(acc, sc, wave, lane, osc);
        if (lane == 0) p.amax_out[(int64_t)rb * PL_AMAX + blk] = bm;
    }
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
        const f32x4 v0 = {acc[8 * t2], acc[8 * t2 + 1], acc[8 * t2 + 2], acc[8 * t2 + 3]};
        const f32x4 v1 = {acc[8 * t2 + 4], acc[8 * t2 + 5], acc[8 * t2 + 6], acc[8 * t2 + 7]};
        const int k = 32 * blk + 16 * t2 + 4 * h;
        if (p.l == 0) {
            if (row_ok && k < K) *reinterpret_cast<f32x4*>(p.dx + arow * K + k) = v0;
            if (row_ok && k + 8 < K) *reinterpret_cast<f32x4*>(p.dx + arow * K + k + 8) = v1;
        } else {
            if (p.dz_out) {
                if (k < K) *reinterpret_cast<f32x4*>(p.dz_out + (int64_t)vrow * K + k) = v0;
                if (k + 8 < K) *reinterpret_cast<f32x4*>(p.dz_out + (int64_t)vrow * K + k + 8) = v1;
            }
            f[t2] = make_frag<NP>(v0, v1, osc);
        }
    }
    if (p.l >= 1) emit_planes<NP>(p.dzp_out + ((int64_t)blk * p.tp_steps + 2 * rb) * tile_bytes<NP>(), f, idf, lane, -1, 0, inv_tab);
}

}  // namespace abn
