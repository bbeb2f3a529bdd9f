// tower_wgrad_step.h -- small batches (the reference's canonical regime: 8 word pairs = a few hundred 280-d frame pairs per
// step, abnet3/dataloader.py:248-255): every layer's weight gradient dW_l = dZ_l^T [A_{l-1} | 1] AND the optimizer's rule in
// ONE launch, fp16 x 2 arithmetic.
//
// wgrad_planes_kernel cuts the sum over the batch rows into slabs so that a 128 x 256 tile of one layer fills the chip at
// 8192 rows; slab_reduce_step_kernel adds the slabs and steps the parameters; at ~1000 rows that is split-K machinery with
// nothing to split (16.3 + 10.0 us and a launch boundary for 0.9 GFLOP).  Here a workgroup owns one 64 x 64 tile of
// [dW_l | db_l] OUTRIGHT -- 184 tiles for 280 -> 500 x 2 -> 100 -- and sums all of the batch's row steps itself:
//   * eight waves, each summing every EIGHTH row step of the WHOLE tile: four operand tiles per step (two dZ blocks, two
//     [A | 1] blocks) straight from the transposed images into registers (eight 16-byte buffer loads per lane, four steps
//     in flight), split there (make_frag: no LDS staging, no barrier inside the sum), twelve MFMAs on four accumulators
//     -- 2 KB of operands per block product; a wave per 32 x 32 block needs 4 KB, and the launch is bound by what a CU
//     can pull from its L2 (measured: 27 us that way);
//   * one power of two per operand from the images' tables of maxima (all rows: there is one "slab"), the column of ones
//     under a scale of its own, exactly as wgrad_tile does;
//   * every wave parks its accumulators in LDS (128 KB); each then adds the eight partial sums of ITS eighth of the tile
//     in wave order and applies opt_update_reg element by element: gradient, parameter and optimizer state are read and
//     written once, 128 bytes per half-wave; the reads are issued before the workgroup waits for its slowest wave.
// Rows past a call's end are zero rows of dZ in the images (tower_wide.h): a padded batch sums the same numbers in the same
// order as the unpadded one.  The launch is abn_tower_reduce_step's when the backward deferred its reduction (one process,
// nothing between backward and step) and lent the forward's workspace (abn_tower_desc.fwd_ws): the layer-per-launch kernels
// of tower_wide.h, fp16 x 2, no BatchNorm.
#pragma once
#include "opt_rule.h"
#include "tower_planes.h"

namespace abn {

struct WgsLayer {
    const char* dzp;         // transposed image of dZ_l        [nblk][tp_steps] tiles
    const char* ap;          // transposed image of [A_{l-1} | 1] [kblk][tp_steps]
    const float* amax_dz;    // the images' maxima, PL_AMAX floats per 32-row block
    const float* amax_a;
    int N, K, nblk, kblk;
    int tiles_k;             // 64-wide tiles along K (+ the bias column)
    int first_wg;            // index of the layer's first tile among all layers' tiles
    int64_t w_off, b_off;    // float offsets of W_l / b_l in the flat parameter, gradient and state buffers
};
struct WgsP {
    int n_layers;
    WgsLayer L[ABN_MAX_LAYERS];
    int tp_steps;            // row steps of 16 (virtual rows / 16): even
    OptP o;
    float* params;
    float* grads;
    float* s1;
    float* s2;
    const unsigned* fail_word;
    int32_t* step_ctr;       // abn_step_source: advanced once per step (or null)
    int n_tiles, per_xcd;    // tiles of all layers (L[].first_wg: a layer's first); tiles per XCD = ceil(n_tiles / 8)
};

constexpr int WGS_DEPTH = 4;                       // row steps in flight per wave
constexpr int WGS_WAVES = 8;
constexpr int WGS_NT = 64 * WGS_WAVES;
constexpr size_t WGS_LDS_BYTES = (size_t)WGS_WAVES * 4 * 16 * 64 * 4 + 2 * WGS_WAVES * 4;      // every wave's four accumulators + the maxima

#ifdef ABN_WGS_STAMPS          // diagnostic build only (tools/wgs_stamps.py)
__device__ unsigned long long g_wgs_stamps[512][8];
#define WGS_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 512) g_wgs_stamps[blockIdx.x][k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define WGS_STAMP(k) do {} while (0)
#endif
__global__ __launch_bounds__(WGS_NT) void wgrad_step_small_kernel(WgsP p)
{
    extern __shared__ __attribute__((aligned(16))) char wgs_smem[];
    float (*park)[4][16][64] = reinterpret_cast<float (*)[4][16][64]>(wgs_smem);                 // [wave][block][q][lane]
    float* const red = reinterpret_cast<float*>(wgs_smem + (size_t)WGS_WAVES * 4 * 16 * 64 * 4);  // [2][waves]
    if (p.step_ctr && blockIdx.x == 0 && threadIdx.x == 0) *p.step_ctr += 1;      // (every reader of the step's entry ran in the launches before)
    if (p.fail_word && *p.fail_word != 0u) return;
    WGS_STAMP(0);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Placement.  Every tile reads ALL row steps of its two dZ blocks and its two [A | 1] blocks, and the images were
    // written by the launches before (another XCD's L2, or none): dealt out in launch order each of the eight L2s would
    // fetch every image whole.  Workgroups b and b + 8 are observed to share an XCD (round-robin dispatch: speed only), so
    // XCD x gets a CONTIGUOUS run of the tiles in (layer, tile row, tile column) order -- p.per_xcd of them: three tile
    // rows of one layer at C5's shapes, i.e. 3/8 of that layer's dZ image and its [A | 1] image once.
    const int ti = (int)(blockIdx.x & 7) * p.per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= p.per_xcd || ti >= p.n_tiles) return;
    int li = 0;
    while (li + 1 < p.n_layers && ti >= p.L[li + 1].first_wg) ++li;
    const WgsLayer& L = p.L[li];
    const int local = ti - L.first_wg;
    const int tiles_n = (L.nblk + 1) / 2;
    const int tn = local / L.tiles_k, tk = local - tn * L.tiles_k;
    if (tn >= tiles_n) return;
    constexpr int FR = tile_bytes<2>();
    // blocks past the matrix are clamped copies of the last one: computed, never stored
    const int nb0 = 2 * tn, kb0 = 2 * tk;
    const int nbc[2] = {nb0 < L.nblk ? nb0 : L.nblk - 1, nb0 + 1 < L.nblk ? nb0 + 1 : L.nblk - 1};
    const int kbc[2] = {kb0 < L.kblk ? kb0 : L.kblk - 1, kb0 + 1 < L.kblk ? kb0 + 1 : L.kblk - 1};

    // the operands' scales: the largest magnitude of either image over ALL rows
    // (one 16-byte load per table and thread: ONE round trip -- the host keeps tp_steps / 2 * PL_AMAX <= 4 * WGS_NT)
    float md = 0.0f, ma = 0.0f;
    const int cnt4 = p.tp_steps / 2 * PL_AMAX / 4;
    if ((int)threadIdx.x < cnt4) {
        const f32x4 d4 = reinterpret_cast<const f32x4*>(L.amax_dz)[threadIdx.x], a4 = reinterpret_cast<const f32x4*>(L.amax_a)[threadIdx.x];
        md = fmaxf(fmaxf(d4[0], d4[1]), fmaxf(d4[2], d4[3]));
        ma = fmaxf(fmaxf(a4[0], a4[1]), fmaxf(a4[2], a4[3]));
    }
    md = wave_max(md); ma = wave_max(ma);
    if (lane == 0) { red[wave] = md; red[WGS_WAVES + wave] = ma; }

    // wave w sums row steps w, w + 8, ... of the WHOLE tile: four tiles per step (two dZ blocks, two [A | 1] blocks: 8 KB
    // for four block products -- half the bytes per product of a wave per block; the launch is bound by what a CU can pull)
    __amdgpu_buffer_rsrc_t rs[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
        rs[b] = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>((b < 2 ? L.dzp : L.ap) + (int64_t)(b < 2 ? nbc[b] : kbc[b - 2]) * p.tp_steps * FR),
                                                  0, p.tp_steps * FR, 0x00020000);
    const int n_mine = (p.tp_steps - wave + WGS_WAVES - 1) / WGS_WAVES;
    v4i rq[WGS_DEPTH][8];
    auto load = [&](int slot, int i) {
        const int off = (wave + WGS_WAVES * (i < n_mine ? i : n_mine - 1)) * FR;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            rq[slot][2 * b] = __builtin_amdgcn_raw_buffer_load_b128(rs[b], lane * 16, off, 0);
            rq[slot][2 * b + 1] = __builtin_amdgcn_raw_buffer_load_b128(rs[b], lane * 16, off + 1024, 0);
        }
    };
    if (n_mine > 0) {
#pragma unroll
        for (int i = 0; i < WGS_DEPTH; ++i) load(i, i);
    }
    // (a barrier that orders LDS only: __syncthreads() would also wait for the 32 tile loads each lane has just issued)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    WGS_STAMP(1);
#pragma unroll
    for (int w = 0; w < WGS_WAVES; ++w) { md = fmaxf(md, red[w]); ma = fmaxf(ma, red[WGS_WAVES + w]); }
    float sd, id, sa, ia;
    scale_of(md, sd, id);
    scale_of(ma, sa, ia);
    // the column of ones (the bias gradient) keeps a scale of its own, 1 (wgrad_tile, tower_planes.h)
    float sb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) sb[j] = (kb0 + j == L.K / 32 && (lane & 31) == L.K % 32) ? 1.0f : sa;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;
    if (n_mine > 0) {
        const int n4 = (n_mine + WGS_DEPTH - 1) / WGS_DEPTH * WGS_DEPTH;
        for (int i0 = 0; i0 < n4; i0 += WGS_DEPTH) {
#pragma unroll
            for (int u = 0; u < WGS_DEPTH; ++u) {
                const int i = i0 + u;
                const bool real = i < n_mine;          // (steps past the end: clamped re-reads under a ZERO scale add nothing)
                Frag<2> fa[2], fb[2];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    fa[b] = make_frag<2>(__builtin_bit_cast(f32x4, rq[u][2 * b]), __builtin_bit_cast(f32x4, rq[u][2 * b + 1]), real ? sd : 0.0f);
                    fb[b] = make_frag<2>(__builtin_bit_cast(f32x4, rq[u][4 + 2 * b]), __builtin_bit_cast(f32x4, rq[u][4 + 2 * b + 1]), real ? sb[b] : 0.0f);
                }
                load(u, i + WGS_DEPTH);
#pragma unroll
                for (int t = 0; t < Products<2>::N; ++t)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) acc[a][b] = pl_mfma<2>(fa[a].p[Products<2>::A[t]], fb[b].p[Products<2>::B[t]], acc[a][b]);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WGS_STAMP(2);
    // This wave's share of the result: block (i, j) = wave & 3, accumulator registers 8 (wave >> 2) .. + 7.  Its parameter,
    // state and (for the write) gradient addresses; the reads are issued BEFORE the partial sums are parked and the
    // workgroup waits for its slowest wave: they arrive under that wait.
    const int bi = (wave >> 1) & 1, bj = wave & 1, q0 = 8 * (wave >> 2);
    const int nb = nb0 + bi, kb = kb0 + bj;
    const int h = lane >> 5, k = 32 * kb + (lane & 31);
    const bool col_ok = nb < L.nblk && kb < L.kblk && k <= L.K;
    const bool two = opt_uses_s2(p.o);
    float pv[8], a1[8], a2[8];
    int idx[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int q = q0 + u;
        const int n = 32 * nb + (q & 3) + 8 * (q >> 2) + 4 * h;
        idx[u] = col_ok && n < L.N ? (int)(k < L.K ? L.w_off + (int64_t)n * L.K + k : L.b_off + n) : -1;
        const int j = idx[u] >= 0 ? idx[u] : (int)L.w_off;
        pv[u] = p.params[j];
        a1[u] = p.s1[j];
        a2[u] = two ? p.s2[j] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) park[wave][2 * i + j][q][lane] = acc[i][j][q];
    __syncthreads();
    WGS_STAMP(3);
    const float out_inv = k < L.K ? id * ia : id;
    float g[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        float sum = park[0][2 * bi + bj][q0 + u][lane];
#pragma unroll
        for (int w = 1; w < WGS_WAVES; ++w) sum += park[w][2 * bi + bj][q0 + u][lane];      // (the waves in order: deterministic)
        g[u] = sum * out_inv;
        pv[u] = opt_update_reg(p.o, pv[u], g[u], a1[u], a2[u]);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        if (idx[u] < 0) continue;
        p.grads[idx[u]] = g[u];
        p.params[idx[u]] = pv[u];
        p.s1[idx[u]] = a1[u];
        if (two) p.s2[idx[u]] = a2[u];
    }
#ifdef ABN_WGS_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WGS_STAMP(4);
#endif
}

}  // namespace abn
