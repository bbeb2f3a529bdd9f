// tower_fused.h -- the whole tower forward in ONE launch (no BatchNorm, widths <= 512).
//
// Per-layer GEMM launches pay, per layer, a kernel boundary, a prologue and a
// store-bound epilogue during which no MFMA issues (~12-18 us of a ~50 us
// launch at C2).  Here a workgroup owns 32 rows (one MFMA row
// block) and walks all layers with its activations resident in LDS:
//
//   LDS  X[32][512]              the layer's input rows (fp32), rewritten in place;
//                                16-byte chunk c of row r sits at chunk c ^ (r & 7)
//                                of its 128-byte segment (2-way instead of 16-way
//                                conflicts for the MFMA A-fragment column reads)
//        Wst[wave][3][32*BPW][16]  per-WAVE ring of weight tiles filled by LDS-DMA
//
//   8 waves; wave w owns output columns  w*32*BPW .. +32*BPW  (BPW = 1 or 2 MFMA blocks)
//   k-loop: 16-deep tiles; the wave's own weight slice W[cols][k0..k0+16) comes in
//   by global_load_lds_dwordx4, two tiles ahead of the MFMAs (L2 latency under
//   this load is ~1.6 us).  The weight ring is private to the wave, so the k-loop
//   has NO workgroup barrier: a wave only waits on its own vmcnt.  Barriers
//   happen twice per layer (before X is overwritten, after).  Outputs stay in
//   accumulator registers for the whole layer (16*BPW <= 64 VGPRs), then bias +
//   dropout mask + activation -> X (next layer's input, zero padded to the k-tile)
//   and, asynchronously, -> HBM for the backward.
//
// The whole layer body is instantiated per BPW with its accumulators as LOCAL
// values of one loop: an accumulator array shared by several call sites made
// hipcc copy AGPR ranges on every loop back-edge (v_accvgpr_mov x32 per tile,
// each waiting for the MFMA it reads: 133 instead of 64 cycles per MFMA, measured).
//
// FLOPs per workgroup equal the per-layer path's; weights are re-read from L2 by
// every workgroup (256 x 2.3 MB at C2).
#pragma once
#include "gemm_f32.h"

namespace abn {

constexpr int FUSED_ROWS = 32;
constexpr int FUSED_MAXW = 512;                 // widest layer the LDS image holds
constexpr int FUSED_XS = 512;                   // row stride of X (swizzled chunks)
constexpr int FUSED_BK = 16;                    // k-tile of the weight ring
constexpr int FUSED_STAGES = 3;                 // ring depth: two tiles in flight
constexpr int FUSED_WAVES = 8;                  // two wavefronts per SIMD: one's LDS / DMA waits hide behind the other's MFMAs
constexpr int FUSED_NT = 64 * FUSED_WAVES;
constexpr int FUSED_MAXBPW = FUSED_MAXW / 32 / FUSED_WAVES;      // 2 MFMA column blocks per wave at the widest layer
constexpr int FUSED_WTILE = 32 * FUSED_MAXBPW * FUSED_BK;        // floats of one weight stage of one wave
constexpr size_t FUSED_LDS_BYTES = sizeof(float) * (FUSED_ROWS * FUSED_XS + FUSED_WAVES * FUSED_STAGES * FUSED_WTILE);
static_assert(FUSED_LDS_BYTES <= 160 * 1024, "fused tower image must fit the 160 KiB LDS");

struct FusedFwdP {
    int n_layers;
    int rows;                      // total rows (both towers)
    int rows_call;                 // rows per forward_once call (x1 | x2 split)
    int bf16;                      // throughput mode: operands rounded to bf16 at fragment time
    int dims[ABN_MAX_LAYERS + 1];
    int act[ABN_MAX_LAYERS];
    const float* x1;
    const float* x2;               // may be null: all rows in x1
    float* x_copy;                 // [rows, dims[0]] concatenated copy for the backward (may be null)
    const float* W[ABN_MAX_LAYERS];
    const float* b[ABN_MAX_LAYERS];
    const float* mask[ABN_MAX_LAYERS];
    float* out[ABN_MAX_LAYERS];    // [rows, dims[l+1]] post-activation outputs
#ifdef ABN_STAMPS
    unsigned long long* stamps;
#endif
};

#ifdef ABN_STAMPS
#define FSTAMP(slot) do { if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 128 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FSTAMP(slot) do {} while (0)
#endif

// float offset of element (row, k) of the swizzled X image
__device__ __forceinline__ int x_off(int row, int k)
{
    return row * FUSED_XS + (k & ~31) + 4 * (((k >> 2) & 7) ^ (row & 7)) + (k & 3);
}

// One layer for one workgroup.  X holds the input rows (zero beyond K up to the
// next multiple of 32); on return it holds this layer's output the same way.
// KS = 2 (narrow layers, at most 4 column blocks): waves 4..7 take the second half of
// K for the same columns as waves 0..3 and hand their partial sums over through LDS,
// so that every SIMD still runs two wavefronts.
template <int BPW, int KS, int BF16>
__device__ __forceinline__ void fused_layer(const FusedFwdP& p, int l, float* __restrict__ X,
                                            float* __restrict__ Wst, int wave, int lane, int row0)
{
    constexpr int NINSTR = 2 * BPW;              // DMA instructions per tile: 16 rows x 64 B each
    constexpr int CW = 32 * BPW;
    const int K = p.dims[l], N = p.dims[l + 1];
    const float* __restrict__ W = p.W[l];
    const int rl = lane & 31, h = lane >> 5;
    const int nkt_all = (K + 31) / 32 * 2;       // 16-deep tiles over K padded to 32 (X is zero there)
    const int khalf = KS == 2 ? wave / (FUSED_WAVES / 2) : 0;
    const int kt_first = KS == 2 ? khalf * (nkt_all / 2) : 0;
    const int nkt = KS == 2 ? nkt_all / 2 : nkt_all;                     // tiles this wave sweeps (nkt_all is even)
    const int col0 = (KS == 2 ? wave % (FUSED_WAVES / 2) : wave) * CW;

    f32x16 acc[BPW];
#pragma unroll
    for (int j = 0; j < BPW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    // DMA of one weight tile: CW rows (output columns) x 16 k.  Rows past N and k
    // past K are clamped to valid memory: their products land in unused columns /
    // meet the zero padding of X.  Tile image: unit u = n*4 + c, chunk c of row n
    // holds logical chunk c ^ ((n >> 1) & 3).  Everything but k0 is loop
    // invariant and hoisted (32-bit offsets: the host checks W < 2^31 floats):
    // recomputing the 64-bit row products per tile cost ~1000 cycles per tile.
    uint32_t rowoff[NINSTR];
#pragma unroll
    for (int i = 0; i < NINSTR; ++i) {
        int gn = col0 + 16 * i + (lane >> 2);
        gn = gn < N ? gn : N - 1;
        rowoff[i] = (uint32_t)gn * (uint32_t)K;
    }
    const int kc_lane = 4 * ((lane & 3) ^ ((lane >> 3) & 3));      // (n >> 1) & 3 == (lane >> 3) & 3
    auto dma = [&](int kt, int stage) {
#ifdef FEXP_NODMA
        return;
#endif
        float* dst = Wst + stage * FUSED_WTILE;
        int gk = (kt_first + kt) * FUSED_BK + kc_lane;
        gk = gk <= K - 4 ? gk : K - 4;
        const float* src = W + gk;
#pragma unroll
        for (int i = 0; i < NINSTR; ++i)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + rowoff[i]),
                (__attribute__((address_space(3))) void*)(dst + 256 * i), 16, 0, 0);
    };
    FSTAMP(1 + 3 * l);
    const int npad = (N + 31) / 32 * 32;
    const bool active = col0 < npad;              // narrow layers leave the upper waves without columns
    if (active) {
    dma(0, 0);
    if (nkt > 1) dma(1, 1);
    // lane-invariant parts of the fragment addresses
    const float* const xrow = X + rl * FUSED_XS;
    const int xsw = rl & 7;
    int boff[BPW][2];                             // B fragment offset inside a stage, per block and k-group
#pragma unroll
    for (int j = 0; j < BPW; ++j)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int n = 32 * j + rl;
            boff[j][g] = n * FUSED_BK + 4 * ((2 * g + h) ^ ((n >> 1) & 3));
        }
    auto load_frags = [&](f32x4& fa, f32x4* fb, const float* ws, int k0, int g) {
        const int kc = k0 + 8 * g + 4 * h;                                  // multiple of 4: one chunk
        fa = *reinterpret_cast<const f32x4*>(xrow + (kc & ~31) + 4 * (((kc >> 2) & 7) ^ xsw));
#pragma unroll
        for (int j = 0; j < BPW; ++j) fb[j] = *reinterpret_cast<const f32x4*>(ws + boff[j][g]);
    };
    auto mfmas = [&](const f32x4& fa, const f32x4* fb) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < BPW; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[j][e], acc[j], 0, 0, 0);
    };
    for (int kt = 0; kt < nkt; ++kt) {
        // tile kt has landed when at most the NEXT tile's DMAs are outstanding
        // (vmcnt retires in order; older activation stores drain first)
        if (kt + 1 < nkt) {
            if constexpr (BPW == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // stage (kt+2)%3 == (kt-1)%3 was read during tile kt-1: free now
        if (kt + 2 < nkt) dma(kt + 2, (kt + 2) % FUSED_STAGES);
        const float* ws = Wst + (kt % FUSED_STAGES) * FUSED_WTILE;
        const int k0 = (kt_first + kt) * FUSED_BK;
        // with one wave per SIMD nothing but the wave itself hides LDS latency:
        // the second k-group's fragments are read before the first group's MFMAs
        f32x4 fa0, fa1, fb0[BPW], fb1[BPW];
#ifdef FEXP_NOFRAG
        fa0 = fa1 = f32x4{(float)kt, 1.f, 2.f, 3.f};
        for (int j = 0; j < BPW; ++j) fb0[j] = fb1[j] = f32x4{(float)lane, 1.f, (float)kt, 3.f};
#else
        load_frags(fa0, fb0, ws, k0, 0);
        load_frags(fa1, fb1, ws, k0, 1);
#endif
#ifdef FEXP_NOMFMA
        for (int j = 0; j < BPW; ++j) acc[j][kt & 15] += fa0[0] * fb0[j][1] + fa1[2] * fb1[j][3];
        continue;
#endif
        if constexpr (BF16 == 2) {                // bf16 x 3 (gemm_f32.h): six bf16 MFMAs per column block and tile
            const bf16x8x3 pa = split_bf16x3(fa0, fa1);
#pragma unroll
            for (int j = 0; j < BPW; ++j) {
                const bf16x8x3 pb = split_bf16x3(fb0[j], fb1[j]);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa.lo, pb.hi, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa.hi, pb.lo, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa.mid, pb.mid, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa.mid, pb.hi, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa.hi, pb.mid, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa.hi, pb.hi, acc[j], 0, 0, 0);
            }
        } else if constexpr (BF16 == 1) {         // one 16-deep bf16 MFMA per column block and tile
            const bf16x8 pa = pack_bf16(fa0, fa1);
#pragma unroll
            for (int j = 0; j < BPW; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, pack_bf16(fb0[j], fb1[j]), acc[j], 0, 0, 0);
        } else {
            mfmas(fa0, fb0);
            mfmas(fa1, fb1);
        }
    }
    }
    // epilogue: bias, dropout mask, activation -> X (input of the next layer), zero up to
    // the next multiple of 32 (the next layer's k padding).  The waves leave the k-loop
    // a few thousand cycles apart (each streams its own weights), so everything that
    // does not touch X -- bias, mask, activation, in registers -- runs BEFORE the
    // barrier, in the time the wave would otherwise spend waiting for the slowest one.
    const int act = p.act[l];
    const float* __restrict__ mask = p.mask[l];
    auto finish = [&]() {
#pragma unroll
        for (int j = 0; j < BPW; ++j) {
            const int col = col0 + 32 * j + rl;
            const int colc = col < N ? col : 0;
            const float bias = p.b[l] ? p.b[l][colc] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[j][r] + bias;
                if (mask) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int gr = min(row0 + row, p.rows - 1);
                    v *= mask[(int64_t)gr * N + colc];
                }
                acc[j][r] = act_apply(v, act);
            }
        }
    };
    if (KS == 1) {
        if (active) finish();
    } else if (active && khalf == 1) {             // partial sums of the upper K half -> this wave's (idle) ring
#pragma unroll
        for (int r = 0; r < 16; ++r) Wst[r * 64 + lane] = acc[0][r];
    }
    FSTAMP(2 + 3 * l);
    __syncthreads();                               // every wave is done reading X
    if (KS == 2 && active && khalf == 0) {
        const float* part = Wst + (FUSED_WAVES / 2) * FUSED_STAGES * FUSED_WTILE;   // wave + 4's ring
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] += part[r * 64 + lane];
        finish();
    }
#pragma unroll
    for (int j = 0; j < BPW; ++j) {
        const int col = col0 + 32 * j + rl;
        if (col < npad && khalf == 0) {         // one branch per 32-column block, not per element
            const bool live = col < N;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                X[x_off(row, col)] = live ? acc[j][r] : 0.0f;
            }
        }
    }
    __syncthreads();
    FSTAMP(3 + 3 * l);
    // X -> HBM (saved activation for the backward / the embedding), full rows,
    // 16 bytes per lane; asynchronous: nobody waits for these stores
    float* __restrict__ out = p.out[l];
#ifdef FEXP_NOOUT
    if (p.rows > 0) return;
#endif
    const int n4 = N / 4;
    for (int i = threadIdx.x; i < FUSED_ROWS * n4; i += FUSED_NT) {
        const int r = i / n4, c = 4 * (i % n4);
        if (row0 + r < p.rows)
            *reinterpret_cast<f32x4*>(out + (int64_t)(row0 + r) * N + c) =
                *reinterpret_cast<const f32x4*>(X + x_off(r, c));
    }
}

template <int BF16>
__global__ __launch_bounds__(FUSED_NT) void tower_fwd_fused_kernel(FusedFwdP p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const X = smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* const Wst = smem + FUSED_ROWS * FUSED_XS + wave * FUSED_STAGES * FUSED_WTILE;
    const int row0 = blockIdx.x * FUSED_ROWS;
    const int D0 = p.dims[0];
    FSTAMP(0);

    // input rows -> X (zero padded), + the concatenated copy for the backward
    for (int i = threadIdx.x; i < FUSED_ROWS * (FUSED_XS / 4); i += FUSED_NT) {
        const int r = i / (FUSED_XS / 4), c = 4 * (i % (FUSED_XS / 4));
        const int gr = row0 + r;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gr < p.rows && c < D0) {
            const float* src = (p.x2 && gr >= p.rows_call) ? p.x2 + (int64_t)(gr - p.rows_call) * D0 : p.x1 + (int64_t)gr * D0;
            v = *reinterpret_cast<const f32x4*>(src + c);
            if (p.x_copy) *reinterpret_cast<f32x4*>(p.x_copy + (int64_t)gr * D0 + c) = v;
        }
        *reinterpret_cast<f32x4*>(X + x_off(r, c)) = v;
    }
    __syncthreads();

    for (int l = 0; l < p.n_layers; ++l) {
        const int N = p.dims[l + 1];
        if (N > 32 * FUSED_WAVES) fused_layer<2, 1, BF16>(p, l, X, Wst, wave, lane, row0);
        else if (N > 16 * FUSED_WAVES) fused_layer<1, 1, BF16>(p, l, X, Wst, wave, lane, row0);
        else fused_layer<1, 2, BF16>(p, l, X, Wst, wave, lane, row0);
    }
    FSTAMP(1 + 3 * p.n_layers);
}

}  // namespace abn
