// gemm_f32.h -- LDS-tiled exact-fp32 GEMM on the gfx950 matrix cores
// (v_mfma_f32_32x32x2_f32), the workhorse of the Siamese tower.
//
//   C[m][n] = sum_k A(m,k) * B(n,k)
//
// Each operand is given either "K-contiguous" (X[m*ld + k], the layout of an
// activation matrix or of an nn.Linear weight used in forward) or "k-major"
// (X[k*ld + m], what the same tensors look like when the reduction runs over
// their ROW index, as in dgrad / wgrad).  The three tower products map to
//   forward  Y = X W^T      A = X  (K-contig)   B = W   (K-contig)
//   dgrad    dX = dZ W      A = dZ (K-contig)   B = W   (k-major)
//   wgrad    dW = dZ^T X    A = dZ (k-major)    B = X   (k-major), split-K
// (reference op sequences: SURVEY.md section 2, "Linear fwd / bwd").
//
// Design for CDNA4:
//  * one workgroup = 4 waves (256 threads), one wave per SIMD, each wave owns a
//    (BM/2) x (BN/2) block of C as TM x TN accumulators of 32x32 (16 VGPRs each);
//  * the fp32 MFMA issues one 32x32x2 product per 64 cycles per SIMD, so LDS
//    bandwidth is never the limit; what matters is keeping the global->LDS
//    stream ahead of it: register-staged double buffering, one barrier per
//    32-deep k-tile (cdna_hip_programming.md T14 order: issue loads, compute,
//    then write LDS);
//  * K-contiguous tiles are stored [row][32+4] and read with ds_read_b128
//    (row stride 36 dwords is conflict-free for the 16-lane b128 groups);
//    k-major tiles are stored [k][BM+4] and read with ds_read_b32 (32
//    consecutive dwords per half-wave: conflict-free);
//  * the k index consumed by lane-half h at MFMA step (g,e) is 8g+4h+e for both
//    operands, so a 16-byte LDS read feeds four consecutive MFMAs;
//  * blockIdx -> tile mapping gives each XCD a contiguous run of tiles so that
//    the N-tiles sharing an A row-block hit the same L2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace abn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { EPI_FWD = 0, EPI_DGRAD = 1, EPI_WGRAD = 2 };
enum { ACT_NONE = 0, ACT_SIGMOID = 1, ACT_RELU = 2, ACT_TANH = 3 };

struct GemmP {
    const float* A; int64_t lda;
    const float* B; int64_t ldb;
    float* C; int64_t ldc;
    int M, N, K;
    int k_chunk;            // split-K: blockIdx.y owns k in [y*k_chunk, +k_chunk)
    const float* bias;      // FWD: [N] or nullptr
    int act;                // FWD: activation applied; DGRAD: act of `aux`
    const float* aux;       // DGRAD: activation OUTPUT a_prev[M][N] (ld = ldaux)
    int64_t ldaux;
    float* C2;              // WGRAD: destination of the ones column (bias grad)
    int64_t slab_stride;    // WGRAD: floats between split-K slabs of C / C2
    int ones_col;           // WGRAD: column of B that is identically 1, or -1
    int a_vec, b_vec;       // 16-byte global loads allowed for A / B
};

__device__ __forceinline__ float act_apply(float z, int act)
{
    if (act == ACT_SIGMOID) return 1.0f / (1.0f + expf(-z));
    if (act == ACT_RELU) return z > 0.0f ? z : 0.0f;
    if (act == ACT_TANH) return tanhf(z);
    return z;
}

// derivative expressed on the activation output (what autograd keeps)
__device__ __forceinline__ float act_grad(float a, int act)
{
    if (act == ACT_SIGMOID) return a * (1.0f - a);
    if (act == ACT_RELU) return a > 0.0f ? 1.0f : 0.0f;
    if (act == ACT_TANH) return 1.0f - a * a;
    return 1.0f;
}

constexpr int BK = 32;
constexpr int KPAD = 4;     // K-contiguous tiles: row stride 36 dwords
constexpr int MPAD = 4;     // k-major tiles: row stride BM+4 dwords

template <int BMN, bool KCONTIG>
struct TileShape {
    static constexpr int rows = KCONTIG ? BMN : BK;
    static constexpr int stride = KCONTIG ? (BK + KPAD) : (BMN + MPAD);
    static constexpr int floats = rows * stride;
    static constexpr int units = BMN * BK / 4;          // float4 units
    static constexpr int per_thread = units / 256;
};

// Loads this thread's float4 units of one operand tile into registers.
// mn0: first row(M/N index) of the tile, k0: first k, k_end: exclusive k bound.
template <int BMN, bool KCONTIG>
__device__ __forceinline__ void tile_load(f32x4* r,
                                          const float* __restrict__ X, int64_t ld,
                                          int MN, int mn0, int k0, int k_end, bool vec,
                                          int ones_col)
{
    constexpr int PT = TileShape<BMN, KCONTIG>::per_thread;
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int u = t + 256 * i;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if constexpr (KCONTIG) {
            const int row = u >> 3, kq = u & 7;
            const int m = mn0 + row, k = k0 + 4 * kq;
            if (m < MN) {
                const float* p = X + (int64_t)m * ld + k;
                if (vec && k + 3 < k_end) {
                    v = *reinterpret_cast<const f32x4*>(p);
                } else {
                    if (k + 0 < k_end) v.x = p[0];
                    if (k + 1 < k_end) v.y = p[1];
                    if (k + 2 < k_end) v.z = p[2];
                    if (k + 3 < k_end) v.w = p[3];
                }
            }
        } else {
            constexpr int UPR = BMN / 4;            // units per k row
            const int kk = u / UPR, mq = u % UPR;
            const int k = k0 + kk, m = mn0 + 4 * mq;
            if (k < k_end) {
                const float* p = X + (int64_t)k * ld + m;
                if (vec && m + 3 < MN) {
                    v = *reinterpret_cast<const f32x4*>(p);
                } else {
                    if (m + 0 < MN) v.x = p[0];
                    if (m + 1 < MN) v.y = p[1];
                    if (m + 2 < MN) v.z = p[2];
                    if (m + 3 < MN) v.w = p[3];
                }
                if (ones_col >= 0) {                // wgrad: bias column of ones
                    const int d = ones_col - m;
                    if (d == 0) v.x = 1.0f;
                    if (d == 1) v.y = 1.0f;
                    if (d == 2) v.z = 1.0f;
                    if (d == 3) v.w = 1.0f;
                }
            }
        }
        r[i] = v;
    }
}

template <int BMN, bool KCONTIG>
__device__ __forceinline__ void tile_store(const f32x4* r,
                                           float* __restrict__ lds)
{
    constexpr int PT = TileShape<BMN, KCONTIG>::per_thread;
    constexpr int ST = TileShape<BMN, KCONTIG>::stride;
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int u = t + 256 * i;
        int off;
        if constexpr (KCONTIG) off = (u >> 3) * ST + 4 * (u & 7);
        else { constexpr int UPR = BMN / 4; off = (u / UPR) * ST + 4 * (u % UPR); }
        *reinterpret_cast<f32x4*>(lds + off) = r[i];
    }
}

// Fragment of one 32-row MFMA operand block for k-group g (8 k values):
// element e is the operand of MFMA step (g,e); k = 8g + 4*(lane>>5) + e.
template <int BMN, bool KCONTIG>
__device__ __forceinline__ f32x4 frag_read(const float* __restrict__ lds, int row0, int g, int lane)
{
    constexpr int ST = TileShape<BMN, KCONTIG>::stride;
    const int r = row0 + (lane & 31), h = lane >> 5;
    if constexpr (KCONTIG) {
        return *reinterpret_cast<const f32x4*>(lds + r * ST + 8 * g + 4 * h);
    } else {
        const float* p = lds + (8 * g + 4 * h) * ST + r;
        f32x4 v;
        v.x = p[0]; v.y = p[ST]; v.z = p[2 * ST]; v.w = p[3 * ST];
        return v;
    }
}

// XCD-aware tile order: blocks b, b+8, b+16, ... share an XCD (and its L2), so
// hand XCD x the contiguous tile range [x*n/8, (x+1)*n/8) when n % 8 == 0.
__device__ __forceinline__ int xcd_tile_index(int b, int n)
{
    if ((n & 7) != 0) return b;
    return (b & 7) * (n >> 3) + (b >> 3);
}

template <int BM, int BN, bool A_KC, bool B_KC, int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmP p)
{
    static_assert(BM % 64 == 0 && BN % 64 == 0, "tile must split over 2x2 waves of 32x32 MFMAs");
    constexpr int TM = BM / 64, TN = BN / 64;
    using TA = TileShape<BM, A_KC>;
    using TB = TileShape<BN, B_KC>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As = smem;                       // two A stages, then two B stages
    float* const Bs = smem + 2 * TA::floats;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int tile = xcd_tile_index(blockIdx.x, tiles_m * tiles_n);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int kb = blockIdx.y * p.k_chunk;
    const int ke = min(p.K, kb + p.k_chunk);
    const int nkt = (ke - kb + BK - 1) / BK;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    f32x4 ra[TA::per_thread], rb[TB::per_thread];
    const int ones = (EPI == EPI_WGRAD) ? p.ones_col : -1;
    // the ones column is synthesised, never read: B really has `ones` columns
    const int nB = (ones >= 0) ? ones : p.N;
    if (nkt > 0) {
        tile_load<BM, A_KC>(ra, p.A, p.lda, p.M, m0, kb, ke, p.a_vec, -1);
        tile_load<BN, B_KC>(rb, p.B, p.ldb, nB, n0, kb, ke, p.b_vec, ones);
        tile_store<BM, A_KC>(ra, As);
        tile_store<BN, B_KC>(rb, Bs);
    }
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nkt;
        if (more) {
            tile_load<BM, A_KC>(ra, p.A, p.lda, p.M, m0, kb + (kt + 1) * BK, ke, p.a_vec, -1);
            tile_load<BN, B_KC>(rb, p.B, p.ldb, nB, n0, kb + (kt + 1) * BK, ke, p.b_vec, ones);
        }
        const int kvalid = min(BK, ke - (kb + kt * BK));
        const int ng = (kvalid + 7) >> 3;
        for (int g = 0; g < ng; ++g) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = frag_read<BM, A_KC>(As + cur * TA::floats, wm0 + 32 * i, g, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = frag_read<BN, B_KC>(Bs + cur * TB::floats, wn0 + 32 * j, g, lane);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e],
                                                                        acc[i][j], 0, 0, 0);
        }
        if (more) {
            tile_store<BM, A_KC>(ra, As + (cur ^ 1) * TA::floats);
            tile_store<BN, B_KC>(rb, Bs + (cur ^ 1) * TB::floats);
        }
        __syncthreads();
    }

    // Epilogue. Accumulator register r of lane l holds
    //   row = (r&3) + 8*(r>>2) + 4*(l>>5), col = l&31   of its 32x32 block.
    const int col_l = lane & 31, rsub = 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn0 + 32 * j + col_l;
        if (n >= p.N) continue;
        float bias = 0.0f;
        if constexpr (EPI == EPI_FWD) bias = p.bias ? p.bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + rsub;
                if (m >= p.M) continue;
                float v = acc[i][j][r];
                if constexpr (EPI == EPI_FWD) {
                    p.C[(int64_t)m * p.ldc + n] = act_apply(v + bias, p.act);
                } else if constexpr (EPI == EPI_DGRAD) {
                    if (p.aux) v *= act_grad(p.aux[(int64_t)m * p.ldaux + n], p.act);
                    p.C[(int64_t)m * p.ldc + n] = v;
                } else {
                    const int64_t slab = (int64_t)blockIdx.y * p.slab_stride;
                    if (n == p.ones_col) p.C2[slab + m] = v;
                    else p.C[slab + (int64_t)m * p.ldc + n] = v;
                }
            }
        }
    }
}

template <int BM, int BN, bool A_KC, bool B_KC>
constexpr size_t gemm_lds_bytes()
{
    return sizeof(float) * 2 * (TileShape<BM, A_KC>::floats + TileShape<BN, B_KC>::floats);
}

}  // namespace abn
