// dtw.hip -- batched DTW frame alignment for gfx950.
//
// Replaces, for a whole batch of token pairs at once,
//   abnet3/utils.py:40-60    cosine_distance  (arccos(cos)/pi, float32 math)
//   abnet3/utils.py:147-153  get_dtw_alignment -> third-party dtw.DTW(...)
// whose per-pair Python/Cython loop is the producer-side hot loop of
// abnet3/dataloader.py:166-261 and :617-671.
//
// Three kernels per batch:
//  1. dist_kernel   64x64 tiles of the angular distance matrix, one 32x32 quadrant
//                   per wavefront on the float32 matrix cores: the MFMA
//                   accumulates every cell as one sequential fused chain over k,
//                   exactly the oracle's fmaf loop (= what sgemm computes behind
//                   the reference's np.dot), and the rest of the cell -- norms in
//                   numpy's summation order, one division, glibc's acosf, / pi --
//                   follows the reference's statements operation by operation
//                   (dist_ref.h), so the values are bit-identical to the C oracle
//                   and to the reference's own output (tests/golden/cosdist_libm.npz;
//                   the file is compiled with -ffp-contract=off).  The
//                   matrix is written in the layout the DP reads with 16-byte
//                   coalesced loads:
//                     S4[g][phys(i)][e] = dist(i, 4g + e - i)
//                   i.e. one float4 holds row i's cells on the four
//                   anti-diagonals of group g, and phys(i) = (i % SL)*64 + i/SL
//                   puts the SL rows a DP lane owns 64 float4s apart.
//  2. dp_kernel<SL> one wavefront per pair sweeps the anti-diagonals with the two
//                   previous diagonals of float64 costs in REGISTERS.  Lane l
//                   owns the SL consecutive rows l*SL .. l*SL+SL-1, so row i-1
//                   is the same lane's previous register and only the lane's
//                   first row needs one wave rotation per step.  One float4
//                   load per row serves four diagonals and is issued a whole
//                   group (four steps) ahead; the 2-bit back-pointers of four
//                   diagonals are packed into one byte store.
//                   cost = D + min(diag, up, left), first minimum in that order
//                   wins (the oracle's tie-break).
//  3. the same kernel then walks the back-pointers from (N-1, M-1): the wave
//                   stages a 64-row x 64-diagonal window of back-pointers in
//                   LDS, lane 0 walks it at LDS latency (>= 30 steps per
//                   window), and all lanes finally reverse the path into place.
// The DP is dependency-bound (N+M-1 sequential steps per pair), not HBM-bound:
// parallelism comes from running thousands of pairs side by side, longest first.
#include "common.h"
#include "dist_ref.h"
#include <algorithm>
#include <mutex>
#include <vector>
#include <type_traits>

namespace abn {

struct PairMeta {
    int64_t off1, off2;      // first row of each token in feats1 / feats2
    int32_t n1, n2;
    int64_t ws_off;          // float offset of this pair's S4 region
    int64_t dir_off;         // byte offset of this pair's packed back-pointers
    int32_t tile0;           // first tile id of this pair (dist kernel)
    int32_t tiles_n;         // tiles along j
    int32_t slots;           // SL: rows per DP lane (the pair's size class)
    int32_t groups;          // ceil((n1 + n2 - 1) / 4) groups of four anti-diagonals
};

constexpr int TS = 64;       // distance tile

// tile id -> pair id table, filled on the device from the per-pair tile ranges
__global__ void expand_tiles_kernel(const PairMeta* __restrict__ meta, int npairs, int32_t* __restrict__ tile_pair)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npairs) return;
    const PairMeta m = meta[p];
    const int tm = (m.n1 + TS - 1) / TS, tn = (m.n2 + TS - 1) / TS;
    for (int t = 0; t < tm * tn; ++t) tile_pair[m.tile0 + t] = p;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int TSP = TS + 4;  // padded tile row: float4 rows of the MFMA layout land on distinct LDS banks
constexpr int KC = 40;       // k chunk staged in LDS (the 40-d filterbank frame in one piece)
constexpr int KCP = KC + 4;  // padded: 16-byte row reads of 8 consecutive rows hit 32 distinct banks

// One 64x64 tile of a pair's distance matrix per workgroup, one 32x32 quadrant
// per wavefront.  The dot products run on the matrix cores: a float32 MFMA
// (v_mfma_f32_32x32x2_f32, k = 2s + lane/32) accumulates each cell as ONE
// sequential fused chain over k, bit-identical to the oracle's fmaf loop
// (tools/mfma_exact_probe.hip: 0 mismatches in 204 800 cells), which leaves the
// vector ALU to the acos epilogue -- the part that bounds this kernel.
// Rows are staged through LDS with coalesced 16-byte loads (a token's rows are
// contiguous), and the tile's row norms are recomputed from the staged rows (40
// fmas per row: cheaper than a separate pass over the corpus).  A = rows of y (j), B = rows of x (i): a lane then holds one i and runs of four
// consecutive j, which go to LDS as float4s and leave in DP order.
#ifdef ABN_DIST_STAMPS      // diagnostic build only (tools/dist_stamps.py): cycles per phase, summed over blocks
__device__ unsigned long long g_dist_cycles[8];
#define DSTAMP() do { if (threadIdx.x == 0) tk[nk++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DSTAMP() do {} while (0)
#endif

__global__ __launch_bounds__(256) void dist_kernel(const float* __restrict__ feats1, const float* __restrict__ feats2,
                                                   const PairMeta* __restrict__ meta, const int32_t* __restrict__ tile_pair,
                                                   int ntiles, int D, int vec, float* __restrict__ ws,
                                                   int32_t* __restrict__ bad)
{
    // x / y row chunks [64][KC] for the MFMA loop; the finished tile reuses the space
    __shared__ __attribute__((aligned(16))) float smem[2 * TS * KCP];
    __shared__ float nx_s[TS], ny_s[TS];
    static_assert(2 * TS * KCP >= TS * TSP, "the distance tile must fit in the staging buffers");
    float (*xs)[KCP] = reinterpret_cast<float (*)[KCP]>(smem);
    float (*ys)[KCP] = reinterpret_cast<float (*)[KCP]>(smem + TS * KCP);
    float* tile = smem;
#ifdef ABN_DIST_STAMPS
    unsigned long long tk[8];
    int nk = 0;
#endif
    DSTAMP();
    // XCD-aware order: workgroups b, b+8, b+16, ... share an XCD and its L2, so XCD x takes
    // the contiguous tile range [x*G/8, (x+1)*G/8): the tiles of one pair (consecutive ids)
    // then run on ONE XCD and re-read their x / y rows from its L2 instead of HBM
    // (round-robin order: 4.4 GB fetched for 1 GB of features).
    const int tile_id = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    if (tile_id >= ntiles) return;
    const int p = tile_pair[tile_id];
    const PairMeta m = meta[p];
    const int t = tile_id - m.tile0;
    const int i0 = (t / m.tiles_n) * TS, j0 = (t % m.tiles_n) * TS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int ib = 32 * (wave >> 1), jb = 32 * (wave & 1);        // this wave's quadrant
    // rows past the token end are clamped: their cells are computed and discarded
    const float* xbase = feats1 + m.off1 * D;
    const float* ybase = feats2 + m.off2 * D;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
    for (int k0 = 0; k0 < D; k0 += KC) {
        const int kn = min(KC, D - k0);
        if (vec) {                                // 16-byte loads; a token's rows are contiguous in memory
            for (int u = threadIdx.x; u < 2 * TS * (KC / 4); u += 256) {
                const int which = u >= TS * (KC / 4), v = u - which * TS * (KC / 4);
                const int row = v / (KC / 4), c4 = (v % (KC / 4)) * 4;
                float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c4 < kn) {
                    const float* src = which ? ybase + (int64_t)min(j0 + row, m.n2 - 1) * D : xbase + (int64_t)min(i0 + row, m.n1 - 1) * D;
                    q = *reinterpret_cast<const float4*>(src + k0 + c4);
                }
                *reinterpret_cast<float4*>(which ? &ys[row][c4] : &xs[row][c4]) = q;
            }
        } else {
            for (int u = threadIdx.x; u < 2 * TS * KC; u += 256) {
                const int which = u >= TS * KC, v = u - which * TS * KC;
                const int row = v / KC, c = v % KC;
                float q = 0.0f;
                if (c < kn) q = which ? ybase[(int64_t)min(j0 + row, m.n2 - 1) * D + k0 + c] : xbase[(int64_t)min(i0 + row, m.n1 - 1) * D + k0 + c];
                (which ? ys : xs)[row][c] = q;
            }
        }
        __syncthreads();
        DSTAMP();                                 // (D <= KC: one trip) operands staged
        // the chunk is zero-filled past kn, and fma(0, 0, acc) == acc exactly, so
        // whole float4 groups can be consumed; k still ascends one at a time
        // the tile's 64 + 64 row norms, np.sqrt(np.sum(x ** 2, axis=1)) in numpy's pairwise
        // order: from the staged rows when a frame fits one chunk (the 40-d case), else below
        if (wave < 2 && D <= KC) (wave == 0 ? nx_s : ny_s)[lane] = row_norm_numpy(wave == 0 ? xs[lane] : ys[lane], D);
        for (int k = 0; k < kn; k += 4) {
            const float4 yq = *reinterpret_cast<const float4*>(&ys[jb + r][k]);
            const float4 xq = *reinterpret_cast<const float4*>(&xs[ib + r][k]);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? yq.y : yq.x, h ? xq.y : xq.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? yq.w : yq.z, h ? xq.w : xq.z, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    DSTAMP();                                     // norms + MFMA + barrier
    if (wave < 2 && D > KC)                       // wide frames: the summation order spans the chunks, read the row itself
        (wave == 0 ? nx_s : ny_s)[lane] = wave == 0 ? row_norm_numpy(xbase + (int64_t)min(i0 + lane, m.n1 - 1) * D, D)
                                                      : row_norm_numpy(ybase + (int64_t)min(j0 + lane, m.n2 - 1) * D, D);
    __syncthreads();                              // the norms are staged
    DSTAMP();
    const int i = i0 + ib + r;
    const float nx = nx_s[ib + r];
    // zero rows are rare: tiles without one skip their handling
    const bool zero_rows = __any(nx_s[lane] == 0.0f || ny_s[lane] == 0.0f);
    bool any_bad = false;
    auto epilogue = [&](auto zr) {
#pragma unroll
        for (int qg = 0; qg < 4; ++qg) {
            const int jl = jb + 8 * qg + 4 * h;   // MFMA rows 8*qg + 4*h + (0..3)
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = angular_distance_ref<decltype(zr)::value>(acc[4 * qg + e], nx, ny_s[jl + e]);
                const bool valid = i < m.n1 && j0 + jl + e < m.n2;
                any_bad |= valid && !(d >= 0.0f);     // utils.py:59 assert
                v[e] = valid ? d : 0.0f;
            }
            *reinterpret_cast<float4*>(&tile[(ib + r) * TSP + jl]) = make_float4(v[0], v[1], v[2], v[3]);
        }
    };
    if (zero_rows) epilogue(std::true_type{}); else epilogue(std::false_type{});
    if (any_bad) atomicOr(&bad[p], 1);
    __syncthreads();
    DSTAMP();                                     // acos epilogue
    // write-out in DP order: one float4 = row i on the four diagonals of group g
    // (cells j = 4g - i .. 4g - i + 3); a wave takes one group, lanes take rows
    {
        const int SL = m.slots;
        const int i = i0 + lane;
        const int iend = min(i0 + TS, m.n1) - 1, jend = min(j0 + TS, m.n2) - 1;
        const int64_t gstride = (int64_t)64 * SL;                 // float4s per group
        float4* S4 = reinterpret_cast<float4*>(ws + m.ws_off) + (i % SL) * 64 + i / SL;
        for (int g = ((i0 + j0) >> 2) + wave; g <= ((iend + jend) >> 2); g += 4) {
            const int jl = 4 * g - i - j0;                        // tile column of element 0
            if (i > iend || jl + 3 < 0 || jl > jend - j0) continue;
            float* dst = reinterpret_cast<float*>(S4 + g * gstride);
            const float* src = &tile[lane * TSP + jl];
            if (jl >= 0 && jl + 3 <= jend - j0) {
                *reinterpret_cast<float4*>(dst) = make_float4(src[0], src[1], src[2], src[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (jl + e >= 0 && jl + e <= jend - j0) dst[e] = src[e];
            }
        }
    }
    DSTAMP();                                     // write-out issued
#ifdef ABN_DIST_STAMPS
    if (threadIdx.x == 0) {
        for (int q = 1; q < nk && q < 8; ++q) atomicAdd(&g_dist_cycles[q], tk[q] - tk[q - 1]);
        atomicAdd(&g_dist_cycles[0], 1ull);
    }
#endif
}

constexpr int DP_MAXN = 1024;        // longest first token a wavefront can sweep (16 rows per lane)
constexpr int DP_CLASSES[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16};      // rows per lane the DP is instantiated for
constexpr int DP_NCLASSES = sizeof(DP_CLASSES) / sizeof(int);
constexpr int WIN = 16;              // traceback window: 16 groups (64 diagonals) x 64 rows
constexpr int DP_WAVES = 4;          // pairs (wavefronts) per DP workgroup

// back-pointer codes
enum { DIR_DIAG = 0, DIR_UP = 1, DIR_LEFT = 2 };

static inline int dp_class_of(int n1)
{
    const int need = (std::max(n1, 1) + 63) / 64;
    for (int c = 0; c < DP_NCLASSES; ++c)
        if (DP_CLASSES[c] >= need) return c;
    return DP_NCLASSES - 1;
}

// LDS hand-off inside ONE wavefront: its LDS operations complete in order, so draining
// them (and keeping the compiler from moving accesses across) is all that is needed
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// lane l receives lane l-1's value (lane 0: lane 63's)
__device__ __forceinline__ double rotate_up(double v, int src_lane) { return __shfl(v, src_lane, 64); }

template <int SL>
__global__ __launch_bounds__(64 * DP_WAVES) void dp_kernel(const PairMeta* __restrict__ meta, const int32_t* __restrict__ order,
                                                int npairs, float* __restrict__ ws, uint8_t* __restrict__ dirs,
                                                const int32_t* __restrict__ bad, int32_t* __restrict__ path1,
                                                int32_t* __restrict__ path2, int32_t* __restrict__ path_len,
                                                int64_t path_stride, double* __restrict__ total_cost)
{
    // DP_WAVES independent pairs per workgroup, one per wavefront (so the waves land on
    // different SIMDs); nothing below synchronises across waves
    __shared__ uint8_t win_all[DP_WAVES][WIN][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ((int)blockIdx.x * DP_WAVES + wave >= npairs) return;
    uint8_t (*win)[64] = win_all[wave];
    const int p = order[blockIdx.x * DP_WAVES + wave];
    const PairMeta m = meta[p];
    const int N = m.n1, M = m.n2;
    if (N <= 0 || M <= 0 || bad[p]) {
        if (lane == 0) { path_len[p] = 0; if (total_cost) total_cost[p] = 0.0; }
        return;
    }
    constexpr int64_t GS = 64 * SL;               // rows (float4s / bytes) per group
    const float4* S4 = reinterpret_cast<const float4*>(ws + m.ws_off) + lane;
    uint8_t* Dr = dirs + m.dir_off;
    const int G = m.groups, row0 = lane * SL;
    const double INF = __builtin_inf();
    double p1[SL], p2[SL];                        // rows row0 + c on diagonals d-1, d-2
    float4 nxt[SL];
#pragma unroll
    for (int c = 0; c < SL; ++c) {
        p1[c] = INF; p2[c] = INF;
        nxt[c] = S4[c * 64];                      // rows >= N are allocated (never written, never used)
    }
    bool row_ok[SL];
#pragma unroll
    for (int c = 0; c < SL; ++c) {
        row_ok[c] = row0 + c < N;
        // consume the first group here: otherwise the loop header inherits "loads pending" from
        // this edge and waits for ALL memory operations, the back-pointer stores included, every trip
        asm volatile("" ::"v"(nxt[c].x), "v"(nxt[c].y), "v"(nxt[c].z), "v"(nxt[c].w));
    }
    const int src = (lane + 63) & 63;
    // group g holds diagonals 4g .. 4g+3; diagonal d holds cells (i, d - i)
    for (int g = 0; g < G; ++g) {
        float4 cur[SL];
#pragma unroll
        for (int c = 0; c < SL; ++c) cur[c] = nxt[c];
        {   // prefetch the next group a whole group (four steps) ahead.  Unconditional (the
            // last iteration re-reads its own group): a guard equal to the loop condition
            // lets the compiler sink the loads into the latch, right in front of their use.
            const int gn = min(g + 1, G - 1);
#pragma unroll
            for (int c = 0; c < SL; ++c) nxt[c] = S4[gn * GS + c * 64];
        }
        uint32_t bits[SL];
#pragma unroll
        for (int c = 0; c < SL; ++c) bits[c] = 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int d = 4 * g + e;
            // row row0 - 1: the previous lane's last row; row -1 is +inf except the virtual (-1,-1) = 0
            double rot1 = rotate_up(p1[SL - 1], src), rot2 = rotate_up(p2[SL - 1], src);
            if (lane == 0) { rot1 = INF; rot2 = d == 0 ? 0.0 : INF; }
#pragma unroll
            for (int c = SL - 1; c >= 0; --c) {   // descending: p1[c-1], p2[c-1] still hold the old diagonals
                const double up = c ? p1[c - 1] : rot1;       // (i-1, j)   on diagonal d-1
                const double dg = c ? p2[c - 1] : rot2;       // (i-1, j-1) on diagonal d-2
                const double left = p1[c];                    // (i, j-1)   on diagonal d-1
                // straight-line selects (no divergent branches): first minimum in the order diag, up, left
                const bool take_up = up < dg;
                const double b1 = take_up ? up : dg;
                const bool take_left = left < b1;
                const double best = take_left ? left : b1;
                const uint32_t dir = take_left ? (uint32_t)DIR_LEFT : take_up ? (uint32_t)DIR_UP : (uint32_t)DIR_DIAG;
                const float dist = e == 0 ? cur[c].x : e == 1 ? cur[c].y : e == 2 ? cur[c].z : cur[c].w;
                const double cost = (double)dist + best;
                // cell (row0 + c, d - row0 - c) exists?  rows never reached keep +inf: that is the boundary condition
                const bool on = row_ok[c] && (uint32_t)(d - row0 - c) < (uint32_t)M;
                p2[c] = left;
                p1[c] = on ? cost : left;
                bits[c] |= on ? dir << (2 * e) : 0u;
            }
        }
#pragma unroll
        for (int c = 0; c < SL; ++c) Dr[g * GS + c * 64 + lane] = (uint8_t)bits[c];
    }
    if (total_cost) {                             // p1 of row N-1 still holds cell (N-1, M-1)
        double last = 0.0;
#pragma unroll
        for (int c = 0; c < SL; ++c)
            if (c == (N - 1) % SL) last = p1[c];
        last = __shfl(last, (N - 1) / SL, 64);
        if (lane == 0) total_cost[p] = last;
    }
    // The back-pointers were stored by THIS wavefront and are read back by it: its stores
    // only have to be complete (the vector L1 is write-through, and no line of this region
    // was ever loaded before).  An agent-scope fence would write the XCD's whole L2 back
    // -- per pair -- and made short pairs several times slower.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);

    // traceback.  The distances are dead now: their region takes the reversed path.
    int32_t* tmp = reinterpret_cast<int32_t*>(ws + m.ws_off);
    int i = N - 1, j = M - 1, k = 0;
    if (lane == 0) { tmp[0] = i; tmp[1] = j; }
    while (i > 0 || j > 0) {                      // wave-uniform
        const int gh = (i + j) >> 2, rlo = i - 63;
        const int r = rlo + lane;
        const int phys = r >= 0 ? (r % SL) * 64 + r / SL : 0;
#pragma unroll
        for (int q = 0; q < WIN; ++q) {
            const int gq = gh - q;
            win[q][lane] = (r >= 0 && gq >= 0) ? Dr[gq * GS + phys] : (uint8_t)0;
        }
        wave_lds_sync();
        if (lane == 0) {
            while ((i > 0 || j > 0) && i >= rlo && ((i + j) >> 2) > gh - WIN) {
                const int d = i + j;
                const int dir = (win[gh - (d >> 2)][i - rlo] >> (2 * (d & 3))) & 3;
                if (dir == DIR_DIAG) { --i; --j; } else if (dir == DIR_UP) --i; else --j;
                ++k;
                tmp[2 * k] = i;
                tmp[2 * k + 1] = j;
            }
        }
        i = __shfl(i, 0, 64);
        j = __shfl(j, 0, 64);
        k = __shfl(k, 0, 64);
        wave_lds_sync();
    }
    // lane 0's stores are complete; the loads below bypass the L1 (which may still hold
    // the distances that lived in these lines)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    const int len = k + 1;
    int32_t* o1 = path1 + (int64_t)p * path_stride;
    int32_t* o2 = path2 + (int64_t)p * path_stride;
    for (int t = lane; t < len; t += 64) {
        o1[t] = __hip_atomic_load(&tmp[2 * (len - 1 - t)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        o2[t] = __hip_atomic_load(&tmp[2 * (len - 1 - t) + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) path_len[p] = len;
}

// plain [N, M] float64 distance matrix of one pair (abn_cosine_distance)
__global__ void dist_plain_kernel(const float* __restrict__ x, int N, const float* __restrict__ y, int M, int D,
                                  double* __restrict__ d, int32_t* __restrict__ bad)
{
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * M) return;
    const int i = (int)(idx / M), j = (int)(idx % M);
    const float* a = x + (int64_t)i * D;
    const float* b = y + (int64_t)j * D;
    float dot = 0.0f;
    for (int k = 0; k < D; ++k) dot = fmaf(a[k], b[k], dot);
    const float v = angular_distance_ref(dot, row_norm_numpy(a, D), row_norm_numpy(b, D));
    if (!(v >= 0.0f) && bad) atomicOr(bad, 1);
    d[idx] = (double)v;
}

// float64 inputs: the reference computes in the input precision (utils.py:41-42).  The
// same statements in double; arccos is the device library's (within a few ulp of libm's).
__global__ void dist_plain_f64_kernel(const double* __restrict__ x, int N, const double* __restrict__ y, int M, int D,
                                      double* __restrict__ d, int32_t* __restrict__ bad)
{
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * M) return;
    const int i = (int)(idx / M), j = (int)(idx % M);
    const double* a = x + (int64_t)i * D;
    const double* b = y + (int64_t)j * D;
    double dot = 0.0, sa = 0.0, sb = 0.0;
    for (int k = 0; k < D; ++k) {
        dot = fma(a[k], b[k], dot);
        sa += a[k] * a[k];
        sb += b[k] * b[k];
    }
    const double na = sqrt(sa), nb = sqrt(sb);
    double v = acos(dot / (na * nb)) / 3.14159265358979323846;
    if (na == 0.0 || nb == 0.0) v = 1.0;
    if (na == 0.0 && nb == 0.0) v = 0.0;
    if (!(v >= 0.0) && bad) atomicOr(bad, 1);
    d[idx] = v;
}

// acosf_ref over an array: lets the tests compare the device routine with libm's acosf
// argument by argument (all 2^31 of them, tools/acosf_gpu_exhaustive.py)
__global__ void arccos_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = acosf_ref(x[i]);
}

struct WsPlan {
    int64_t meta_off, tilepair_off, order_off, bad_off, dist_off, dirs_off, total;
    int64_t total_tiles, dist_floats;
};

}  // namespace abn

using namespace abn;

// Side streams for the DP size classes, one set per device, created on first use.
constexpr int N_SIDE = 4;
struct SideStreams {
    std::mutex mu;              // the fork / join events are shared by all callers
    hipStream_t s[N_SIDE];
    hipEvent_t fork, join[N_SIDE];
    bool ok = false;
};
static SideStreams* side_streams()
{
    static SideStreams per_device[16];
    static std::mutex init_mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    SideStreams& ss = per_device[dev];
    std::lock_guard<std::mutex> g(init_mu);
    if (!ss.ok) {
        bool good = hipEventCreateWithFlags(&ss.fork, hipEventDisableTiming) == hipSuccess;
        for (int q = 0; q < N_SIDE && good; ++q)
            good = hipStreamCreateWithFlags(&ss.s[q], hipStreamNonBlocking) == hipSuccess &&
                   hipEventCreateWithFlags(&ss.join[q], hipEventDisableTiming) == hipSuccess;
        if (!good) { (void)hipGetLastError(); return nullptr; }
        ss.ok = true;
    }
    return &ss;
}

// S4 rows of one pair: groups x 64 x SL (float4s for the distances, bytes for the back-pointers)
static inline int64_t pair_rows(int64_t a, int64_t b)
{
    if (a <= 0 || b <= 0) return 0;
    return ((a + b - 1 + 3) / 4) * 64 * DP_CLASSES[dp_class_of((int)a)];
}

// Workspace: [PairMeta x P][tile->pair x tiles][order x P][bad x P][S4 dist f32][dirs u8]
static WsPlan plan_ws(const int32_t* n1, const int32_t* n2, int64_t P, int64_t rows1, int64_t rows2)
{
    WsPlan w;
    int64_t tiles = 0, rows = 0;
    for (int64_t p = 0; p < P; ++p) {
        const int64_t a = n1[p] > 0 ? n1[p] : 0, b = n2[p] > 0 ? n2[p] : 0;
        tiles += ((a + TS - 1) / TS) * ((b + TS - 1) / TS);
        rows += pair_rows(a, b);
    }
    int64_t o = 0;
    auto take = [&](int64_t bytes) { int64_t r = o; o += align_up(bytes, 256); return r; };
    w.meta_off = take(P * (int64_t)sizeof(PairMeta));
    w.tilepair_off = take(tiles * 4);
    w.order_off = take(P * 4);
    w.bad_off = take(P * 4);
    w.dist_off = take(rows * 16);
    w.dirs_off = take(rows);
    w.total = o;
    w.total_tiles = tiles;
    w.dist_floats = rows * 4;
    return w;
}

#ifdef ABN_DIST_STAMPS
extern "C" int abn_debug_dist_cycles(unsigned long long* out8)
{
    return hipMemcpyFromSymbol(out8, HIP_SYMBOL(abn::g_dist_cycles), 64) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int64_t abn_dtw_ws_bytes(const int32_t* n1_host, const int32_t* n2_host, int64_t npairs,
                                         int64_t rows1, int64_t rows2)
{
    if (!n1_host || !n2_host || npairs < 0 || rows1 < 0 || rows2 < 0) return -1;
    return plan_ws(n1_host, n2_host, npairs, rows1, rows2).total;
}

extern "C" int64_t abn_dtw_host_stage_bytes(const int32_t* n1_host, const int32_t* n2_host, int64_t npairs)
{
    if (!n1_host || !n2_host || npairs < 0) return -1;
    const WsPlan w = plan_ws(n1_host, n2_host, npairs, 0, 0);
    (void)w;
    return align_up(npairs * (int64_t)sizeof(PairMeta), 256) + align_up(npairs * 4, 256);
}

extern "C" int abn_dtw_batched(const float* feats1, int64_t rows1, const float* feats2, int64_t rows2,
                                    const int64_t* off1_host, const int32_t* n1_host, const int64_t* off2_host,
                                    const int32_t* n2_host, int64_t npairs, int64_t D, int32_t* path1,
                                    int32_t* path2, int32_t* path_len, int64_t path_stride, double* total_cost,
                                    void* ws, int64_t ws_bytes, void* host_stage, int64_t host_stage_bytes,
                                    void* stream)
{
    ABN_REQUIRE(npairs >= 0 && D >= 1 && D < (1 << 20), "dtw: bad npairs/D");
    if (npairs == 0) return ABN_OK;
    ABN_REQUIRE(feats1 && feats2 && off1_host && n1_host && off2_host && n2_host && path1 && path2 && path_len && ws &&
                    host_stage,
                "dtw: null pointer");
    int32_t maxlen = 0;
    for (int64_t p = 0; p < npairs; ++p) {
        ABN_REQUIRE(n1_host[p] >= 0 && n2_host[p] >= 0, "dtw: negative token length at pair %lld", (long long)p);
        ABN_REQUIRE(n1_host[p] <= DP_MAXN, "dtw: token of %d frames exceeds the %d-frame limit", n1_host[p], DP_MAXN);
        ABN_REQUIRE(off1_host[p] >= 0 && off1_host[p] + n1_host[p] <= rows1 && off2_host[p] >= 0 &&
                        off2_host[p] + n2_host[p] <= rows2,
                    "dtw: pair %lld reads outside the feature arrays", (long long)p);
        const int32_t need = n1_host[p] + n2_host[p] - 1;
        maxlen = need > maxlen ? need : maxlen;
    }
    ABN_REQUIRE(path_stride >= maxlen, "dtw: path_stride %lld < longest possible path %d", (long long)path_stride, maxlen);
    const WsPlan w = plan_ws(n1_host, n2_host, npairs, rows1, rows2);
    if (ws_bytes < w.total) { set_error("dtw: workspace too small (%lld < %lld bytes)", (long long)ws_bytes, (long long)w.total); return ABN_E_WORKSPACE; }
    const int64_t meta_bytes = align_up(npairs * (int64_t)sizeof(PairMeta), 256);
    if (host_stage_bytes < meta_bytes + align_up(npairs * 4, 256)) { set_error("dtw: host staging buffer too small"); return ABN_E_WORKSPACE; }

    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)ws;
    PairMeta* hm = (PairMeta*)host_stage;
    int32_t* hord = (int32_t*)((char*)host_stage + meta_bytes);
    int64_t tiles = 0, rows = 0;
    int64_t class_count[DP_NCLASSES] = {};
    for (int64_t p = 0; p < npairs; ++p) {
        const int64_t a = n1_host[p], b = n2_host[p];
        const int64_t tm = (a + TS - 1) / TS, tn = (b + TS - 1) / TS;
        const int cls = dp_class_of((int)a);
        hm[p].off1 = off1_host[p]; hm[p].off2 = off2_host[p];
        hm[p].n1 = (int32_t)a; hm[p].n2 = (int32_t)b;
        hm[p].ws_off = rows * 4;
        hm[p].dir_off = rows;
        hm[p].tile0 = (int32_t)tiles;
        hm[p].tiles_n = (int32_t)(tn > 0 ? tn : 1);
        hm[p].slots = DP_CLASSES[cls];
        hm[p].groups = (int32_t)((a + b - 1 + 3) / 4);
        tiles += tm * tn;
        rows += pair_rows(a, b);
        ++class_count[cls];
    }
    // DP launch order: by size class, longest sweep first inside a class (the short ones fill
    // the tail).  One integer key per pair: class | inverted length | index.
    {
        std::vector<uint64_t> keys((size_t)npairs);
        for (int64_t p = 0; p < npairs; ++p) {
            const uint64_t len = (uint64_t)n1_host[p] + (uint64_t)n2_host[p];
            keys[p] = ((uint64_t)dp_class_of(n1_host[p]) << 58) | ((((uint64_t)1 << 26) - 1 - std::min<uint64_t>(len, (1u << 26) - 1)) << 32) |
                      (uint64_t)p;
        }
        std::sort(keys.begin(), keys.end());
        for (int64_t p = 0; p < npairs; ++p) hord[p] = (int32_t)(keys[p] & 0xffffffffu);
    }
    if (hipMemcpyAsync(base + w.meta_off, hm, npairs * sizeof(PairMeta), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(base + w.order_off, hord, npairs * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemsetAsync(base + w.bad_off, 0, npairs * 4, st) != hipSuccess) {
        set_error("dtw: metadata upload failed");
        return ABN_E_LAUNCH;
    }
    if (tiles > 0) {
        hipLaunchKernelGGL(expand_tiles_kernel, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, st,
                           (const PairMeta*)(base + w.meta_off), (int)npairs, (int32_t*)(base + w.tilepair_off));
        hipLaunchKernelGGL(dist_kernel, dim3((unsigned)align_up(tiles, 8)), dim3(256), 0, st, feats1, feats2,
                           (const PairMeta*)(base + w.meta_off), (const int32_t*)(base + w.tilepair_off), (int)tiles, (int)D,
                           (int)(D % 4 == 0 && aligned16(feats1) && aligned16(feats2)), (float*)(base + w.dist_off),
                           (int32_t*)(base + w.bad_off));
    }
    const PairMeta* dm = (const PairMeta*)(base + w.meta_off);
    const int32_t* dord = (const int32_t*)(base + w.order_off);
    float* dd = (float*)(base + w.dist_off);
    uint8_t* dr = (uint8_t*)(base + w.dirs_off);
    const int32_t* db = (const int32_t*)(base + w.bad_off);
    // The size classes are independent: fork them over side streams so that the
    // short tail of one class overlaps the others (joined back into `st` below).
    int64_t first[DP_NCLASSES], acc_first = 0;
    int by_work[DP_NCLASSES], nclasses = 0;
    for (int c = 0; c < DP_NCLASSES; ++c) {
        first[c] = acc_first;
        acc_first += class_count[c];
        if (class_count[c] > 0) by_work[nclasses++] = c;
    }
    // A class with few pairs is latency-bound (one wavefront sweeps a pair; its time grows
    // with rows-per-lane x steps), so the widest classes start first, each on its own
    // stream; the narrow, fast ones queue up behind them.
    std::sort(by_work, by_work + nclasses, [&](int x, int y) { return x > y; });
    SideStreams* side = nclasses > 1 ? side_streams() : nullptr;
    std::unique_lock<std::mutex> guard;
    if (side) {
        guard = std::unique_lock<std::mutex>(side->mu);
        if (hipEventRecord(side->fork, st) != hipSuccess) side = nullptr;
    }
    bool used[N_SIDE] = {};
    for (int k = 0; k < nclasses; ++k) {
        const int c = by_work[k];
        hipStream_t cs = st;
        if (side && k % (N_SIDE + 1) != 0) {
            const int q = k % (N_SIDE + 1) - 1;
            if (!used[q] && hipStreamWaitEvent(side->s[q], side->fork, 0) != hipSuccess) {
                set_error("dtw: side stream fork failed");
                return ABN_E_LAUNCH;
            }
            used[q] = true;
            cs = side->s[q];
        }
        const dim3 grid((unsigned)((class_count[c] + DP_WAVES - 1) / DP_WAVES));
        const int32_t* ord = dord + first[c];
#define ABN_DP_CASE(C)                                                                                                   \
    case C:                                                                                                              \
        hipLaunchKernelGGL((dp_kernel<DP_CLASSES[C]>), grid, dim3(64 * DP_WAVES), 0, cs, dm, ord, (int)class_count[c], dd, dr, \
                           db, path1, path2, path_len, path_stride, total_cost);                                         \
        break;
        switch (c) {
            ABN_DP_CASE(0) ABN_DP_CASE(1) ABN_DP_CASE(2) ABN_DP_CASE(3) ABN_DP_CASE(4)
            ABN_DP_CASE(5) ABN_DP_CASE(6) ABN_DP_CASE(7) ABN_DP_CASE(8) ABN_DP_CASE(9)
        }
#undef ABN_DP_CASE
    }
    static_assert(DP_NCLASSES == 10, "one ABN_DP_CASE per size class");
    for (int q = 0; q < N_SIDE; ++q)
        if (used[q] && (hipEventRecord(side->join[q], side->s[q]) != hipSuccess ||
                        hipStreamWaitEvent(st, side->join[q], 0) != hipSuccess)) {
            set_error("dtw: side stream join failed");
            return ABN_E_LAUNCH;
        }
    ABN_CHECK_LAUNCH("dtw");
    return ABN_OK;
}

extern "C" int abn_cosine_distance(const float* x, int64_t N, const float* y, int64_t M, int64_t D, double* d,
                                   int32_t* bad_flag, void* stream)
{
    ABN_REQUIRE(N >= 0 && M >= 0 && D >= 1 && N * M < (1LL << 40), "cosine_distance: bad shape");
    if (N * M == 0) return ABN_OK;
    ABN_REQUIRE(x && y && d, "cosine_distance: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (bad_flag && hipMemsetAsync(bad_flag, 0, 4, st) != hipSuccess) { set_error("cosine_distance: memset failed"); return ABN_E_LAUNCH; }
    hipLaunchKernelGGL(dist_plain_kernel, dim3((unsigned)((N * M + 255) / 256)), dim3(256), 0, st, x, (int)N, y, (int)M,
                       (int)D, d, bad_flag);
    ABN_CHECK_LAUNCH("cosine_distance");
    return ABN_OK;
}

extern "C" int abn_cosine_distance_f64(const double* x, int64_t N, const double* y, int64_t M, int64_t D, double* d,
                                       int32_t* bad_flag, void* stream)
{
    ABN_REQUIRE(N >= 0 && M >= 0 && D >= 1 && N * M < (1LL << 40), "cosine_distance_f64: bad shape");
    if (N * M == 0) return ABN_OK;
    ABN_REQUIRE(x && y && d, "cosine_distance_f64: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (bad_flag && hipMemsetAsync(bad_flag, 0, 4, st) != hipSuccess) { set_error("cosine_distance_f64: memset failed"); return ABN_E_LAUNCH; }
    hipLaunchKernelGGL(dist_plain_f64_kernel, dim3((unsigned)((N * M + 255) / 256)), dim3(256), 0, st, x, (int)N, y,
                       (int)M, (int)D, d, bad_flag);
    ABN_CHECK_LAUNCH("cosine_distance_f64");
    return ABN_OK;
}

extern "C" int abn_arccos_f32(const float* x, int64_t n, float* out, void* stream)
{
    ABN_REQUIRE(n >= 0, "arccos_f32: negative length");
    if (n == 0) return ABN_OK;
    ABN_REQUIRE(x && out, "arccos_f32: null pointer");
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(arccos_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, (hipStream_t)stream, x, n, out);
    ABN_CHECK_LAUNCH("arccos_f32");
    return ABN_OK;
}
