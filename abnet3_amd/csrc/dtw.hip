// dtw.hip -- batched DTW frame alignment for gfx950.
//
// Replaces, for a whole batch of token pairs at once,
//   abnet3/utils.py:40-60    cosine_distance  (arccos(cos)/pi, float32 math)
//   abnet3/utils.py:147-153  get_dtw_alignment -> third-party dtw.DTW(...)
// whose per-pair Python/Cython loop is the producer-side hot loop of
// abnet3/dataloader.py:166-261 and :617-671.
//
// Three kernels per batch:
//  1. dist_kernel   64x64 tiles of the angular distance matrix, x / y rows
//                   staged through LDS in 32-wide k chunks, 4x4 cells per
//                   thread.  Every cell's dot product is ONE sequential fmaf
//                   chain over k (and acos is an explicit float32 routine), so
//                   the values are bit-identical to the C oracle; the file is
//                   compiled with -ffp-contract=off.  The matrix is written in
//                   the layout the DP reads with 16-byte coalesced loads:
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
#include <algorithm>

namespace abn {

// ---- float32 acos, operation for operation the oracle's (oracle/dtw.c):
// division-free Abramowitz & Stegun 4.4.46, Horner with explicit fmaf ---------
__device__ __forceinline__ float acos_f32(float x)
{
    const float pi_f = 3.14159274101257324f;
    const float ax = fabsf(x);
    if (!(ax <= 1.0f)) return __builtin_nanf("");
    float p = -0.0012624911f;
    p = fmaf(p, ax, 0.0066700901f);
    p = fmaf(p, ax, -0.0170881256f);
    p = fmaf(p, ax, 0.0308918810f);
    p = fmaf(p, ax, -0.0501743046f);
    p = fmaf(p, ax, 0.0889789874f);
    p = fmaf(p, ax, -0.2145988016f);
    p = fmaf(p, ax, 1.5707963050f);
    const float r = sqrtf(1.0f - ax) * p;
    return x < 0.0f ? pi_f - r : r;
}

// inx / iny are 1/|x|, 1/|y| (+inf for a zero row)
__device__ __forceinline__ float angular_distance(float dot, float inx, float iny)
{
    const float inv_pi_f = 0.318309873342514038f;
    const bool zx = __builtin_isinf(inx), zy = __builtin_isinf(iny);
    if (zx && zy) return 0.0f;                        // utils.py:57-58
    if (zx || zy) return 1.0f;                        // utils.py:55-56
    return acos_f32((dot * inx) * iny) * inv_pi_f;
}

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
constexpr int KC = 32;       // k chunk staged in LDS

// reciprocal row norms 1/sqrt(sum x^2): sequential fmaf chain over k (the
// oracle's order), one thread per row; 16-byte loads when D % 4 == 0
__global__ void norm_kernel(const float* __restrict__ f, int64_t rows, int D, int vec, float* __restrict__ out)
{
    const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* v = f + r * D;
    float s = 0.0f;
    if (vec) {
        for (int k = 0; k < D; k += 4) {
            const float4 q = *reinterpret_cast<const float4*>(v + k);
            s = fmaf(q.x, q.x, s); s = fmaf(q.y, q.y, s); s = fmaf(q.z, q.z, s); s = fmaf(q.w, q.w, s);
        }
    } else {
        for (int k = 0; k < D; ++k) s = fmaf(v[k], v[k], s);
    }
    out[r] = 1.0f / sqrtf(s);
}

__global__ __launch_bounds__(256) void dist_kernel(const float* __restrict__ feats1, const float* __restrict__ feats2,
                                                   const float* __restrict__ norm1, const float* __restrict__ norm2,
                                                   const PairMeta* __restrict__ meta, const int32_t* __restrict__ tile_pair,
                                                   int D, float* __restrict__ ws, int32_t* __restrict__ bad)
{
    __shared__ __attribute__((aligned(16))) float xs[TS][KC + 4], ys[TS][KC + 4];   // rows 16-byte aligned
    __shared__ float tile[TS * TS];
    const int p = tile_pair[blockIdx.x];
    const PairMeta m = meta[p];
    const int t = blockIdx.x - m.tile0;
    const int i0 = (t / m.tiles_n) * TS, j0 = (t % m.tiles_n) * TS;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;      // 16 x 16 threads, 4x4 cells each
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0f;

    for (int k0 = 0; k0 < D; k0 += KC) {
        const int kn = min(KC, D - k0);
        for (int u = threadIdx.x; u < TS * KC; u += 256) {
            const int r = u / KC, k = u % KC;
            float xv = 0.0f, yv = 0.0f;
            if (k < kn) {
                if (i0 + r < m.n1) xv = feats1[(m.off1 + i0 + r) * D + k0 + k];
                if (j0 + r < m.n2) yv = feats2[(m.off2 + j0 + r) * D + k0 + k];
            }
            xs[r][k] = xv;
            ys[r][k] = yv;
        }
        __syncthreads();
        // the chunk is zero-filled past kn, and fma(0, 0, acc) == acc exactly, so
        // whole float4 groups can be consumed; k still ascends one at a time
        for (int k = 0; k < kn; k += 4) {
            float4 xa[4], yb[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) xa[a] = *reinterpret_cast<const float4*>(&xs[ty + 16 * a][k]);
#pragma unroll
            for (int b = 0; b < 4; ++b) yb[b] = *reinterpret_cast<const float4*>(&ys[tx + 16 * b][k]);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    float c = acc[a][b];
                    c = fmaf(xa[a].x, yb[b].x, c);
                    c = fmaf(xa[a].y, yb[b].y, c);
                    c = fmaf(xa[a].z, yb[b].z, c);
                    c = fmaf(xa[a].w, yb[b].w, c);
                    acc[a][b] = c;
                }
        }
        __syncthreads();
    }
    bool any_bad = false;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int i = i0 + ty + 16 * a;
        const float nx = i < m.n1 ? norm1[m.off1 + i] : 1.0f;          // reciprocal norms
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int j = j0 + tx + 16 * b;
            float v = 0.0f;
            if (i < m.n1 && j < m.n2) {
                v = angular_distance(acc[a][b], nx, norm2[m.off2 + j]);
                if (!(v >= 0.0f)) any_bad = true;      // utils.py:59 assert
            }
            tile[(ty + 16 * a) * TS + tx + 16 * b] = v;
        }
    }
    if (any_bad) atomicOr(&bad[p], 1);
    __syncthreads();
    // write-out in DP order: one float4 = row i on the four diagonals of group g
    // (cells j = 4g - i .. 4g - i + 3); a wave takes one group, lanes take rows
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int SL = m.slots;
    const int i = i0 + lane;
    const int iend = min(i0 + TS, m.n1) - 1, jend = min(j0 + TS, m.n2) - 1;
    const int64_t gstride = (int64_t)64 * SL;                     // float4s per group
    float4* S4 = reinterpret_cast<float4*>(ws + m.ws_off) + (i % SL) * 64 + i / SL;
    for (int g = ((i0 + j0) >> 2) + wave; g <= ((iend + jend) >> 2); g += 4) {
        const int jl = 4 * g - i - j0;                            // tile column of element 0
        if (i > iend || jl + 3 < 0 || jl > jend - j0) continue;
        float* dst = reinterpret_cast<float*>(S4 + g * gstride);
        const float* src = &tile[lane * TS + jl];
        if (jl >= 0 && jl + 3 <= jend - j0) {
            *reinterpret_cast<float4*>(dst) = make_float4(src[0], src[1], src[2], src[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (jl + e >= 0 && jl + e <= jend - j0) dst[e] = src[e];
        }
    }
}

constexpr int DP_MAXN = 1024;        // longest first token a wavefront can sweep (16 rows per lane)
constexpr int DP_CLASSES[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16};      // rows per lane the DP is instantiated for
constexpr int DP_NCLASSES = sizeof(DP_CLASSES) / sizeof(int);
constexpr int WIN = 16;              // traceback window: 16 groups (64 diagonals) x 64 rows

// back-pointer codes
enum { DIR_DIAG = 0, DIR_UP = 1, DIR_LEFT = 2 };

static inline int dp_class_of(int n1)
{
    const int need = (std::max(n1, 1) + 63) / 64;
    for (int c = 0; c < DP_NCLASSES; ++c)
        if (DP_CLASSES[c] >= need) return c;
    return DP_NCLASSES - 1;
}

// lane l receives lane l-1's value (lane 0: lane 63's)
__device__ __forceinline__ double rotate_up(double v, int src_lane) { return __shfl(v, src_lane, 64); }

template <int SL>
__global__ __launch_bounds__(64) void dp_kernel(const PairMeta* __restrict__ meta, const int32_t* __restrict__ order,
                                                float* __restrict__ ws, uint8_t* __restrict__ dirs,
                                                const int32_t* __restrict__ bad, int32_t* __restrict__ path1,
                                                int32_t* __restrict__ path2, int32_t* __restrict__ path_len,
                                                int64_t path_stride, double* __restrict__ total_cost)
{
    __shared__ uint8_t win[WIN][64];
    const int p = order[blockIdx.x];
    const PairMeta m = meta[p];
    const int N = m.n1, M = m.n2, lane = threadIdx.x;
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
        nxt[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + c < N) nxt[c] = S4[c * 64];
    }
    const int src = (lane + 63) & 63;
    // group g holds diagonals 4g .. 4g+3; diagonal d holds cells (i, d - i)
    for (int g = 0; g < G; ++g) {
        float4 cur[SL];
#pragma unroll
        for (int c = 0; c < SL; ++c) cur[c] = nxt[c];
        if (g + 1 < G) {
#pragma unroll
            for (int c = 0; c < SL; ++c)
                if (row0 + c < N) nxt[c] = S4[(g + 1) * GS + c * 64];
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
                double best = dg;
                uint32_t dir = DIR_DIAG;
                if (up < best) { best = up; dir = DIR_UP; }
                if (left < best) { best = left; dir = DIR_LEFT; }
                const float dist = e == 0 ? cur[c].x : e == 1 ? cur[c].y : e == 2 ? cur[c].z : cur[c].w;
                const double cost = (double)dist + best;
                const int i = row0 + c, j = d - i;
                p2[c] = p1[c];
                if (i < N && j >= 0 && j < M) {   // rows never reached keep +inf: that is the boundary condition
                    p1[c] = cost;
                    bits[c] |= dir << (2 * e);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < SL; ++c)
            if (row0 + c < N) Dr[g * GS + c * 64 + lane] = (uint8_t)bits[c];
    }
    if (total_cost) {                             // p1 of row N-1 still holds cell (N-1, M-1)
        double last = 0.0;
#pragma unroll
        for (int c = 0; c < SL; ++c)
            if (c == (N - 1) % SL) last = p1[c];
        last = __shfl(last, (N - 1) / SL, 64);
        if (lane == 0) total_cost[p] = last;
    }
    __threadfence();                              // back-pointers are in L2 before anyone reads them

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
        __syncthreads();
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
        __syncthreads();
    }
    __threadfence();                              // lane 0's stores are visible to the whole wave
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
    float dot = 0.0f, sa = 0.0f, sb = 0.0f;
    for (int k = 0; k < D; ++k) {
        dot = fmaf(a[k], b[k], dot);
        sa = fmaf(a[k], a[k], sa);
        sb = fmaf(b[k], b[k], sb);
    }
    const float v = angular_distance(dot, 1.0f / sqrtf(sa), 1.0f / sqrtf(sb));
    if (!(v >= 0.0f) && bad) atomicOr(bad, 1);
    d[idx] = (double)v;
}

struct WsPlan {
    int64_t meta_off, tilepair_off, order_off, norm1_off, norm2_off, bad_off, dist_off, dirs_off, total;
    int64_t total_tiles, dist_floats;
};

}  // namespace abn

using namespace abn;

// S4 rows of one pair: groups x 64 x SL (float4s for the distances, bytes for the back-pointers)
static inline int64_t pair_rows(int64_t a, int64_t b)
{
    if (a <= 0 || b <= 0) return 0;
    return ((a + b - 1 + 3) / 4) * 64 * DP_CLASSES[dp_class_of((int)a)];
}

// Workspace: [PairMeta x P][tile->pair x tiles][order x P][norms][bad x P][S4 dist f32][dirs u8]
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
    w.norm1_off = take(rows1 * 4);
    w.norm2_off = take(rows2 * 4);
    w.bad_off = take(P * 4);
    w.dist_off = take(rows * 16);
    w.dirs_off = take(rows);
    w.total = o;
    w.total_tiles = tiles;
    w.dist_floats = rows * 4;
    return w;
}

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
    return align_up(npairs * (int64_t)sizeof(PairMeta), 256) + align_up(w.total_tiles * 4, 256) + align_up(npairs * 4, 256);
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
    const int64_t tp_bytes = align_up(w.total_tiles * 4, 256);
    if (host_stage_bytes < meta_bytes + tp_bytes + align_up(npairs * 4, 256)) { set_error("dtw: host staging buffer too small"); return ABN_E_WORKSPACE; }

    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)ws;
    PairMeta* hm = (PairMeta*)host_stage;
    int32_t* htp = (int32_t*)((char*)host_stage + meta_bytes);
    int32_t* hord = (int32_t*)((char*)host_stage + meta_bytes + tp_bytes);
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
        for (int64_t t = 0; t < tm * tn; ++t) htp[tiles + t] = (int32_t)p;
        tiles += tm * tn;
        rows += pair_rows(a, b);
        ++class_count[cls];
        hord[p] = (int32_t)p;
    }
    // DP launch order: by size class, longest sweep first inside a class (the short ones fill the tail)
    std::sort(hord, hord + npairs, [&](int32_t x, int32_t y) {
        const int cx = dp_class_of(n1_host[x]), cy = dp_class_of(n1_host[y]);
        if (cx != cy) return cx < cy;
        const int64_t lx = (int64_t)n1_host[x] + n2_host[x], ly = (int64_t)n1_host[y] + n2_host[y];
        return lx != ly ? lx > ly : x < y;
    });
    if (hipMemcpyAsync(base + w.meta_off, hm, npairs * sizeof(PairMeta), hipMemcpyHostToDevice, st) != hipSuccess ||
        (tiles > 0 && hipMemcpyAsync(base + w.tilepair_off, htp, tiles * 4, hipMemcpyHostToDevice, st) != hipSuccess) ||
        hipMemcpyAsync(base + w.order_off, hord, npairs * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemsetAsync(base + w.bad_off, 0, npairs * 4, st) != hipSuccess) {
        set_error("dtw: metadata upload failed");
        return ABN_E_LAUNCH;
    }
    float* norm1 = (float*)(base + w.norm1_off);
    float* norm2 = (float*)(base + w.norm2_off);
    if (rows1 > 0) hipLaunchKernelGGL(norm_kernel, dim3((unsigned)((rows1 + 255) / 256)), dim3(256), 0, st, feats1, rows1, (int)D, (int)(D % 4 == 0 && aligned16(feats1)), norm1);
    if (rows2 > 0) hipLaunchKernelGGL(norm_kernel, dim3((unsigned)((rows2 + 255) / 256)), dim3(256), 0, st, feats2, rows2, (int)D, (int)(D % 4 == 0 && aligned16(feats2)), norm2);
    if (tiles > 0)
        hipLaunchKernelGGL(dist_kernel, dim3((unsigned)tiles), dim3(256), 0, st, feats1, feats2, norm1, norm2,
                           (const PairMeta*)(base + w.meta_off), (const int32_t*)(base + w.tilepair_off), (int)D,
                           (float*)(base + w.dist_off), (int32_t*)(base + w.bad_off));
    const PairMeta* dm = (const PairMeta*)(base + w.meta_off);
    const int32_t* dord = (const int32_t*)(base + w.order_off);
    float* dd = (float*)(base + w.dist_off);
    uint8_t* dr = (uint8_t*)(base + w.dirs_off);
    const int32_t* db = (const int32_t*)(base + w.bad_off);
    int64_t first = 0;
#define ABN_DP_CLASS(C)                                                                                                 \
    if (class_count[C] > 0) {                                                                                           \
        hipLaunchKernelGGL((dp_kernel<DP_CLASSES[C]>), dim3((unsigned)class_count[C]), dim3(64), 0, st, dm, dord + first, \
                           dd, dr, db, path1, path2, path_len, path_stride, total_cost);                                \
        first += class_count[C];                                                                                        \
    }
    ABN_DP_CLASS(0) ABN_DP_CLASS(1) ABN_DP_CLASS(2) ABN_DP_CLASS(3) ABN_DP_CLASS(4)
    ABN_DP_CLASS(5) ABN_DP_CLASS(6) ABN_DP_CLASS(7) ABN_DP_CLASS(8) ABN_DP_CLASS(9)
#undef ABN_DP_CLASS
    static_assert(DP_NCLASSES == 10, "one ABN_DP_CLASS line per size class");
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
