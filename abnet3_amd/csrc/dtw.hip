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
//                   a SKEWED layout  S[(i+j)*N + i]  so that an anti-diagonal
//                   is contiguous in memory.
//  2. dp_kernel     one wavefront per pair sweeps the anti-diagonals: three
//                   rotating diagonals of float64 costs live in LDS, lane l
//                   owns rows l, l+64, ...; reads of the distance diagonal and
//                   writes of the 2-bit back-pointers are coalesced thanks to
//                   the skew.  cost = D + min(diag, up, left), first minimum in
//                   that order wins (the oracle's tie-break).
//  3. the same kernel's lane 0 walks the back-pointers from (N-1, M-1) and
//                   writes the path reversed into place.
// The DP is dependency-bound (N+M-1 sequential steps per pair), not HBM-bound:
// parallelism comes from running thousands of pairs side by side.
#include "common.h"

namespace abn {

// ---- float32 acos, operation for operation the oracle's (oracle/dtw.c) ------
__device__ __forceinline__ float acos_r(float z)
{
    const float pS0 = 1.6666586697e-01f, pS1 = -4.2743422091e-02f, pS2 = -8.6563630030e-03f,
                qS1 = -7.0662963390e-01f;
    const float p = z * (pS0 + z * (pS1 + z * pS2));
    const float q = 1.0f + z * qS1;
    return p / q;
}

__device__ __forceinline__ float acos_f32(float x)
{
    const float pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f;
    const float ax = fabsf(x);
    if (!(ax < 1.0f)) {
        if (x == 1.0f) return 0.0f;
        if (x == -1.0f) return 2.0f * pio2_hi + 0x1p-120f;
        return __builtin_nanf("");
    }
    if (ax < 0.5f) {
        if (ax <= 0x1p-26f) return pio2_hi + 0x1p-120f;
        return pio2_hi - (x - (pio2_lo - x * acos_r(x * x)));
    }
    if (x < 0.0f) {
        const float z = (1.0f + x) * 0.5f;
        const float s = sqrtf(z);
        const float w = acos_r(z) * s - pio2_lo;
        return 2.0f * (pio2_hi - (s + w));
    }
    const float z = (1.0f - x) * 0.5f;
    const float s = sqrtf(z);
    const float df = __uint_as_float(__float_as_uint(s) & 0xfffff000u);
    const float c = (z - df * df) / (s + df);
    const float w = acos_r(z) * s + c;
    return 2.0f * (df + w);
}

__device__ __forceinline__ float angular_distance(float dot, float nx, float ny)
{
    const float pi_f = 3.14159274101257324f;
    if (nx == 0.0f && ny == 0.0f) return 0.0f;        // utils.py:57-58
    if (nx == 0.0f || ny == 0.0f) return 1.0f;        // utils.py:55-56
    return acos_f32(dot / (nx * ny)) / pi_f;
}

struct PairMeta {
    int64_t off1, off2;      // first row of each token in feats1 / feats2
    int32_t n1, n2;
    int64_t ws_off;          // float offset of this pair's skewed matrix
    int32_t tile0;           // first tile id of this pair (dist kernel)
    int32_t tiles_n;         // tiles along j
};

constexpr int TS = 64;       // distance tile
constexpr int KC = 32;       // k chunk staged in LDS

// row norms: sequential fmaf chain over k (the oracle's order), one thread per
// row; 16-byte loads when rows are 16-byte aligned (D % 4 == 0)
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
    out[r] = sqrtf(s);
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
        const float nx = i < m.n1 ? norm1[m.off1 + i] : 1.0f;
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
    // skewed write-out: anti-diagonal dd of the tile is contiguous in S
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* S = ws + m.ws_off;
    for (int dd = wave; dd < 2 * TS - 1; dd += 4) {
        const int jl = dd - lane;
        if (jl >= 0 && jl < TS) {
            const int i = i0 + lane, j = j0 + jl;
            if (i < m.n1 && j < m.n2) S[(int64_t)(i + j) * m.n1 + i] = tile[lane * TS + jl];
        }
    }
}

constexpr int DP_MAXN = 1024;        // rows a wavefront can sweep (LDS: 3 x 8 KB)

// back-pointer codes
enum { DIR_DIAG = 0, DIR_UP = 1, DIR_LEFT = 2 };

constexpr int DP_SLOTS = DP_MAXN / 64;      // rows a lane can own

__global__ __launch_bounds__(64) void dp_kernel(const PairMeta* __restrict__ meta, const float* __restrict__ ws,
                                                uint8_t* __restrict__ dirs, const int32_t* __restrict__ bad,
                                                int32_t* __restrict__ path1, int32_t* __restrict__ path2,
                                                int32_t* __restrict__ path_len, int64_t path_stride,
                                                double* __restrict__ total_cost, int lds_rows)
{
    // Three rotating anti-diagonals of float64 costs; slot i+1 holds row i.
    // Interior cells only ever read predecessors that are real cells of the
    // previous two diagonals, so no sentinel is needed.  Sized by the longest
    // token of the batch (dynamic LDS) so that more pairs fit on a CU.
    extern __shared__ double diag_mem[];
    double* const diag0 = diag_mem;
    const int p = blockIdx.x;
    const PairMeta m = meta[p];
    const int N = m.n1, M = m.n2, lane = threadIdx.x;
    if (N <= 0 || M <= 0 || bad[p]) {
        if (lane == 0) { path_len[p] = 0; if (total_cost) total_cost[p] = 0.0; }
        return;
    }
    const float* S = ws + m.ws_off;
    uint8_t* Dr = dirs + m.ws_off;
    const int nc = (N + 63) / 64;                 // slots this pair uses (<= DP_SLOTS)
    // distances of the NEXT diagonal are fetched while the current one is being
    // relaxed: the DP is a chain of N+M-1 dependent steps and an un-prefetched
    // global load per step would put its full latency on that chain
    float nxt[DP_SLOTS];
    {
        // diagonal 0 is the single cell (0,0)
#pragma unroll
        for (int c = 0; c < DP_SLOTS; ++c) nxt[c] = 0.0f;
        if (lane == 0) nxt[0] = S[0];
    }
    // diagonal d holds cells (i, d - i), max(0, d-M+1) <= i <= min(d, N-1)
    for (int d = 0; d < N + M - 1; ++d) {
        double* cur = diag0 + (d % 3) * lds_rows;
        const double* p1 = diag0 + ((d + 2) % 3) * lds_rows;     // diagonal d-1
        const double* p2 = diag0 + ((d + 1) % 3) * lds_rows;     // diagonal d-2
        const int ilo = max(0, d - M + 1), ihi = min(d, N - 1);
        float dist[DP_SLOTS];
#pragma unroll
        for (int c = 0; c < DP_SLOTS; ++c) dist[c] = nxt[c];
        if (d + 1 < N + M - 1) {
            const int nlo = max(0, d + 1 - M + 1), nhi = min(d + 1, N - 1);
#pragma unroll
            for (int c = 0; c < DP_SLOTS; ++c) {
                const int i = nlo + lane + 64 * c;
                if (c < nc && i <= nhi) nxt[c] = S[(int64_t)(d + 1) * N + i];
            }
        }
#pragma unroll
        for (int c = 0; c < DP_SLOTS; ++c) {
            const int i = ilo + lane + 64 * c;
            if (c < nc && i <= ihi) {
                const int j = d - i;
                double best;
                int dir;
                if (i == 0 && j == 0) { best = 0.0; dir = DIR_DIAG; }
                else if (i == 0) { best = p1[i + 1]; dir = DIR_LEFT; }
                else if (j == 0) { best = p1[i]; dir = DIR_UP; }
                else {
                    best = p2[i]; dir = DIR_DIAG;                       // (i-1, j-1)
                    const double up = p1[i], left = p1[i + 1];          // (i-1, j), (i, j-1)
                    if (up < best) { best = up; dir = DIR_UP; }
                    if (left < best) { best = left; dir = DIR_LEFT; }
                }
                cur[i + 1] = (double)dist[c] + best;
                Dr[(int64_t)d * N + i] = (uint8_t)dir;
            }
        }
        __syncthreads();
    }
    if (lane == 0) {
        if (total_cost) total_cost[p] = diag0[((N + M - 2) % 3) * lds_rows + N];
        // traceback: first pass counts the steps, second writes the path in place
        int i = N - 1, j = M - 1, len = 1;
        while (i > 0 || j > 0) {
            const int dir = Dr[(int64_t)(i + j) * N + i];
            if (dir == DIR_DIAG) { --i; --j; } else if (dir == DIR_UP) --i; else --j;
            ++len;
        }
        int32_t* o1 = path1 + (int64_t)p * path_stride;
        int32_t* o2 = path2 + (int64_t)p * path_stride;
        i = N - 1; j = M - 1;
        int pos = len - 1;
        o1[pos] = i; o2[pos] = j;
        while (i > 0 || j > 0) {
            const int dir = Dr[(int64_t)(i + j) * N + i];
            if (dir == DIR_DIAG) { --i; --j; } else if (dir == DIR_UP) --i; else --j;
            --pos;
            o1[pos] = i; o2[pos] = j;
        }
        path_len[p] = len;
    }
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
    const float v = angular_distance(dot, sqrtf(sa), sqrtf(sb));
    if (!(v >= 0.0f) && bad) atomicOr(bad, 1);
    d[idx] = (double)v;
}

struct WsPlan {
    int64_t meta_off, tilepair_off, norm1_off, norm2_off, bad_off, dist_off, dirs_off, total;
    int64_t total_tiles, dist_floats;
};

}  // namespace abn

using namespace abn;

// Workspace: [PairMeta x P][tile->pair x tiles][norms][bad x P][skewed dist f32][dirs u8]
static WsPlan plan_ws(const int32_t* n1, const int32_t* n2, int64_t P, int64_t rows1, int64_t rows2)
{
    WsPlan w;
    int64_t tiles = 0, cells = 0;
    for (int64_t p = 0; p < P; ++p) {
        const int64_t a = n1[p] > 0 ? n1[p] : 0, b = n2[p] > 0 ? n2[p] : 0;
        tiles += ((a + TS - 1) / TS) * ((b + TS - 1) / TS);
        cells += align_up((a + b) * a, 64);          // skewed: (N+M-1) rows of N
    }
    int64_t o = 0;
    auto take = [&](int64_t bytes) { int64_t r = o; o += align_up(bytes, 256); return r; };
    w.meta_off = take(P * (int64_t)sizeof(PairMeta));
    w.tilepair_off = take(tiles * 4);
    w.norm1_off = take(rows1 * 4);
    w.norm2_off = take(rows2 * 4);
    w.bad_off = take(P * 4);
    w.dist_off = take(cells * 4);
    w.dirs_off = take(cells);
    w.total = o;
    w.total_tiles = tiles;
    w.dist_floats = cells;
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
    return align_up(npairs * (int64_t)sizeof(PairMeta), 256) + align_up(w.total_tiles * 4, 256);
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
    if (host_stage_bytes < meta_bytes + align_up(w.total_tiles * 4, 256)) { set_error("dtw: host staging buffer too small"); return ABN_E_WORKSPACE; }

    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)ws;
    PairMeta* hm = (PairMeta*)host_stage;
    int32_t* htp = (int32_t*)((char*)host_stage + meta_bytes);
    int64_t tiles = 0, cells = 0;
    for (int64_t p = 0; p < npairs; ++p) {
        const int64_t a = n1_host[p], b = n2_host[p];
        const int64_t tm = (a + TS - 1) / TS, tn = (b + TS - 1) / TS;
        hm[p].off1 = off1_host[p]; hm[p].off2 = off2_host[p];
        hm[p].n1 = (int32_t)a; hm[p].n2 = (int32_t)b;
        hm[p].ws_off = cells;
        hm[p].tile0 = (int32_t)tiles;
        hm[p].tiles_n = (int32_t)(tn > 0 ? tn : 1);
        for (int64_t t = 0; t < tm * tn; ++t) htp[tiles + t] = (int32_t)p;
        tiles += tm * tn;
        cells += align_up((a + b) * a, 64);
    }
    if (hipMemcpyAsync(base + w.meta_off, hm, npairs * sizeof(PairMeta), hipMemcpyHostToDevice, st) != hipSuccess ||
        (tiles > 0 && hipMemcpyAsync(base + w.tilepair_off, htp, tiles * 4, hipMemcpyHostToDevice, st) != hipSuccess) ||
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
    int32_t max_n1 = 1;
    for (int64_t p = 0; p < npairs; ++p) max_n1 = n1_host[p] > max_n1 ? n1_host[p] : max_n1;
    const int lds_rows = (int)align_up(max_n1 + 1, 2);
    hipLaunchKernelGGL(dp_kernel, dim3((unsigned)npairs), dim3(64), (size_t)3 * lds_rows * sizeof(double), st,
                       (const PairMeta*)(base + w.meta_off), (const float*)(base + w.dist_off),
                       (uint8_t*)(base + w.dirs_off), (const int32_t*)(base + w.bad_off), path1, path2, path_len,
                       path_stride, total_cost, lds_rows);
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
