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
#include <stdlib.h>

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
    int64_t dir_off;         // dword offset of this pair's packed back-pointers
    int32_t nbands;          // ceil(n1 / 32): bands of 32 rows
    int32_t nrounds;         // ceil((n2 + 31) / 32): 32-diagonal rounds per band
};

constexpr int BAND = 32;     // rows of token 1 a lane-half sweeps at once (one row per lane)
constexpr int KCH = 40;      // k processed per MFMA chain segment (the 40-d filterbank frame in one piece)
constexpr int KST = KCH / 2; // v_mfma_f32_32x32x2_f32 steps per segment

// back-pointer codes
enum { DIR_DIAG = 0, DIR_UP = 1, DIR_LEFT = 2 };

struct DtwP {
    const float* feats1;
    const float* feats2;
    const PairMeta* meta;
    const int32_t* order;        // work queue: pair ids, largest first
    int32_t* counter;            // next queue position (zeroed by the host call)
    int32_t npairs;
    int32_t D;
    uint32_t* dirs;              // 2-bit back-pointers, 16 diagonals per dword: [pair][band][16-diag group][32 rows]
    double* bound;               // band boundary rows: [slot][2][mcap] accumulated costs of a band's last row
    int64_t mcap;
    int32_t* bad;                // [pair] set when a distance is NaN / negative (utils.py:59)
    double* total_cost;          // [pair] or NULL
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

// lane l receives lane l-1's value (lanes 0 and 32 are overridden by the caller)
__device__ __forceinline__ double wave_shr1(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);      // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ int64_t readlane64(int64_t v, int l)
{
    const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, l), hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// LDS hand-off inside ONE wavefront: its LDS operations complete in order, so keeping the
// compiler from moving accesses across is all that is needed
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// MFMA operand fragment of one row for k in [k0, k0 + 40): element t = row[k0 + 2t + h]
// (h = lane / 32), zero past D -- fma(0, 0, acc) == acc exactly, so padding never shows
template <bool VEC>
__device__ __forceinline__ void load_frag(float* __restrict__ f, const float* __restrict__ row, int k0, int D, int h)
{
    if (VEC) {
#pragma unroll
        for (int q = 0; q < KST / 2; ++q) {
            const int k = k0 + 4 * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < D) v = *reinterpret_cast<const float4*>(row + k);          // D % 4 == 0 on this path
            f[2 * q] = h ? v.y : v.x;
            f[2 * q + 1] = h ? v.w : v.z;
        }
    } else {
#pragma unroll
        for (int t = 0; t < KST; ++t) {
            const int k = k0 + 2 * t + h;
            f[t] = k < D ? row[k] : 0.0f;
        }
    }
}

// np.sum(v ** 2) for 40 floats held as ten float4: numpy's order for n = 40 (8 partial sums
// over 5 passes, then the tree)
__device__ __forceinline__ float sumsq40(const float4* v)
{
    float r[8];
    r[0] = v[0].x * v[0].x; r[1] = v[0].y * v[0].y; r[2] = v[0].z * v[0].z; r[3] = v[0].w * v[0].w;
    r[4] = v[1].x * v[1].x; r[5] = v[1].y * v[1].y; r[6] = v[1].z * v[1].z; r[7] = v[1].w * v[1].w;
#pragma unroll
    for (int c = 1; c < 5; ++c) {
        r[0] += v[2 * c].x * v[2 * c].x; r[1] += v[2 * c].y * v[2 * c].y;
        r[2] += v[2 * c].z * v[2 * c].z; r[3] += v[2 * c].w * v[2 * c].w;
        r[4] += v[2 * c + 1].x * v[2 * c + 1].x; r[5] += v[2 * c + 1].y * v[2 * c + 1].y;
        r[6] += v[2 * c + 1].z * v[2 * c + 1].z; r[7] += v[2 * c + 1].w * v[2 * c + 1].w;
    }
    return ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
}

// one 40-float row -> MFMA fragment (element t = row[2t + h]) and its norm, from ONE set of loads
__device__ __forceinline__ float load_row40(float* __restrict__ f, const float* __restrict__ row, int h)
{
    float4 v[10];
#pragma unroll
    for (int q = 0; q < 10; ++q) v[q] = *reinterpret_cast<const float4*>(row + 4 * q);
#pragma unroll
    for (int q = 0; q < 10; ++q) {
        f[2 * q] = h ? v[q].y : v[q].x;
        f[2 * q + 1] = h ? v[q].w : v[q].z;
    }
    return sqrtf(sumsq40(v));
}

// ---------------------------------------------------------------------------------------
// The whole alignment of a pair in one kernel, the cost matrix never leaves the CU:
//
//  * one wavefront per workgroup; its two lane-halves are two independent SLOTS, each
//    sweeping one token pair at a time (pairs come from a work queue, largest first);
//  * a slot walks its pair in BANDS of 32 rows of token 1 (one row per lane) and, inside a
//    band, in ROUNDS of 32 anti-diagonals.  A round first PRODUCES the 32 x 32 block of
//    distances of the next 32 columns -- 20 v_mfma_f32_32x32x2_f32 on the fp32 matrix
//    cores (one sequential fma chain per cell = the reference's sgemm), then the
//    reference's division / acosf / pi per cell on all 64 lanes (dist_ref.h) -- and drops
//    it into a 64-diagonal ring in LDS, diagonal-major (row = (i + j) & 63, column = row
//    of the band: conflict-free both ways).  Then the slot's 32 lanes SWEEP 32
//    anti-diagonals of float64 costs: cost = d + min(diag, up, left), first minimum in
//    that order wins; the previous two diagonals live in registers, the neighbour row
//    comes over a DPP wave shift, the row above the band (the previous band's last row)
//    through a small LDS window of a per-slot scratch row;
//  * HBM sees the features once per band (L2), 2 bits per cell of back-pointers
//    (16 diagonals per dword, [band][group][row]: 128-byte stores) and the band boundary
//    rows: ~0.3 B per cell.  Tokens of any length.
//
// The produce phase is throughput code on all 64 lanes, the sweep a dependency chain on
// 32 + 32: co-resident wavefronts (8 per CU at this LDS footprint) interleave the two.
// ---------------------------------------------------------------------------------------
template <bool VEC, bool F40>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void dtw_fused_kernel(DtwP P)
{
    __shared__ __attribute__((aligned(16))) float ring[2][64][BAND];
    __shared__ __attribute__((aligned(16))) float ny_s[2][BAND];
    __shared__ double top_s[2][BAND];
    __shared__ double bot_s[2][2 * BAND];
    const int lane = threadIdx.x, half = lane >> 5, n = lane & 31;
    const int D = P.D;
    const double INF = __builtin_inf();
    const int slot = 2 * (int)blockIdx.x + half;
    double* const bnd = P.bound + (int64_t)slot * 2 * P.mcap;

    // slot state, uniform inside a lane-half
    int pair = -1, N = 0, M = 0, nbands = 0, nrounds = 0, band = 0, u = 0;
    int64_t xoff = 0, yoff = 0, dir_off = 0;
    bool exhausted = false;
    // sweep state, per lane (= per row of the band)
    double p1 = INF, p2 = INF, topprev = INF;
    float xf[2][KST];                      // x fragments of both slots' bands (D <= 40: loaded once per band)
    float nx[2] = {0.0f, 0.0f};
    int anybad[2] = {0, 0};

    for (;;) {
        // ---- work queue: a slot without a pair takes the next one
        {
            const bool need = pair < 0 && !exhausted;
            int idx = -1;
            if (need && n == 0) idx = atomicAdd(P.counter, 1);
            idx = __shfl(idx, half * 32);
            if (need) {
                if (idx < P.npairs) {
                    pair = P.order[idx];
                    const PairMeta m = P.meta[pair];
                    N = m.n1; M = m.n2; nbands = m.nbands; nrounds = m.nrounds;
                    xoff = m.off1 * D; yoff = m.off2 * D; dir_off = m.dir_off;
                    band = 0; u = 0;
                    p1 = INF; p2 = INF; topprev = 0.0;       // the virtual cell (-1, -1) costs 0
                } else {
                    exhausted = true;
                }
            }
        }
        if (!__any(pair >= 0)) break;
        const bool active = pair >= 0;
        const int i0 = band * BAND, j0 = u * BAND;

        // the band's boundary row above: this round's 32 columns of the previous band's last row
        double topv = INF;
        if (active && band > 0 && j0 + n < M)
            topv = __hip_atomic_load(&bnd[(int64_t)((band & 1) ^ 1) * P.mcap + j0 + n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

        // ---- produce: distances of columns j0 .. j0+31 for both slots, all 64 lanes
        int q_pair[2], qM[2], qN[2], qj0[2], qi0[2];
        bool have[2];
        f32x16 acc[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int src = 32 * q;
            q_pair[q] = __builtin_amdgcn_readlane(pair, src);
            qM[q] = __builtin_amdgcn_readlane(M, src); qN[q] = __builtin_amdgcn_readlane(N, src);
            qj0[q] = __builtin_amdgcn_readlane(j0, src); qi0[q] = __builtin_amdgcn_readlane(i0, src);
            have[q] = q_pair[q] >= 0 && qj0[q] < qM[q];                     // wave-uniform
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;
        }
        if (F40) {
            // one MFMA chain segment per cell: rows -> fragments + norms from one set of loads,
            // then the two slots' chains interleaved on the matrix pipe
            float yf[2][KST];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (!have[q]) continue;
                const float* xrow = P.feats1 + readlane64(xoff, 32 * q) + (int64_t)min(qi0[q] + n, qN[q] - 1) * D;
                const float* yrow = P.feats2 + readlane64(yoff, 32 * q) + (int64_t)min(qj0[q] + n, qM[q] - 1) * D;
#ifndef ABN_EXP_NOLOAD
                const float ny = load_row40(yf[q], yrow, half);
#else
                float ny = 1.0f + (float)(intptr_t)yrow * 1e-30f;
                for (int t = 0; t < KST; ++t) yf[q][t] = ny;
#endif
                if (half == 0) ny_s[q][n] = ny;
                __builtin_amdgcn_sched_barrier(0);          // one row's ten float4 in flight at a time (registers)
                if (qj0[q] == 0) nx[q] = load_row40(xf[q], xrow, half);
                __builtin_amdgcn_sched_barrier(0);
            }
#ifndef ABN_EXP_NOMFMA
#pragma unroll
            for (int t = 0; t < KST; ++t) {                                 // A = token 2 rows (j), B = token 1 rows (i)
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(yf[0][t], xf[0][t], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(yf[1][t], xf[1][t], acc[1], 0, 0, 0);
            }
#else
            for (int t = 0; t < KST; ++t) { acc[0][t & 15] += yf[0][t] * xf[0][t]; acc[1][t & 15] += yf[1][t] * xf[1][t]; }
#endif
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (!have[q]) continue;
                const float* xrow = P.feats1 + readlane64(xoff, 32 * q) + (int64_t)min(qi0[q] + n, qN[q] - 1) * D;   // rows past the token
                const float* yrow = P.feats2 + readlane64(yoff, 32 * q) + (int64_t)min(qj0[q] + n, qM[q] - 1) * D;   // end are clamped
                if (qj0[q] == 0) nx[q] = row_norm_numpy(xrow, D);           // np.sqrt(np.sum(x ** 2, axis=1)), numpy's order
                if (half == 0) ny_s[q][n] = row_norm_numpy(yrow, D);
                for (int k0 = 0; k0 < D; k0 += KCH) {
                    float yf[KST];
                    if (D > KCH || qj0[q] == 0) load_frag<VEC>(xf[q], xrow, k0, D, half);
                    load_frag<VEC>(yf, yrow, k0, D, half);
#ifndef ABN_EXP_NOMFMA
#pragma unroll
                    for (int t = 0; t < KST; ++t)
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(yf[t], xf[q][t], acc[q], 0, 0, 0);
#endif
                }
            }
        }
        wave_lds_sync();                                                   // ny_s is staged
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (!have[q]) continue;
            // accumulator r of lane (n, h) is cell (row i0 + n, column j0 + m), m = (r & 3) + 8 (r >> 2) + 4 h
            const float nxl = nx[q];
            const bool plain = __all(norm_is_plain(nxl) && norm_is_plain(ny_s[q][n]));     // the usual case
            const bool rowok = qi0[q] + n < qN[q];
            bool bad = false;
            float* rw = &ring[q][0][n];
            const int rbase = (qj0[q] & 32) + n + 4 * half;                 // ring row of accumulator 0
            auto epilogue = [&](auto zr) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 ny4 = *reinterpret_cast<const float4*>(&ny_s[q][8 * g + 4 * half]);
                    const float nyv[4] = {ny4.x, ny4.y, ny4.z, ny4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int m = 8 * g + 4 * half + e;
#ifndef ABN_EXP_NOEPI
                        const float d = angular_distance_ref<decltype(zr)::value>(acc[q][4 * g + e], nxl, nyv[e]);
#else
                        const float d = fabsf(acc[q][4 * g + e] * nxl * nyv[e]) * 1e-3f;
#endif
                        bad |= rowok && qj0[q] + m < qM[q] && !(d >= 0.0f);  // utils.py:59 assert
#ifndef ABN_EXP_NORING
                        rw[((rbase + 8 * g + e) & 63) * BAND] = d;
#else
                        if (d == 12345.0f) rw[0] = d;
#endif
                    }
                }
            };
            if (plain) epilogue(std::true_type{}); else epilogue(std::false_type{});
            anybad[q] |= __any(bad) ? 1 : 0;
        }
        // boundary window: the row above the band for this round's 32 diagonals
        top_s[half][n] = topv;
        wave_lds_sync();

        // ---- sweep: 32 anti-diagonals; lane n owns row i0 + n, at diagonal s its column is s - n
        {
            const float* rg = &ring[half][(u & 1) * BAND][n];             // diagonal 32 u + e lives in ring row (32 u + e) & 63
            const int sbase = j0;                                          // band-local diagonal of step 0
            const bool rowok = active && i0 + n < N;
            const bool feed_next = active && band + 1 < nbands;             // the last row feeds the next band
            const bool last_lane = n == BAND - 1 && feed_next;
            uint32_t* dptr = P.dirs + dir_off + ((int64_t)(band * 2 * nrounds + 2 * u) * BAND + n);
#ifdef ABN_EXP_NODP
            for (int e16 = 0; e16 < 0; e16 += 16) {
#else
            for (int e16 = 0; e16 < BAND; e16 += 16) {
#endif
            uint32_t bits = 0u;
#pragma unroll
            for (int ee = 0; ee < 16; ++ee) {
                const int e = e16 + ee;
                const int j = sbase + e - n;
                const float dist = rg[e * BAND];
                const double topc = top_s[half][e];
                double up = wave_shr1(p1), dg = wave_shr1(p2);
                if (n == 0) { up = topc; dg = topprev; topprev = topc; }
                const double left = p1;
                // straight-line selects: first minimum in the order diag, up, left
                const bool take_up = up < dg;
                const double b1 = take_up ? up : dg;
                const bool take_left = left < b1;
                const double best = take_left ? left : b1;
                const uint32_t dir = take_left ? (uint32_t)DIR_LEFT : take_up ? (uint32_t)DIR_UP : (uint32_t)DIR_DIAG;
                const double cost = (double)dist + best;
                const bool on = rowok && (uint32_t)j < (uint32_t)M;
                p2 = left;
                p1 = on ? cost : left;                                     // past the row's end the last cost stays put
                bits |= on ? dir << (2 * ee) : 0u;
                if (last_lane && on) bot_s[half][j & 63] = cost;
            }
            if (active) dptr[(e16 >> 4) * BAND] = bits;
            }
            wave_lds_sync();
            // the band's last row, for the band below: block u-1 of 32 columns is complete now
            if (feed_next) {
                double* dst = bnd + (int64_t)(band & 1) * P.mcap;
                if (u >= 1) {
                    const int j = (u - 1) * BAND + n;
                    if (j < M) __hip_atomic_store(&dst[j], bot_s[half][j & 63], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (u == nrounds - 1) {
                    const int j = u * BAND + n;
                    if (j < M) __hip_atomic_store(&dst[j], bot_s[half][j & 63], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }

        // ---- advance the slot
        if (active) {
            ++u;
            if (u == nrounds) {
                u = 0;
                ++band;
                if (band == nbands) {
                    // lane (N-1) % 32 still holds cost(N-1, M-1)
                    if (P.total_cost && n == ((N - 1) & 31)) P.total_cost[pair] = p1;
                    if (n == 0 && (half ? anybad[1] : anybad[0])) P.bad[pair] = 1;
                    pair = -1;
                }
                p1 = INF; p2 = INF; topprev = INF;
            }
        }
        // The boundary row of a finished band is re-read by this same wavefront: its stores are
        // write-through (sc1) and only have to be complete; the loads bypass the L1 (sc1).  An
        // agent-scope release would write the XCD's whole L2 back, once per band.
        if (__any(active && u == 0)) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
        }
        // a slot that finished its pair: its bad flag starts afresh with the next pair
        if (__builtin_amdgcn_readlane(pair, 0) < 0) anybad[0] = 0;
        if (__builtin_amdgcn_readlane(pair, 32) < 0) anybad[1] = 0;
    }
}

// ---------------------------------------------------------------------------------------
// Producer / consumer form of the same algorithm for 40-value frames (the hot case): a
// workgroup is TWO wavefronts sharing two slots.
//   wave 0 (producer)  rows -> fragments + norms, 2 x 20 MFMAs, the reference's distance per
//                      cell on all 64 lanes: throughput code, ~170 registers, no DP state;
//   wave 1 (consumer)  the two slots' float64 sweeps (32 lanes each): a dependency chain,
//                      ~60 registers, plus boundary rows and back-pointer stores.
// The producer works one block AHEAD: while the consumer sweeps round t it computes the block
// of round t+1.  A block's 63 anti-diagonals fall into the ring half the consumer is NOT
// reading (its first 32 diagonals: written at once) and the half it IS reading (the other 31:
// held in registers until the consumer has finished the round -- barrier A -- then written,
// barrier B).  Ring rows are counted by a per-slot round counter that never resets, so the
// halves alternate across band and pair changes too.  The producer also drives the slots'
// schedule (next block / next band / next pair from the queue) and publishes one descriptor
// per slot and round through LDS.  Twice the wavefronts per CU at the same LDS footprint:
// the producers keep the vector ALU fed while the consumers wait on their chains.
// ---------------------------------------------------------------------------------------
#ifdef ABN_DTW_STAMPS         // diagnostic build only (tools/dtw_stamps.py): cycles per phase, summed over workgroups
__device__ unsigned long long g_dtw_cycles[16];
#define PSTAMP(k) do { if (lane == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_dtw_cycles[k], t_ - tlast); tlast = t_; } } while (0)
#else
#define PSTAMP(k) do {} while (0)
#endif

struct RoundDesc {            // what a slot does in one round (LDS; written by the producer)
    int32_t pair;             // -1: slot idle
    int32_t N, M, nbands, nrounds, band, u;
    int32_t v;                // the slot's running round counter (ring phase)
    int64_t xoff, yoff, dir_off;
};

__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 4))) void dtw_pc_kernel(DtwP P)
{
    __shared__ __attribute__((aligned(16))) float ring[2][64][BAND];
    __shared__ __attribute__((aligned(16))) float ny_s[2][BAND];
    __shared__ double top_s[2][BAND];
    __shared__ double bot_s[2][2 * BAND];
    __shared__ RoundDesc desc[2][2];                   // [round parity][slot]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5, n = lane & 31;
    const int D = P.D;

    if (wave == 0) {
        // =========================== producer ===========================
        bool exhausted = false;
        auto fetch = [&](RoundDesc& d) {               // next pair of the queue (wave-uniform values)
            d.pair = -1;
            if (exhausted) return;
            int idx = 0;
            if (lane == 0) idx = atomicAdd(P.counter, 1);
            idx = __builtin_amdgcn_readfirstlane(idx);
            if (idx >= P.npairs) { exhausted = true; return; }
            const int p = __builtin_amdgcn_readfirstlane(P.order[idx]);
            const PairMeta* m = P.meta + p;
            d.pair = p;
            d.N = __builtin_amdgcn_readfirstlane(m->n1); d.M = __builtin_amdgcn_readfirstlane(m->n2);
            d.nbands = __builtin_amdgcn_readfirstlane(m->nbands); d.nrounds = __builtin_amdgcn_readfirstlane(m->nrounds);
            d.xoff = readlane64(m->off1, 0) * D; d.yoff = readlane64(m->off2, 0) * D; d.dir_off = readlane64(m->dir_off, 0);
            d.band = 0; d.u = 0;
        };
        RoundDesc cur[2];                              // the block being produced (scalar registers)
        uint32_t orbits[2] = {0u, 0u};                 // OR of the distances' bit patterns: >= 0x7f800000 iff one is NaN
        float dreg[2][16];

        // block of `cur[q]`: loads, MFMA, distances into dreg[q].  One slot at a time and the band's
        // rows re-read every round (they are L1 / L2 hot): the kernel has to fit 128 registers so
        // that four wavefronts share a SIMD.
        auto produce = [&](bool have0, bool have1) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (!(q ? have1 : have0)) continue;
                const RoundDesc& d = cur[q];
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
                float nxl, ny;
                {
                    float xf[KST], yf[KST];
                    const float* yrow = P.feats2 + d.yoff + (int64_t)min(d.u * BAND + n, d.M - 1) * D;
                    ny = load_row40(yf, yrow, half);
                    __builtin_amdgcn_sched_barrier(0);
                    const float* xrow = P.feats1 + d.xoff + (int64_t)min(d.band * BAND + n, d.N - 1) * D;
                    nxl = load_row40(xf, xrow, half);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < KST; ++t)                           // A = token 2 rows (j), B = token 1 rows (i)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(yf[t], xf[t], acc, 0, 0, 0);
                }
                if (half == 0) ny_s[q][n] = ny;
                wave_lds_sync();                                           // ny_s (this wave's own writes)
                const bool plain = __all(norm_is_plain(nxl) && norm_is_plain(ny));
                uint32_t ob = 0u;
                auto epilogue = [&](auto pl) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        __builtin_amdgcn_sched_barrier(0);
                        const float4 ny4 = *reinterpret_cast<const float4*>(&ny_s[q][8 * g + 4 * half]);
                        const float nyv[4] = {ny4.x, ny4.y, ny4.z, ny4.w};
                        if constexpr (decltype(pl)::value) {          // two cells per instruction (dist_ref.h)
#pragma unroll
                            for (int e = 0; e < 4; e += 2) {
                                const f32x2 dv = angular_distance_plain2(f32x2{acc[4 * g + e], acc[4 * g + e + 1]}, nxl, f32x2{nyv[e], nyv[e + 1]});
                                ob |= __float_as_uint(dv.x) | __float_as_uint(dv.y);      // padded rows / columns repeat real ones: no masking needed
                                dreg[q][4 * g + e] = dv.x;
                                dreg[q][4 * g + e + 1] = dv.y;
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float dv = angular_distance_ref<false>(acc[4 * g + e], nxl, nyv[e]);
                                ob |= __float_as_uint(dv);
                                dreg[q][4 * g + e] = dv;
                            }
                        }
                    }
                };
                if (plain) epilogue(std::true_type{}); else epilogue(std::false_type{});
                orbits[q] |= ob;
            }
        };
        // accumulator c of lane (n, h) is column m = (c & 3) + 8 (c >> 2) + 4 h of the block: diagonal m + n
        auto write_ring = [&](int q, bool late) {
            const int v = cur[q].v;
            float* rw = &ring[q][0][n];
            const int base = ((v & 1) * BAND) + n + 4 * half;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int mo = (c & 3) + 8 * (c >> 2);
                const int dg = n + 4 * half + mo;                           // diagonal inside the block, 0 .. 62
                if ((dg >= BAND) == late) rw[((base + mo) & 63) * BAND] = dreg[q][c];
            }
        };
        auto finish_pair_flags = [&](int q) {          // the block just produced was the pair's last one
            const RoundDesc& d = cur[q];
            const bool last_block = d.band + 1 == d.nbands && (d.u + 1) * BAND >= d.M;
            if (last_block) {
                const bool isbad = __any(orbits[q] >= 0x7f800000u);
                if (isbad && lane == 0) P.bad[d.pair] = 1;                  // utils.py:59: NaN (or negative) distance
                orbits[q] = 0u;
            }
        };

        // prologue: first pairs, their first blocks, complete in the ring before the consumer starts
        fetch(cur[0]); cur[0].v = 0;
        fetch(cur[1]); cur[1].v = 0;
        if (lane == 0) { desc[0][0] = cur[0]; desc[0][1] = cur[1]; }
        {
            const bool h0 = cur[0].pair >= 0, h1 = cur[1].pair >= 0;
            produce(h0, h1);
            if (h0) { write_ring(0, false); write_ring(0, true); finish_pair_flags(0); }
            if (h1) { write_ring(1, false); write_ring(1, true); finish_pair_flags(1); }
        }
        __syncthreads();                                                    // barrier B of "round -1"

#ifdef ABN_DTW_STAMPS
        unsigned long long tlast = __builtin_amdgcn_s_memtime();
#endif
        for (int t = 0;; ++t) {
            if (cur[0].pair < 0 && cur[1].pair < 0) break;                  // = desc[t & 1]: the consumer sees the same
            PSTAMP(0);
            // next round of each slot
            bool have[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                RoundDesc& d = cur[q];
                if (d.pair >= 0) {
                    const int v = d.v + 1;
                    if (d.u + 1 < d.nrounds) d.u = d.u + 1;
                    else if (d.band + 1 < d.nbands) { d.band = d.band + 1; d.u = 0; }
                    else fetch(d);
                    d.v = v;
                }
                have[q] = d.pair >= 0 && d.u * BAND < d.M;
            }
            if (lane == 0) { desc[(t + 1) & 1][0] = cur[0]; desc[(t + 1) & 1][1] = cur[1]; }
            PSTAMP(1);
            produce(have[0], have[1]);
            PSTAMP(2);
            if (have[0]) write_ring(0, false);
            if (have[1]) write_ring(1, false);
            PSTAMP(3);
            __syncthreads();                                                // A: the consumer has finished round t
            PSTAMP(4);
            if (have[0]) { write_ring(0, true); finish_pair_flags(0); }
            if (have[1]) { write_ring(1, true); finish_pair_flags(1); }
            __syncthreads();                                                // B: round t+1 is complete in the ring
            PSTAMP(5);
        }
    } else {
        // =========================== consumer ===========================
        const double INF = __builtin_inf();
        double* const bnd = P.bound + (int64_t)(2 * (int)blockIdx.x + half) * 2 * P.mcap;
        double p1 = INF, p2 = INF, topprev = INF;
        int prev_pair = -1;
        __syncthreads();                                                    // B of "round -1": first blocks are in the ring
#ifdef ABN_DTW_STAMPS
        unsigned long long tlast = __builtin_amdgcn_s_memtime();
#endif
        for (int t = 0;; ++t) {
            const RoundDesc* dd = desc[t & 1];
            if (dd[0].pair < 0 && dd[1].pair < 0) break;
            PSTAMP(8);
            const RoundDesc& d = dd[half];
            const int pair = d.pair, N = d.N, M = d.M, nbands = d.nbands, nrounds = d.nrounds, band = d.band, u = d.u, v = d.v;
            const bool active = pair >= 0;
            if (active && u == 0) {                      // a band starts: fresh diagonals
                p1 = INF; p2 = INF;
                topprev = (band == 0) ? 0.0 : INF;       // a pair starts from the virtual cell (-1, -1)
            }
            const int i0 = band * BAND, j0 = u * BAND;
            double topv = INF;
            if (active && band > 0 && j0 + n < M)
                topv = __hip_atomic_load(&bnd[(int64_t)((band & 1) ^ 1) * P.mcap + j0 + n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            top_s[half][n] = topv;
            wave_lds_sync();
            PSTAMP(9);
            {
                const float* rg = &ring[half][(v & 1) * BAND][n];
                const bool rowok = active && i0 + n < N;
                const bool feed_next = active && band + 1 < nbands;
                const bool last_lane = n == BAND - 1 && feed_next;
                uint32_t* dptr = P.dirs + d.dir_off + ((int64_t)(band * 2 * nrounds + 2 * u) * BAND + n);
                for (int e16 = 0; e16 < BAND; e16 += 16) {
                    uint32_t bits = 0u;
#pragma unroll
                    for (int ee = 0; ee < 16; ++ee) {
                        const int e = e16 + ee;
                        const int j = j0 + e - n;
                        const float dist = rg[e * BAND];
                        const double topc = top_s[half][e];
                        double up = wave_shr1(p1), dg = wave_shr1(p2);
                        if (n == 0) { up = topc; dg = topprev; topprev = topc; }
                        const double left = p1;
                        const bool take_up = up < dg;
                        const double b1 = take_up ? up : dg;
                        const bool take_left = left < b1;
                        const double best = take_left ? left : b1;
                        const uint32_t dir = take_left ? (uint32_t)DIR_LEFT : take_up ? (uint32_t)DIR_UP : (uint32_t)DIR_DIAG;
                        const double cost = (double)dist + best;
                        const bool on = rowok && (uint32_t)j < (uint32_t)M;
                        p2 = left;
                        p1 = on ? cost : left;
                        bits |= on ? dir << (2 * ee) : 0u;
                        if (last_lane && on) bot_s[half][j & 63] = cost;
                    }
                    if (active) dptr[(e16 >> 4) * BAND] = bits;
                }
                wave_lds_sync();
                if (feed_next) {
                    double* dst = bnd + (int64_t)(band & 1) * P.mcap;
                    if (u >= 1) {
                        const int j = (u - 1) * BAND + n;
                        if (j < M) __hip_atomic_store(&dst[j], bot_s[half][j & 63], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (u == nrounds - 1) {
                        const int j = u * BAND + n;
                        if (j < M) __hip_atomic_store(&dst[j], bot_s[half][j & 63], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (active && u + 1 == nrounds && band + 1 == nbands && P.total_cost && n == ((N - 1) & 31))
                    P.total_cost[pair] = p1;             // lane (N-1) % 32 holds cost(N-1, M-1)
            }
            // a finished band's boundary row is re-read by this wavefront: write-through stores that
            // only have to be complete; the loads bypass the L1
            if (__any(active && u + 1 == nrounds)) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_s_waitcnt(0);
            }
            PSTAMP(10);
            __syncthreads();                                                // A
            PSTAMP(11);
            __syncthreads();                                                // B
            PSTAMP(12);
        }
    }
}

#ifdef ABN_DTW_STAMPS
extern "C" int abn_debug_dtw_cycles(unsigned long long* out16, int reset)
{
    if (reset) { unsigned long long z[16] = {}; return hipMemcpyToSymbol(HIP_SYMBOL(abn::g_dtw_cycles), z, 128) == hipSuccess ? 0 : -1; }
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(abn::g_dtw_cycles), 128) == hipSuccess ? 0 : -1;
}
#endif

// Walks the back-pointers of the pairs from (N-1, M-1) to (0, 0).  The k-th cell visited is the
// k-th from the END of the path, so the path is written right-aligned into its output row --
// entries [path_stride - len, path_stride) -- in forward order, with no second pass.
// One thread per pair.  The walk is a chain of dependent loads, so a thread fetches a WINDOW at
// a time -- the 8 rows at and above its position, the current 16-diagonal group and the one
// before it: two runs of 8 consecutive dwords, 16 independent loads -- into its private strip
// of LDS and takes 8 to 17 steps from there at LDS latency.
constexpr int TB_ROWS = 8;
__global__ __launch_bounds__(64) void dtw_traceback_kernel(const PairMeta* __restrict__ meta, const int32_t* __restrict__ order,
                                                           int npairs, const uint32_t* __restrict__ dirs,
                                                           const int32_t* __restrict__ bad,
                                                           int32_t* __restrict__ path1, int32_t* __restrict__ path2,
                                                           int32_t* __restrict__ path_len, int64_t path_stride,
                                                           double* __restrict__ total_cost)
{
    __shared__ uint32_t win[2 * TB_ROWS][64];            // [group offset * 8 + row offset][thread]: conflict-free
    const int lane = threadIdx.x;
    if ((int)(blockIdx.x * 64 + lane) >= npairs) return;
    // queue order (pairs of similar size side by side): the threads of a wavefront walk paths of
    // similar length.  Empty pairs are not in the queue: their path_len was zeroed by the call.
    const int p = order[blockIdx.x * 64 + lane];
    const PairMeta m = meta[p];
    const int N = m.n1, M = m.n2;
    if (bad[p]) {
        if (total_cost) total_cost[p] = 0.0;
        return;                                       // path_len[p] stays 0: the pair is dropped
    }
    const uint32_t* dp = dirs + m.dir_off;
    const int nsg = 2 * m.nrounds;
    int32_t* o1 = path1 + (int64_t)p * path_stride + (path_stride - 1);
    int32_t* o2 = path2 + (int64_t)p * path_stride + (path_stride - 1);
    int i = N - 1, j = M - 1, k = 0;
    o1[0] = i; o2[0] = j;
    while (i > 0 || j > 0) {
        const int b = i >> 5, r = i & 31, g = (j + r) >> 4;
        const int rlo = max(r - (TB_ROWS - 1), 0);
        const uint32_t* src = dp + (int64_t)(b * nsg + g) * BAND + rlo;     // rows rlo .. rlo+7 <= 31 exist in every band
        uint32_t w0[TB_ROWS], w1[TB_ROWS];
#pragma unroll
        for (int q = 0; q < TB_ROWS; ++q) {
            w0[q] = src[q];
            w1[q] = g > 0 ? src[q - BAND] : 0u;
        }
#pragma unroll
        for (int q = 0; q < TB_ROWS; ++q) {
            win[q][lane] = w0[q];
            win[TB_ROWS + q][lane] = w1[q];
        }
        // the window holds rows rlo .. r of this band on diagonals 16 (g-1) .. 16 g + 15
        while (i > 0 || j > 0) {
            const int rr = i & 31, s = j + rr;
            if ((i >> 5) != b || rr < rlo || (s >> 4) < g - 1) break;
            const uint32_t w = win[(g - (s >> 4)) * TB_ROWS + (rr - rlo)][lane];
            const int dir = (w >> (2 * (s & 15))) & 3;
            if (dir == DIR_DIAG) { --i; --j; } else if (dir == DIR_UP) --i; else --j;
            ++k;
            o1[-k] = i;
            o2[-k] = j;
        }
    }
    path_len[p] = k + 1;
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
    const float v = angular_distance_ref<false>(dot, row_norm_numpy(a, D), row_norm_numpy(b, D));
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
__global__ void arccos_kernel(const float* __restrict__ x, int64_t n, int over_pi, float* __restrict__ out)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float a = acosf_ref(x[i]);
        out[i] = over_pi ? div_pi(a) : a;
    }
}

struct WsPlan {
    int64_t meta_off, order_off, bad_off, counter_off, dirs_off, bound_off, total;
    int64_t mcap;
    int32_t nwg;
};

}  // namespace abn

using namespace abn;

// Workspace: [PairMeta x P][order x P][bad x P][queue counter][back-pointers][boundary rows]
static WsPlan plan_ws(const int32_t* n1, const int32_t* n2, int64_t P)
{
    WsPlan w;
    int64_t dwords = 0, mmax = 1;
    for (int64_t p = 0; p < P; ++p) {
        const int64_t a = n1[p] > 0 ? n1[p] : 0, b = n2[p] > 0 ? n2[p] : 0;
        if (a > 0 && b > 0) {
            dwords += ((a + BAND - 1) / BAND) * 2 * ((b + 2 * BAND - 1) / BAND) * BAND;
            mmax = b > mmax ? b : mmax;
        }
    }
    w.mcap = align_up(mmax, 32);
    // persistent grid: one wavefront per workgroup, two pairs in flight per wavefront, up to 9
    // wavefronts per CU (LDS); fewer when the boundary rows of that many slots would outgrow 256 MiB
    int64_t nwg = (P + 1) / 2;
    if (nwg > 256 * 9) nwg = 256 * 9;
    const int64_t cap = (256LL << 20) / (4 * w.mcap * 8);
    if (nwg > cap) nwg = cap < 1 ? 1 : cap;
    if (nwg < 1) nwg = 1;
    w.nwg = (int32_t)nwg;
    int64_t o = 0;
    auto take = [&](int64_t bytes) { int64_t r = o; o += align_up(bytes, 256); return r; };
    w.meta_off = take(P * (int64_t)sizeof(PairMeta));
    w.order_off = take(P * 4);
    w.bad_off = take(P * 4);
    w.counter_off = take(4);
    w.dirs_off = take(dwords * 4);
    w.bound_off = take(2 * nwg * 2 * w.mcap * 8);
    w.total = o;
    return w;
}

extern "C" int64_t abn_dtw_ws_bytes(const int32_t* n1_host, const int32_t* n2_host, int64_t npairs,
                                         int64_t rows1, int64_t rows2)
{
    if (!n1_host || !n2_host || npairs < 0 || rows1 < 0 || rows2 < 0) return -1;
    return plan_ws(n1_host, n2_host, npairs).total;
}

extern "C" int64_t abn_dtw_host_stage_bytes(const int32_t* n1_host, const int32_t* n2_host, int64_t npairs)
{
    if (!n1_host || !n2_host || npairs < 0) return -1;
    return align_up(npairs * (int64_t)sizeof(PairMeta), 256) + align_up(npairs * 4, 256);
}

extern "C" int abn_dtw_batched(const float* feats1, int64_t rows1, const float* feats2, int64_t rows2,
                                    const int64_t* off1_host, const int32_t* n1_host, const int64_t* off2_host,
                                    const int32_t* n2_host, int64_t npairs, int64_t D, int32_t* path1,
                                    int32_t* path2, int32_t* path_len, int64_t path_stride, double* total_cost,
                                    void* ws, int64_t ws_bytes, void* host_stage, int64_t host_stage_bytes,
                                    void* stream)
{
    ABN_REQUIRE(npairs >= 0 && npairs < (1LL << 31) && D >= 1 && D < (1 << 20), "dtw: bad npairs/D");
    if (npairs == 0) return ABN_OK;
    ABN_REQUIRE(feats1 && feats2 && off1_host && n1_host && off2_host && n2_host && path1 && path2 && path_len && ws &&
                    host_stage,
                "dtw: null pointer");
    int64_t maxlen = 0;
    for (int64_t p = 0; p < npairs; ++p) {
        ABN_REQUIRE(n1_host[p] >= 0 && n2_host[p] >= 0, "dtw: negative token length at pair %lld", (long long)p);
        ABN_REQUIRE(off1_host[p] >= 0 && off1_host[p] + n1_host[p] <= rows1 && off2_host[p] >= 0 &&
                        off2_host[p] + n2_host[p] <= rows2,
                    "dtw: pair %lld reads outside the feature arrays", (long long)p);
        const int64_t need = (int64_t)n1_host[p] + n2_host[p] - 1;
        maxlen = need > maxlen ? need : maxlen;
    }
    ABN_REQUIRE(path_stride >= maxlen, "dtw: path_stride %lld < longest possible path %lld", (long long)path_stride,
                (long long)maxlen);
    ABN_REQUIRE(rows1 * D < (1LL << 62) && rows2 * D < (1LL << 62), "dtw: feature array too large");
    const WsPlan w = plan_ws(n1_host, n2_host, npairs);
    if (ws_bytes < w.total) { set_error("dtw: workspace too small (%lld < %lld bytes)", (long long)ws_bytes, (long long)w.total); return ABN_E_WORKSPACE; }
    const int64_t meta_bytes = align_up(npairs * (int64_t)sizeof(PairMeta), 256);
    if (host_stage_bytes < meta_bytes + align_up(npairs * 4, 256)) { set_error("dtw: host staging buffer too small"); return ABN_E_WORKSPACE; }

    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)ws;
    PairMeta* hm = (PairMeta*)host_stage;
    int32_t* hord = (int32_t*)((char*)host_stage + meta_bytes);
    int64_t dwords = 0;
    for (int64_t p = 0; p < npairs; ++p) {
        const int64_t a = n1_host[p], b = n2_host[p];
        hm[p].off1 = off1_host[p]; hm[p].off2 = off2_host[p];
        hm[p].n1 = (int32_t)a; hm[p].n2 = (int32_t)b;
        hm[p].dir_off = dwords;
        hm[p].nbands = (int32_t)((a + BAND - 1) / BAND);
        hm[p].nrounds = (int32_t)((b + 2 * BAND - 1) / BAND);
        if (a > 0 && b > 0) dwords += (int64_t)hm[p].nbands * 2 * hm[p].nrounds * BAND;
    }
    // Work queue: largest pairs first (the short ones fill the tail), empty pairs never queued.
    // A counting sort over the number of rounds a pair needs (clamped: the order among giants is free).
    int64_t nq = 0;
    {
        constexpr int NB = 4096;
        static thread_local std::vector<int32_t> count;
        count.assign(NB + 1, 0);
        auto bucket = [&](int64_t p) {
            const int64_t work = (int64_t)hm[p].nbands * hm[p].nrounds;
            return (int)(NB - 1 - (work < NB ? work : NB - 1));                           // descending
        };
        for (int64_t p = 0; p < npairs; ++p)
            if (n1_host[p] > 0 && n2_host[p] > 0) ++count[bucket(p) + 1];
        for (int b = 0; b < NB; ++b) count[b + 1] += count[b];
        nq = count[NB];
        for (int64_t p = 0; p < npairs; ++p)
            if (n1_host[p] > 0 && n2_host[p] > 0) hord[count[bucket(p)]++] = (int32_t)p;
    }
    if (hipMemcpyAsync(base + w.meta_off, hm, npairs * sizeof(PairMeta), hipMemcpyHostToDevice, st) != hipSuccess ||
        (nq > 0 && hipMemcpyAsync(base + w.order_off, hord, nq * 4, hipMemcpyHostToDevice, st) != hipSuccess) ||
        hipMemsetAsync(base + w.bad_off, 0, (size_t)(w.counter_off + 256 - w.bad_off), st) != hipSuccess) {
        set_error("dtw: metadata upload failed");
        return ABN_E_LAUNCH;
    }
    // dropped and empty pairs keep path_len = 0, total_cost = 0
    if (hipMemsetAsync(path_len, 0, (size_t)npairs * 4, st) != hipSuccess ||
        (total_cost && hipMemsetAsync(total_cost, 0, (size_t)npairs * 8, st) != hipSuccess)) {
        set_error("dtw: clearing the outputs failed");
        return ABN_E_LAUNCH;
    }
    const PairMeta* dm = (const PairMeta*)(base + w.meta_off);
    if (nq > 0) {
        DtwP P = {};
        P.feats1 = feats1; P.feats2 = feats2;
        P.meta = dm;
        P.order = (const int32_t*)(base + w.order_off);
        P.counter = (int32_t*)(base + w.counter_off);
        P.npairs = (int32_t)nq;
        P.D = (int32_t)D;
        P.dirs = (uint32_t*)(base + w.dirs_off);
        P.bound = (double*)(base + w.bound_off);
        P.mcap = w.mcap;
        P.bad = (int32_t*)(base + w.bad_off);
        P.total_cost = total_cost;
        int64_t nwg = (nq + 1) / 2;
        if (nwg > w.nwg) nwg = w.nwg;
        const bool vec = D % 4 == 0 && aligned16(feats1) && aligned16(feats2);
        const bool pipelined = switches().dtw_f40;     // A/B switch of the 40-d specialisation
        const bool pc = switches().dtw_pc;              // A/B switch: producer / consumer form
        if (vec && D == KCH && pc) hipLaunchKernelGGL(dtw_pc_kernel, dim3((unsigned)nwg), dim3(128), 0, st, P);
        else if (vec && D == KCH && pipelined) hipLaunchKernelGGL((dtw_fused_kernel<true, true>), dim3((unsigned)nwg), dim3(64), 0, st, P);
        else if (vec) hipLaunchKernelGGL((dtw_fused_kernel<true, false>), dim3((unsigned)nwg), dim3(64), 0, st, P);
        else hipLaunchKernelGGL((dtw_fused_kernel<false, false>), dim3((unsigned)nwg), dim3(64), 0, st, P);
    }
    if (nq > 0)
        hipLaunchKernelGGL(dtw_traceback_kernel, dim3((unsigned)((nq + 63) / 64)), dim3(64), 0, st, dm,
                           (const int32_t*)(base + w.order_off), (int)nq, (const uint32_t*)(base + w.dirs_off),
                           (const int32_t*)(base + w.bad_off), path1, path2, path_len, path_stride, total_cost);
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

extern "C" int abn_arccos_f32(const float* x, int64_t n, int over_pi, float* out, void* stream)
{
    ABN_REQUIRE(n >= 0, "arccos_f32: negative length");
    if (n == 0) return ABN_OK;
    ABN_REQUIRE(x && out, "arccos_f32: null pointer");
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(arccos_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, (hipStream_t)stream, x, n, over_pi, out);
    ABN_CHECK_LAUNCH("arccos_f32");
    return ABN_OK;
}
