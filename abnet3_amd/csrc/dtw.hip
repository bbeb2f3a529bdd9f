// dtw.hip -- batched DTW frame alignment for gfx950.
//
// Replaces, for a whole batch of token pairs at once,
//   abnet3/utils.py:40-60    cosine_distance  (arccos(cos)/pi, float32 math)
//   abnet3/utils.py:147-153  get_dtw_alignment -> third-party dtw.DTW(...)
// whose per-pair Python/Cython loop is the producer-side hot loop of
// abnet3/dataloader.py:166-261 and :617-671.
//
// Two kernels per batch, the cost matrix never leaves the CU:
//  1. dtw_gang_kernel (40-value frames, the hot case) / dtw_fused_kernel<VEC, F40> (any frame width):
//     a persistent grid walks a work queue of pairs, largest first.  A pair is cut into BANDS of 32
//     rows of token 1 (one row per lane of a half-wavefront: a SLOT) and a band into ROUNDS of 32
//     anti-diagonals.  Per round the 32 x 32 block of angular distances of the next 32 columns comes
//     off the float32 matrix cores -- v_mfma_f32_32x32x2_f32 accumulates every cell as ONE sequential
//     fused chain over k, exactly the oracle's fmaf loop (= what sgemm computes behind the reference's
//     np.dot); the rest of the cell (norms in numpy's summation order, one division, glibc's acosf,
//     / pi) follows the reference's statements operation by operation (dist_ref.h; the file is compiled
//     with -ffp-contract=off), so the values are bit-identical to the C oracle and to the reference's
//     own output (tests/golden/cosdist_libm.npz) -- and the slot's 32 lanes sweep 32 anti-diagonals of
//     float64 costs: cost = d + min(diag, up, left), the first minimum in that order wins (the oracle's
//     tie-break).  Two bits of back-pointer per cell go to HBM, plus the last row of each band for the
//     band below.  The gang kernel splits the two jobs over wavefronts (see there).
//  2. dtw_traceback_kernel: one thread per pair walks the back-pointers from (N-1, M-1) and writes the
//     path right-aligned into its output row.
// The sweep is a dependency chain (N + M - 1 sequential steps per band), the distances are throughput
// code: parallelism comes from thousands of pairs side by side.
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

// back-pointers: two bits per cell, (up < diag) << 1 | (left < min(diag, up)); first minimum in the order diag, up, left

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
    float* ynorm;                // gang kernel: [slot][mcap] norms of the pair's token-2 rows, computed by band 0
    int32_t* done;               // gang kernel<true>: [pair] set once every back-pointer of the pair has left the CU (the overlapped traceback polls it)
};

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

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
                const double cost = (double)dist + best;
                const bool on = rowok && (uint32_t)j < (uint32_t)M;
                p2 = left;
                p1 = on ? cost : left;                                     // past the row's end the last cost stays put
                bits = (bits << 2) | (take_up ? 2u : 0u) | (take_left ? 1u : 0u);    // cells outside the matrix: never read
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
// Gang form for 40-value frames (round 4; the default): a workgroup is THREE wavefronts around
// two slots -- one producer wavefront PER SLOT and one consumer wavefront sweeping both.
//   producer q   keeps its band's 32 rows of token 1 as MFMA fragments + norms in REGISTERS for
//                the whole band (they are read from HBM once per band, not once per round), loads
//                32 rows of token 2 per round, 20 MFMAs, the reference's distance on 16 cells per
//                lane, and stores the block COLUMN-MAJOR into one of three 4 KB buffers of its
//                slot: blk[slot][round % 3][column][row].  No diagonal skew on this side, no cells
//                held back: one workgroup barrier per round.
//   consumer     lane n of a half owns row n of the slot's band; at step e of round u its cell is
//                column 32 u + e - n: block u for e >= n, block u - 1 (the buffer of the round
//                before) for e < n -- one select of two per-lane base addresses, the rest of the
//                address is the step's immediate offset.  A step is ~20 vector instructions: the
//                up neighbour arrives over ONE DPP shift (the diagonal neighbour is last step's up),
//                min3 with the oracle's tie-break is two v_min_f64 + two compares whose lane
//                masks are shifted straight into the back-pointer word (v_addc), lane 31 stores the
//                band's boundary row as it goes.
// Back-pointers: 2 bits per cell, (up < diag) << 1 | (left < min(diag, up)), sixteen diagonals to a
// dword, the FIRST of the sixteen in the top bits; cells outside the matrix hold garbage (never
// visited by the traceback).
// ---------------------------------------------------------------------------------------
#ifdef ABN_DTW_STAMPS         // diagnostic build only (tools/dtw_stamps.py): cycles per phase, summed over workgroups
__device__ unsigned long long g_dtw_cycles[16];
#define PSTAMP(k) do { __builtin_amdgcn_sched_barrier(0); { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc[(k) & 3] += t_ - tlast; tlast = t_; } __builtin_amdgcn_sched_barrier(0); } while (0)
#define PSTAMP_INIT unsigned long long tacc[4] = {0, 0, 0, 0}; unsigned long long tlast = __builtin_amdgcn_s_memtime()
#define PSTAMP_FLUSH(base) do { if (lane == 0) for (int k_ = 0; k_ < 4; ++k_) atomicAdd(&g_dtw_cycles[(base) + k_], tacc[k_]); } while (0)
#else
#define PSTAMP(k) do {} while (0)
#define PSTAMP_INIT do {} while (0)
#define PSTAMP_FLUSH(base) do {} while (0)
#endif
constexpr int GS = 2;                                  // slots per workgroup
struct GangDesc {                                      // what a slot does in one round (LDS; written by its producer, or by the dealer)
    int32_t pair;                                      // -1: slot idle
    int32_t N, M, nbands, nrounds, band, u, pad;
    int64_t dir_off;
    // the dealt schedule (SCHED = 1)
    int64_t xoff, yoff;                                // float offsets of the pair's tokens
    int32_t set;                                       // which of the gang's two sets of boundary rows / norm strips the pair uses
    int32_t dpp;                                       // this band's upper neighbour is swept by the OTHER half, one round ahead: lane 32 takes its
                                                       // row above over the wavefront shift, nothing comes from memory
    int32_t feed;                                      // the band's last row goes to memory for the band below
    int32_t final;                                     // no slot has work in this round or any later one
};

// mn = min(a, b); bits = 2 bits + (a < b)
__device__ __forceinline__ void min_lt(double a, double b, double& mn, uint32_t& bits)
{
    uint32_t nb;
    asm("v_cmp_lt_f64_e32 vcc, %3, %4\n\t"
        "v_addc_co_u32_e32 %1, vcc, %2, %2, vcc\n\t"
        "v_min_f64 %0, %3, %4"
        : "=v"(mn), "=&v"(nb)
        : "v"(bits), "v"(a), "v"(b)
        : "vcc");
    bits = nb;
}

__device__ __forceinline__ double dpp_shr1_f64(double v)      // lane l <- lane l - 1 (lanes 0 / 32: overridden by the caller)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

#ifndef ABN_GANG_OCC
#define ABN_GANG_OCC 4
#endif
#ifndef ABN_GANG_YPF
#define ABN_GANG_YPF 1       // the producer requests its rows a round ahead
#endif
// FLAGS: the traceback of a pair may start while this kernel still runs (abn_dtw_batched_overlap: a second launch on
// another stream polls P.done).  What that launch reads of a pair -- back-pointers, bad flag, total cost -- is then
// stored write-through (agent-scope stores: they leave the XCD's L2 for the level all XCDs share, which is where these
// bytes are headed anyway: nothing in this kernel reads them again), the storing wavefront waits until they are
// acknowledged, and only then sets the pair's flag (MI355X_MICROARCH.md, hand-offs: sc1 stores -> vmcnt(0) -> sc1 flag).
#ifndef ABN_GANG_RELEASE
#define ABN_GANG_RELEASE 0
#endif
#ifndef ABN_GANG_NT
#define ABN_GANG_NT 0
#endif
// (Two gangs per workgroup -- six wavefronts that land as 2 producers + 1 consumer on every SIMD, where five one-gang
// workgroups land as 3P+C, 3P+C, 2P+2C, 2P+C -- were measured in round 6: twelve wavefronts per CU instead of fifteen cost 12 %.)
// SCHED = 1 (round 6, the default): the gang's two slots are DEALT the bands of a stream of pairs in turn -- band b of a
// pair to slot 0, band b + 1 to slot 1 ONE ROUND LATER, b + 2 to slot 0 when it is free again ... -- by wavefront 0, three
// rounds ahead of the sweep (desc[round % 4]).  A band whose upper neighbour runs in the other half one round ahead is
// coupled to it inside the consumer wavefront: the two halves are then one 64-row band on the skewed diagonal, lane 32's
// row above arrives over the same wavefront shift as everybody's, no boundary row goes through memory -- and the lower
// band's producer asks for token 2's rows ONE round after the upper band's did: the XCD's L2 still has them, so token 2
// crosses the fabric once per 64 rows instead of once per 32 (6.8 -> ~4 B / cell).  Every other band takes its row above
// from the gang's boundary rows as before, at least three rounds behind the band that writes them.  Nothing idles: when a
// pair has an odd band left, the other slot starts the next pair.
// SCHED = 0: round 4's schedule, every slot walks a pair of its own (ABN_DTW_SCHED=0).
template <bool FLAGS, int SCHED>
__global__ __launch_bounds__(64 * (GS + 1)) __attribute__((amdgpu_waves_per_eu(ABN_GANG_OCC, ABN_GANG_OCC))) void dtw_gang_kernel(DtwP P)
{
#ifdef ABN_EXP_HALFSYNC      // (measurement only, RACY: a workgroup barrier every other round -- what looser coupling of the waves could buy)
    constexpr int DR = SCHED ? 8 : 2;
#ifndef ABN_EXP_SYNCMASK
#define ABN_EXP_SYNCMASK 1
#endif
#define GANG_ROUND_SYNC(t) do { if (((t) & ABN_EXP_SYNCMASK) == ABN_EXP_SYNCMASK) __syncthreads(); } while (0)
#else
    constexpr int DR = SCHED ? 4 : 2;                                            // rounds of descriptors alive at once
#define GANG_ROUND_SYNC(t) __syncthreads()
#endif
    __shared__ __attribute__((aligned(16))) float blk[GS][3][BAND][BAND];      // [slot][round % 3][column][row]
    __shared__ __attribute__((aligned(16))) float ny_s[2][GS][BAND];           // [round parity][slot]
    __shared__ double top_s[GS][BAND];
    __shared__ GangDesc desc[DR][GS];                                            // [round % DR][slot]
    const int gid = (int)blockIdx.x;
    auto all_idle = [&](int par) { return desc[par][0].pair < 0 && desc[par][1].pair < 0; };
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5, n = lane & 31;
    const int D = KCH;
    const int64_t bstride = P.mcap;                     // a boundary row: 32 doubles of slack in front (the host pads mcap)

    if (wave < GS) {
        // =========================== producer of slot `wave` ===========================
        const int q = wave;
        bool exhausted = false;
        int pair = -1, N = 0, M = 0, nbands = 0, nrounds = 0, band = 0, u = 0;
        int64_t xoff = 0, yoff = 0, dir_off = 0;
        int set = q, dpp = 0;                           // SCHED = 1: the pair's set of boundary rows / norm strip; coupled to the other half
        float xf[KST], nx = 1.0f;
        uint32_t orbits = 0u;                           // OR of the distances' bit patterns: >= 0x7f800000 iff one is NaN
        auto fetch = [&]() {                            // next pair of the queue (wave-uniform values)
            pair = -1;
            if (exhausted) return;
            int idx = 0;
            if (lane == 0) idx = atomicAdd(P.counter, 1);
            idx = __builtin_amdgcn_readfirstlane(idx);
            if (idx >= P.npairs) { exhausted = true; return; }
            const int p = __builtin_amdgcn_readfirstlane(P.order[idx]);
            const PairMeta* m = P.meta + p;
            pair = p;
            N = __builtin_amdgcn_readfirstlane(m->n1); M = __builtin_amdgcn_readfirstlane(m->n2);
            nbands = __builtin_amdgcn_readfirstlane(m->nbands); nrounds = __builtin_amdgcn_readfirstlane(m->nrounds);
            xoff = readlane64(m->off1, 0) * D; yoff = readlane64(m->off2, 0) * D; dir_off = readlane64(m->dir_off, 0);
            band = 0; u = 0;
        };
        auto load_x = [&]() {                           // the band's rows: fragments + norms, kept for the whole band
            const float* xrow = P.feats1 + xoff + (int64_t)min(band * BAND + n, N - 1) * D;
            nx = load_row40(xf, xrow, half);
        };
        auto publish = [&](int par) {
            if (lane == 0) {
                GangDesc d;
                d.pair = pair; d.N = N; d.M = M; d.nbands = nbands; d.nrounds = nrounds; d.band = band; d.u = u; d.pad = 0;
                d.dir_off = dir_off;
                desc[par][q] = d;
            }
        };
        // The 32 rows of token 2 of round (band, u), requested one round AHEAD: raw buffer loads (the
        // compiler leaves them where they are written; plain loads sink to their first use), the
        // descriptor spans the pair's token so offsets fit 32 bits.
        // Band 0 of a pair takes whole rows (both halves of a lane pair load the same 160 bytes: the norm
        // needs every value) and leaves the norms in ynorm[]; the bands below it take the norms from
        // there and load each row 4 h bytes further on, so that a piece's .x / .z are the fragment values
        // of BOTH halves -- no selects, no sums (the last piece's fourth value falls behind the row: never
        // used, and behind the token the descriptor returns 0).
        u32x4 yq[10];
        float ynq = 1.0f;
        float* const ynorm_base = P.ynorm + (int64_t)(GS * gid) * P.mcap;      // the gang's two strips (SCHED = 0: one per slot; 1: one per pair in flight)
        auto request_y = [&]() {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.feats2 + yoff), 0, M * (D * 4), 0x00020000);
            const int row = min(u * BAND + n, M - 1);
            const int vo = row * (D * 4) + (band > 0 ? 4 * half : 0);
#pragma unroll
            for (int i = 0; i < 10; ++i) yq[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 16 * i, 0);
            // (a coupled band takes the norms from its upper neighbour's LDS copy of the round before: that band may have
            // computed them only a round ago)
            if (band > 0 && !dpp) ynq = __hip_atomic_load(&ynorm_base[(int64_t)set * P.mcap + row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        // the block of (band, u) into buffer `buf` from the rows requested before; par: the round's parity (ny_s)
        auto produce = [&](int buf, int par) {
            f32x16 acc;
            float ny;
            {
                // row piece by row piece: two fragment values -> two MFMAs; in band 0 also four squares ->
                // numpy's eight partial sums (sumsq40's order; packed: two squares / two sums per instruction);
                // a piece's registers are free once it is used
                auto chain = [&](int i, float f0, float f1) {
#ifndef ABN_EXP_NOMFMA
                    if (i == 0) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f0, xf[0], f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0);
                    else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f0, xf[2 * i], acc, 0, 0, 0);       // A = token 2 rows (j), B = token 1 rows (i)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f1, xf[2 * i + 1], acc, 0, 0, 0);
#else
                    if (i == 0) for (int c = 0; c < 16; ++c) acc[c] = 0.0f;
                    acc[i] += f0 * xf[2 * i] + f1 * xf[2 * i + 1];
#endif
                };
                if (band == 0) {
                    f32x2 r01, r23, r45, r67;
#pragma unroll
                    for (int i = 0; i < 10; ++i) {
                        const float4 v = make_float4(__uint_as_float(yq[i].x), __uint_as_float(yq[i].y), __uint_as_float(yq[i].z), __uint_as_float(yq[i].w));
                        chain(i, half ? v.y : v.x, half ? v.w : v.z);
                        const f32x2 sxy = f32x2{v.x, v.y} * f32x2{v.x, v.y}, szw = f32x2{v.z, v.w} * f32x2{v.z, v.w};
                        if (i == 0) { r01 = sxy; r23 = szw; }
                        else if (i == 1) { r45 = sxy; r67 = szw; }
                        else if ((i & 1) == 0) { r01 += sxy; r23 += szw; }
                        else { r45 += sxy; r67 += szw; }
                    }
                    ny = sqrtf(((r01.x + r01.y) + (r23.x + r23.y)) + ((r45.x + r45.y) + (r67.x + r67.y)));
                    if (nbands > 1 && half == 0 && u * BAND + n < M)
                        __hip_atomic_store(&ynorm_base[(int64_t)set * P.mcap + u * BAND + n], ny, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
#pragma unroll
                    for (int i = 0; i < 10; ++i) chain(i, __uint_as_float(yq[i].x), __uint_as_float(yq[i].z));
                    ny = dpp ? ny_s[par ^ 1][q ^ 1][n] : ynq;      // (coupled: the same rows, a round ago, in the other slot)
                }
            }
            if (half == 0) ny_s[par][q][n] = ny;
            wave_lds_sync();                                               // ny_s (this wave's own writes)
            const bool plain = __all(norm_is_plain(nx) && norm_is_plain(ny));
            // accumulator c of lane (n, h) is column m = (c & 3) + 8 (c >> 2) + 4 h of the block, row n
            float* const out = &blk[q][buf][4 * half][n];
            uint32_t ob = 0u;
            auto epilogue = [&](auto pl) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (u * BAND + 8 * g >= M) break;                       // columns past the token's end: never read by the sweep
                    const float4 ny4 = *reinterpret_cast<const float4*>(&ny_s[par][q][8 * g + 4 * half]);
                    const float nyv[4] = {ny4.x, ny4.y, ny4.z, ny4.w};
                    if constexpr (decltype(pl)::value) {          // two cells per instruction (dist_ref.h)
#if !defined(ABN_EXP_NOEPI) && !defined(ABN_EXP_NOFAST)
                        // the four quotients first; where all 256 of the wavefront sit in acosf's first range
                        // (unrelated frames: most groups) the straight-line statements of that range alone
                        const f32x4 c4 = div_normal4(f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]},
                                                     splat4(nx) * f32x4{nyv[0], nyv[1], nyv[2], nyv[3]});
                        const f32x2 c01 = f32x2{c4.x, c4.y}, c23 = f32x2{c4.z, c4.w};
                        f32x2 d01, d23;
                        if (quotients_small4_all(c01, c23)) {
                            const f32x4 d4 = div_pi4(acosf_small4(c4));
                            d01 = f32x2{d4.x, d4.y};
                            d23 = f32x2{d4.z, d4.w};
                        } else {
                            d01 = div_pi2(acosf_ref2(c01));
                            d23 = div_pi2(acosf_ref2(c23));
                        }
                        ob |= (__float_as_uint(d01.x) | __float_as_uint(d01.y)) | (__float_as_uint(d23.x) | __float_as_uint(d23.y));
                        out[(8 * g) * BAND] = d01.x;                   // padded rows / columns repeat real ones: no masking needed
                        out[(8 * g + 1) * BAND] = d01.y;
                        out[(8 * g + 2) * BAND] = d23.x;
                        out[(8 * g + 3) * BAND] = d23.y;
#else
#pragma unroll
                        for (int e = 0; e < 4; e += 2) {
#ifndef ABN_EXP_NOEPI
                            const f32x2 dv = angular_distance_plain2(f32x2{acc[4 * g + e], acc[4 * g + e + 1]}, nx, f32x2{nyv[e], nyv[e + 1]});
#else
                            const f32x2 dv = f32x2{acc[4 * g + e], acc[4 * g + e + 1]} * splat2(nx * 1e-3f) * f32x2{nyv[e], nyv[e + 1]};
#endif
                            ob |= __float_as_uint(dv.x) | __float_as_uint(dv.y);      // padded rows / columns repeat real ones: no masking needed
                            out[(8 * g + e) * BAND] = dv.x;
                            out[(8 * g + e + 1) * BAND] = dv.y;
                        }
#endif
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float dv = angular_distance_ref<false>(acc[4 * g + e], nx, nyv[e]);
                            ob |= __float_as_uint(dv);
                            out[(8 * g + e) * BAND] = dv;
                        }
                    }
                }
            };
            if (plain) epilogue(std::true_type{}); else epilogue(std::false_type{});
            orbits |= ob;
            if ((SCHED || band + 1 == nbands) && (u + 1) * BAND >= M) {    // the pair's last block (SCHED = 1: this band's -- a pair's bands are spread over both producers)
                const bool isbad = __any(orbits >= 0x7f800000u);
                if (isbad) {                                                // utils.py:59: NaN (or negative) distance
                    if (FLAGS) {                                            // in place before this round's barrier, behind which the consumer flags the pair
                        if (lane == 0) __hip_atomic_store(&P.bad[pair], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    } else if (lane == 0) P.bad[pair] = 1;
                }
                orbits = 0u;
            }
        };

        if constexpr (SCHED == 0) {
        // The wave's state is one round ahead of what it publishes: while the block of round t + 1 is
        // computed, the rows of round t + 2 are already requested (they arrive under the epilogue and the
        // barrier).  Iteration t = -1 is the prologue: the first pair's first block, complete before the
        // consumer starts.
        fetch();
        if (pair >= 0) {
            load_x();
#if ABN_GANG_YPF
            request_y();
#endif
        }
        PSTAMP_INIT;
        for (int t = -1;; ++t) {
            if (t >= 0 && all_idle(t & 1)) break;
            publish((t + 1) & 1);                                           // round t + 1 of this slot
            const bool have = pair >= 0 && u * BAND < M;
            PSTAMP(0);
#if !ABN_GANG_YPF
            if (have) request_y();
#endif
            if (have) produce((t + 1) % 3, 0);
            PSTAMP(1);
            bool newband = false;
            if (pair >= 0) {                                                // on to round t + 2
                if (u + 1 < nrounds) ++u;
                else if (band + 1 < nbands) { ++band; u = 0; newband = true; }
                else { fetch(); newband = pair >= 0; }
            }
            if (newband) load_x();
#if ABN_GANG_YPF
            if (pair >= 0 && u * BAND < M) request_y();
#endif
            PSTAMP(2);
            __syncthreads();                                                // round t + 1 is complete; the consumer has finished round t
            PSTAMP(3);
        }
        } else {
        // ---- the dealer (wavefront 0, scalar work): the schedule of round r for both slots -> desc[r % 4].  Its state lives in
        // LDS (read and written by this wavefront alone, once per round): in registers it cost the producer 48 spilled dwords
        struct Slot { int pair, N, M, nb, nr, band, start, set, dpp, pad; int64_t x, y, dir; };
        struct Dealer {
            int exh, seq, turn;
            int c_pair, c_N, c_M, c_nb, c_nr, c_set, c_next, c_last;       // the pair being dealt
            int64_t c_x, c_y, c_dir;
            Slot s0, s1;
        };
        __shared__ Dealer dl;
        if (q == 0 && lane == 0) {
            dl.exh = 0; dl.seq = 0; dl.turn = 0; dl.c_pair = -1; dl.c_nb = 0; dl.c_next = 0; dl.c_last = -1000;
            dl.s0.pair = -1; dl.s1.pair = -1;
        }
        wave_lds_sync();
#define d_exh dl.exh
#define d_seq dl.seq
#define d_turn dl.turn
#define c_pair dl.c_pair
#define c_N dl.c_N
#define c_M dl.c_M
#define c_nb dl.c_nb
#define c_nr dl.c_nr
#define c_set dl.c_set
#define c_next dl.c_next
#define c_last dl.c_last
#define c_x dl.c_x
#define c_y dl.c_y
#define c_dir dl.c_dir
#define s0 dl.s0
#define s1 dl.s1
        auto deal = [&](int r) {
            if (s0.pair >= 0 && r >= s0.start + s0.nr) s0.pair = -1;
            if (s1.pair >= 0 && r >= s1.start + s1.nr) s1.pair = -1;
            // the slot whose turn it is takes the stream's next band, if it is free and the band may start (twice: both may start in one round)
            auto try_assign = [&]() -> bool {
                Slot& me = d_turn ? s1 : s0;
                if (me.pair >= 0) return false;
                if (c_pair < 0 || c_next >= c_nb) {                         // the stream's next pair
                    c_pair = -1;
                    if (d_exh) return false;
                    int idx = 0;
                    if (lane == 0) idx = atomicAdd(P.counter, 1);
                    idx = __builtin_amdgcn_readfirstlane(idx);
                    if (idx >= P.npairs) { d_exh = true; return false; }
                    const int p = __builtin_amdgcn_readfirstlane(P.order[idx]);
                    const PairMeta* m = P.meta + p;
                    c_pair = p;
                    c_N = __builtin_amdgcn_readfirstlane(m->n1); c_M = __builtin_amdgcn_readfirstlane(m->n2);
                    c_nb = __builtin_amdgcn_readfirstlane(m->nbands); c_nr = __builtin_amdgcn_readfirstlane(m->nrounds);
                    c_x = readlane64(m->off1, 0) * D; c_y = readlane64(m->off2, 0) * D; c_dir = readlane64(m->dir_off, 0);
                    c_set = d_seq & 1; ++d_seq;                             // (at most two pairs are in flight, consecutive ones)
                    c_next = 0; c_last = -1000;
                }
                // a band may start one round behind its upper neighbour if that one runs in slot 0 and this one goes to
                // slot 1 (coupled inside the consumer wavefront), else three rounds behind it (through the boundary rows)
                bool couple = false;
                if (c_next > 0) {
                    couple = d_turn == 1 && r - c_last == 1 && s0.pair == c_pair && s0.band == c_next - 1;
                    if (!couple && r - c_last < 3) return false;
                }
                me.pair = c_pair; me.N = c_N; me.M = c_M; me.nb = c_nb; me.nr = c_nr; me.band = c_next; me.start = r;
                me.set = c_set; me.dpp = couple ? 1 : 0; me.x = c_x; me.y = c_y; me.dir = c_dir;
                c_last = r; ++c_next; d_turn ^= 1;
                return true;
            };
            if (try_assign()) (void)try_assign();
            const bool fin = d_exh && s0.pair < 0 && s1.pair < 0 && (c_pair < 0 || c_next >= c_nb);
            if (lane == 0) {
                auto put = [&](const Slot& me, int k, bool lower_coupled) {
                    GangDesc d;
                    d.pair = me.pair; d.N = me.N; d.M = me.M; d.nbands = me.nb; d.nrounds = me.nr; d.band = me.band;
                    d.u = r - me.start; d.pad = 0; d.dir_off = me.dir; d.xoff = me.x; d.yoff = me.y; d.set = me.set; d.dpp = me.dpp;
                    // (the last row goes to memory unless the band below is coupled to this one: known from the band's second round on)
                    d.feed = me.band + 1 < me.nb && !lower_coupled ? 1 : 0;
                    d.final = fin ? 1 : 0;
                    desc[r & (DR - 1)][k] = d;
                };
                put(s0, 0, s1.pair == s0.pair && s1.band == s0.band + 1 && s1.dpp != 0);
                put(s1, 1, false);
            }
            wave_lds_sync();
        };
#undef d_exh
#undef d_seq
#undef d_turn
#undef c_pair
#undef c_N
#undef c_M
#undef c_nb
#undef c_nr
#undef c_set
#undef c_next
#undef c_last
#undef c_x
#undef c_y
#undef c_dir
#undef s0
#undef s1
        // this slot's context of a round, from its descriptor
        auto take = [&](const GangDesc& d) {      // (wave-uniform values: scalar registers, as the round-4 producer's own state was)
            pair = __builtin_amdgcn_readfirstlane(d.pair); N = __builtin_amdgcn_readfirstlane(d.N); M = __builtin_amdgcn_readfirstlane(d.M);
            nbands = __builtin_amdgcn_readfirstlane(d.nbands); band = __builtin_amdgcn_readfirstlane(d.band); u = __builtin_amdgcn_readfirstlane(d.u);
            xoff = readlane64(d.xoff, 0); yoff = readlane64(d.yoff, 0);
            set = __builtin_amdgcn_readfirstlane(d.set); dpp = __builtin_amdgcn_readfirstlane(d.dpp);
        };
        // Round r's descriptors are written three iterations before the sweep reads them: iteration t produces the blocks
        // of round t + 1 (rows asked for at iteration t - 1) and asks for the rows of round t + 2, whose band's rows of
        // token 1 are fetched behind the old band's last block.
        if (q == 0) { deal(0); deal(1); }
        __syncthreads();
        take(desc[0][q]);
        int held_pair = -1, held_band = -1;                                 // what xf / nx hold
        if (pair >= 0) { load_x(); held_pair = pair; held_band = band; if (u * BAND < M) request_y(); }
        for (int t = -1;; ++t) {
            if (t >= 0 && desc[t & (DR - 1)][0].final) break;
            if (q == 0) deal(t + 3);
            take(desc[(t + 1) & (DR - 1)][q]);
            if (pair >= 0 && u * BAND < M) produce((t + 1) % 3, (t + 1) & 1);
            take(desc[(t + 2) & (DR - 1)][q]);
            if (pair >= 0) {
                if (pair != held_pair || band != held_band) { load_x(); held_pair = pair; held_band = band; }
                if (u * BAND < M) request_y();
            }
            GANG_ROUND_SYNC(t);                                             // round t + 1 is complete; the consumer has finished round t
        }
        }
        PSTAMP_FLUSH(0);
    } else {
        // =========================== consumer ===========================
#ifndef ABN_GANG_CPRIO
#define ABN_GANG_CPRIO 0
#endif
        if (ABN_GANG_CPRIO) __builtin_amdgcn_s_setprio(ABN_GANG_CPRIO);       // the sweep is the workgroup's dependency chain
        const double INF = __builtin_inf();
        double* const bnd_slot = P.bound + (int64_t)(GS * gid + half) * 2 * bstride + 32;      // (SCHED = 0: a slot's own two rows)
        double p1 = INF, upprev = INF;
        uint32_t bits = 0u;
        // the boundary values of the NEXT round, requested while this one is swept (same band: the row
        // was finished a band ago); `pf_key` says which (pair, band, round) they belong to
        double pf_top = INF;
        int pf_pair = -1, pf_band = -1, pf_u = -1;
        constexpr int PF = 4;                                               // LDS reads in flight ahead of the step that uses them
        if (SCHED) __syncthreads();                                         // (the dealer's first two rounds)
        __syncthreads();                                                    // the first blocks are in place
        PSTAMP_INIT;
        for (int t = 0;; ++t) {
            if (SCHED ? desc[t & (DR - 1)][0].final != 0 : all_idle(t & 1)) break;
            const GangDesc& d = desc[t & (DR - 1)][half];
            // SCHED = 1: the pair's set of boundary rows (two pairs at most are in flight in a gang); coupled = this half's row
            // above is the other half's lane 31, one round ahead on the same skewed diagonal
            double* const bnd = SCHED ? P.bound + (int64_t)(GS * gid + d.set) * 2 * bstride + 32 : bnd_slot;
            const bool coupled = SCHED && d.dpp != 0;
            const int pair = d.pair, N = d.N, M = d.M, nbands = d.nbands, nrounds = d.nrounds, band = d.band, u = d.u;
            const bool active = pair >= 0;
            if (active && u == 0) {                      // a band starts: fresh diagonals
                p1 = INF;
                upprev = (band == 0 && n == 0) ? 0.0 : INF;      // a pair starts from the virtual cell (-1, -1)
            }
            const int i0 = band * BAND, j0 = u * BAND;
            const double* const bin = bnd + (int64_t)((band & 1) ^ 1) * bstride;
            double topv = INF;
            if (active && band > 0 && !coupled && j0 + n < M) {
                const bool hit = pf_pair == pair && pf_band == band && pf_u == u;
                topv = pf_top;
                if (!hit) topv = __hip_atomic_load(&bin[j0 + n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            top_s[half][n] = topv;
            wave_lds_sync();
            PSTAMP(8);
            if (active && band > 0 && !coupled && u + 1 < nrounds) {        // stays inside the padded row (plan_ws)
                pf_top = __hip_atomic_load(&bin[j0 + BAND + n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                pf_pair = pair; pf_band = band; pf_u = u + 1;
            }
            const int cur = t % 3, prev = (t + 2) % 3;
            const float* const acur = &blk[half][cur][0][n] - n * BAND;          // + e * BAND: column e - n of this round's block
            const float* const aprev = &blk[half][prev][BAND - 1][n] - (n - 1) * BAND;   // + e * BAND: column 32 + e - n of the block before
            const double* const tops = top_s[half];
            const bool rowok = active && i0 + n < N;
            const bool feed = active && (SCHED ? d.feed != 0 : band + 1 < nbands) && n == BAND - 1;     // the band's last row feeds the next band
            double* const bout = bnd + (int64_t)(band & 1) * bstride + (j0 - (BAND - 1));
            const int jb = j0 - n;
            const bool take_top = n == 0 && !coupled;           // (a coupled band's first row takes lane 31's cost: the shift crosses the halves)
            uint32_t* const dptr = P.dirs + d.dir_off + ((int64_t)(band * 2 * nrounds + 2 * u) * BAND + n);
            float dq[PF];
            double tq[PF];
#pragma unroll
            for (int e = 0; e < PF; ++e) {
                dq[e] = (n > e ? aprev : acur)[e * BAND];
                tq[e] = tops[e];
            }
            // A round is INSIDE the matrix when every cell it sweeps exists: all 32 rows of the band are rows of token 1
            // and the 63 columns the skewed block spans are columns of token 2.  Most rounds of most pairs are; when both
            // halves' are, the step drops the cell's in-matrix test and the select that keeps the cost of a row's last cell.
            const bool inside = active && i0 + BAND <= N && j0 >= BAND - 1 && j0 + BAND <= M;
            auto sweep = [&](auto edge) {
#ifdef ABN_EXP_NOSWEEP
                for (int e = 0; e < 0; ++e) {
#else
#pragma unroll
                for (int e = 0; e < BAND; ++e) {
#endif
                    const float dist = dq[e % PF];
                    const double topc = tq[e % PF];
                    if (e + PF < BAND) {
                        dq[e % PF] = (n > e + PF ? aprev : acur)[(e + PF) * BAND];
                        tq[e % PF] = tops[e + PF];
                    }
                    double up = dpp_shr1_f64(p1);
                    up = take_top ? topc : up;
                    double b1, best;
                    min_lt(up, upprev, b1, bits);            // up < diag
                    min_lt(p1, b1, best, bits);              // left < min(diag, up)
                    const double cost = (double)dist + best;
                    upprev = up;
                    if constexpr (decltype(edge)::value) {
                        const bool on = rowok && (uint32_t)(jb + e) < (uint32_t)M;
                        p1 = on ? cost : p1;                 // past the row's end the last cost stays put
                    } else {
                        p1 = cost;
                    }
                    if (feed) __hip_atomic_store(&bout[e], cost, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if ((e & 15) == 15 && (!decltype(edge)::value || active)) {
                        if (FLAGS && !ABN_GANG_RELEASE) __hip_atomic_store(&dptr[(e >> 4) * BAND], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        else if (ABN_GANG_NT) __builtin_nontemporal_store(bits, &dptr[(e >> 4) * BAND]);      // (streaming: written once, read by the traceback)
                        else dptr[(e >> 4) * BAND] = bits;
                    }
                }
            };
#ifndef ABN_EXP_NOINSIDE
            if (__all(inside)) sweep(std::false_type{}); else
#endif
            sweep(std::true_type{});
            const bool fin = active && u + 1 == nrounds && band + 1 == nbands;      // the pair's last round
            if (fin && P.total_cost && n == ((N - 1) & 31)) {                        // lane (N-1) % 32 holds cost(N-1, M-1)
                if (FLAGS) __hip_atomic_store(&P.total_cost[pair], p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else P.total_cost[pair] = p1;
            }
            if (FLAGS && __any(fin)) {
                if (ABN_GANG_RELEASE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // every store of this wavefront is acknowledged
                if (fin && n == 0) __hip_atomic_store(&P.done[pair], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // a finished band's boundary row is re-read by this wavefront (workgroup scope: the CU's own L1 /
            // L2 path, no write-through to memory): the stores only have to be complete
            if (__any(active && u + 1 == nrounds)) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_s_waitcnt(0);
            }
            PSTAMP(9);
            if (SCHED) GANG_ROUND_SYNC(t); else __syncthreads();
            PSTAMP(10);
        }
        PSTAMP_FLUSH(8);
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
#ifndef ABN_TB_ROWS
#define ABN_TB_ROWS 8
#endif
constexpr int TB_ROWS = ABN_TB_ROWS;
#ifndef TB_PLAIN_LOADS
#define TB_PLAIN_LOADS 0
#endif
// MODE 0: every pair of the queue, after the fill kernel (one stream).
// MODE 1: launched on a SECOND stream beside dtw_gang_kernel<true>: a wavefront waits -- a bounded poll of the pairs' done
//         flags, asleep in between -- until its 64 pairs (neighbours in the queue: filled at about the same time) are
//         complete, then walks them; what the fill kernel stored write-through is read with agent-scope loads.  A pair
//         whose flag has not come by the time limit is left alone (path_len stays 0).
// MODE 2: the sweep behind both streams: the pairs MODE 1 left (normally none).
#ifndef TB_SIDE_LANES
#define TB_SIDE_LANES 64
#endif
#ifndef ABN_TB_WAIT_TICKS
#define ABN_TB_WAIT_TICKS 50000000ull      // 0.5 s of the 100 MHz clock
#endif
#ifdef ABN_TB_STAMPS          // diagnostic build only (tools/tb_stamps.py): when each wavefront of the overlapped traceback started, saw its flags, ended
__device__ unsigned long long g_tb_stamps[4096][4];
#endif
template <int MODE>
__global__ __launch_bounds__(64) void dtw_traceback_kernel(const PairMeta* __restrict__ meta, const int32_t* __restrict__ order,
                                                           int npairs, int ntotal, const uint32_t* dirs,
                                                           const int32_t* bad, int32_t* done,
                                                           int32_t* path1, int32_t* path2,
                                                           int32_t* path_len, int64_t path_stride,
                                                           double* total_cost)
{
#ifdef ABN_EXP_NOTB          // (measurement only: the fill alone; a racy fill's back-pointers must not be walked)
    return;
#endif
    __shared__ uint32_t win[2 * TB_ROWS][64];            // [group offset * 8 + row offset][thread]: conflict-free
    const int lane = threadIdx.x;
    // MODE 1: TB_SIDE_LANES pairs per wavefront (a wavefront starts when ALL its pairs are filled and walks as long as its
    // longest path: fewer pairs per wavefront start earlier and end earlier behind the fill's last pairs)
    constexpr int PER = MODE == 1 ? TB_SIDE_LANES : 64;
    if (lane >= PER) return;
    const int idx = (int)(blockIdx.x * PER + lane);
    // queue order (pairs of similar size side by side): the threads of a wavefront walk paths of similar length.  Behind
    // the npairs queued pairs the list holds the pairs with an empty token: nothing to walk, path_len = 0 (MODE 0 / 2: their
    // grids span the whole list; every pair's path_len and total_cost are written by exactly one of the launches, no memset)
    if (idx >= ntotal || (MODE == 1 && idx >= npairs)) return;
    const int p = order[idx];
    if (idx >= npairs) {
        path_len[p] = 0;
        if (total_cost) total_cost[p] = 0.0;
        return;
    }
#ifdef ABN_TB_STAMPS
    if (MODE != 2 && lane == 0 && blockIdx.x < 4096) g_tb_stamps[blockIdx.x][MODE == 0 ? 1 : 0] = __builtin_amdgcn_s_memrealtime();
#endif
    if (MODE == 1) {
        int ok = 0;
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            if (!ok) ok = __hip_atomic_load(&done[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all(ok)) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > ABN_TB_WAIT_TICKS) break;
            __builtin_amdgcn_s_sleep(127);
        }
        if (!ok) return;                              // left to the MODE 2 launch
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        // the walk is a chain of dependent steps on a SIMD it shares with the fill kernel's (older) wavefronts
        __builtin_amdgcn_s_setprio(3);
#ifdef ABN_TB_STAMPS
        if (lane == 0 && blockIdx.x < 4096) g_tb_stamps[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
#endif
    }
    if (MODE == 2 && done[p] == 2) return;            // MODE 1 has been here
    const PairMeta m = meta[p];
    const int N = m.n1, M = m.n2;
    const int isbad = MODE == 1 ? __hip_atomic_load(&bad[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : bad[p];
    if (isbad) {                                      // the pair is dropped
        path_len[p] = 0;
        if (total_cost) total_cost[p] = 0.0;
        if (MODE == 1) done[p] = 2;
        return;
    }
    const uint32_t* dp = dirs + m.dir_off;
    const int nsg = 2 * m.nrounds;
    int32_t* o1 = path1 + (int64_t)p * path_stride + (path_stride - 1);
    int32_t* o2 = path2 + (int64_t)p * path_stride + (path_stride - 1);
    int i = N - 1, j = M - 1, k = 0;
    o1[0] = i; o2[0] = j;
    // (Round 6: a window's steps parked in LDS and stored only after the NEXT window's fetch has landed -- so that the
    // fetch is not counted in behind the ~30 scattered stores in front of it -- made the launch slower, 263 against 216 us:
    // the walk is a chain of ~40 dependent instructions per step, several of them lane-mask updates, not a wait for stores.)
    while (i > 0 || j > 0) {
        const int b = i >> 5, r = i & 31, g = (j + r) >> 4;
        const int rlo = max(r - (TB_ROWS - 1), 0);
        const uint32_t* src = dp + (int64_t)(b * nsg + g) * BAND + rlo;     // rows rlo .. rlo+7 <= 31 exist in every band
        uint32_t w0[TB_ROWS], w1[TB_ROWS];
#pragma unroll
        for (int q = 0; q < TB_ROWS; ++q) {
            if (MODE == 1 && !TB_PLAIN_LOADS) {
                w0[q] = __hip_atomic_load(src + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                w1[q] = g > 0 ? __hip_atomic_load(src + q - BAND, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            } else {
                w0[q] = src[q];
                w1[q] = g > 0 ? src[q - BAND] : 0u;
            }
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
            const uint32_t c = (w >> (30 - 2 * (s & 15))) & 3u;           // the first of a dword's sixteen diagonals sits in its top bits
            // left < min(diag, up): (0, -1); else up < diag: (-1, 0); else the diagonal: (-1, -1) -- as arithmetic, not as
            // three lane-mask branches in the middle of a dependent chain
            i -= (int)((c & 1u) ^ 1u);
            j -= (int)(c != 2u);
            ++k;
            o1[-k] = i;                                                   // (16-byte stores of four cells at a time were tried: 302 against 232 us)
            o2[-k] = j;
        }
    }
    path_len[p] = k + 1;
    if (MODE == 1) done[p] = 2;
#ifdef ABN_TB_STAMPS
    if (MODE != 2 && blockIdx.x < 4096) { atomicMax(&g_tb_stamps[blockIdx.x][2], (unsigned long long)__builtin_amdgcn_s_memrealtime()); g_tb_stamps[blockIdx.x][3] = k; }
#endif
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
        const float v = x[i];
        if ((over_pi & 2) && fabsf(v) < 0.5f && fabsf(v) > bits_f32(0x32800000u)) {
            // the gang kernel's straight-line form of the first range (what quotients_small4_all admits), one cell per lane
            const f32x4 a4 = acosf_small4(splat4(v));
            out[i] = (over_pi & 1) ? div_pi4(a4).x : a4.x;
            continue;
        }
        const float a = acosf_ref(v);
        out[i] = (over_pi & 1) ? div_pi(a) : a;
    }
}

struct WsPlan {
    int64_t meta_off, order_off, bad_off, done_off, counter_off, dirs_off, bound_off, ynorm_off, total;
    int64_t mcap;
    int32_t nwg;
};

}  // namespace abn

using namespace abn;

// Workspace: [PairMeta x P][order x P][bad x P][done x P][queue counter][back-pointers][boundary rows][norm strips]
// dwords: the pairs' back-pointer words; mmax: the longest token 2 (>= 1)
static WsPlan plan_ws_of(int64_t dwords, int64_t mmax, int64_t P)
{
    WsPlan w;
    // a boundary row: the longest token 2 rounded up, 32 doubles of slack in front (the gang kernel's
    // last lane writes columns -31 .. -1 of a band's first round there) and 32 behind
    w.mcap = align_up(mmax, 32) + 64;
    // persistent grid, two pairs in flight per workgroup: up to 9 single-wavefront workgroups per CU
    // (LDS) in the general kernel, 6 three-wavefront ones in the gang kernel; fewer when the boundary
    // rows of that many slots would outgrow 256 MiB
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
    w.done_off = take(P * 4);           // (between bad and counter: one memset clears all three)
    w.counter_off = take(4);
    w.dirs_off = take(dwords * 4);
    w.bound_off = take(2 * nwg * 2 * w.mcap * 8);
    w.ynorm_off = take(2 * nwg * w.mcap * 4);
    w.total = o;
    return w;
}

static WsPlan plan_ws(const int32_t* n1, const int32_t* n2, int64_t P)
{
    int64_t dwords = 0, mmax = 1;
    for (int64_t p = 0; p < P; ++p) {
        const int64_t a = n1[p] > 0 ? n1[p] : 0, b = n2[p] > 0 ? n2[p] : 0;
        if (a > 0 && b > 0) {
            dwords += ((a + BAND - 1) / BAND) * 2 * ((b + 2 * BAND - 1) / BAND) * BAND;
            mmax = b > mmax ? b : mmax;
        }
    }
    return plan_ws_of(dwords, mmax, P);
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

// the two events abn_dtw_batched_overlap orders its streams with: one pair per host thread and device, made at first use
static bool stream_order_events(hipEvent_t* a, hipEvent_t* b)
{
    struct Pair { hipEvent_t e[2] = {nullptr, nullptr}; };
    static thread_local Pair cache[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return false;
    Pair& c = cache[dev];
    for (int i = 0; i < 2; ++i)
        if (!c.e[i] && hipEventCreateWithFlags(&c.e[i], hipEventDisableTiming) != hipSuccess) { c.e[i] = nullptr; return false; }
    *a = c.e[0]; *b = c.e[1];
    return true;
}

static void launch_gang(bool flags, int64_t ng, hipStream_t st, const DtwP& P)
{
    const dim3 grid((unsigned)ng), block(64 * (GS + 1));
    const bool dealt = switches().dtw_dealt;
    if (dealt) {
        if (flags) hipLaunchKernelGGL((dtw_gang_kernel<true, 1>), grid, block, 0, st, P);
        else hipLaunchKernelGGL((dtw_gang_kernel<false, 1>), grid, block, 0, st, P);
    } else {
        if (flags) hipLaunchKernelGGL((dtw_gang_kernel<true, 0>), grid, block, 0, st, P);
        else hipLaunchKernelGGL((dtw_gang_kernel<false, 0>), grid, block, 0, st, P);
    }
}

static int dtw_batched_impl(const float* feats1, int64_t rows1, const float* feats2, int64_t rows2,
                            const int64_t* off1_host, const int32_t* n1_host, const int64_t* off2_host,
                            const int32_t* n2_host, int64_t npairs, int64_t D, int32_t* path1,
                            int32_t* path2, int32_t* path_len, int64_t path_stride, double* total_cost,
                            void* ws, int64_t ws_bytes, void* host_stage, int64_t host_stage_bytes,
                            void* stream, void* side_stream)
{
    ABN_REQUIRE(npairs >= 0 && npairs < (1LL << 31) && D >= 1 && D < (1 << 20), "dtw: bad npairs/D");
    if (npairs == 0) return ABN_OK;
    ABN_REQUIRE(feats1 && feats2 && off1_host && n1_host && off2_host && n2_host && path1 && path2 && path_len && ws &&
                    host_stage,
                "dtw: null pointer");
    const int64_t meta_bytes = align_up(npairs * (int64_t)sizeof(PairMeta), 256);
    if (host_stage_bytes < meta_bytes + align_up(npairs * 4, 256)) { set_error("dtw: host staging buffer too small"); return ABN_E_WORKSPACE; }
    hipStream_t st = (hipStream_t)stream, side = (hipStream_t)side_stream;
    char* base = (char*)ws;
    PairMeta* hm = (PairMeta*)host_stage;
    int32_t* hord = (int32_t*)((char*)host_stage + meta_bytes);
    // ONE pass over the pairs: validation, the pairs' metadata, the longest path, the longest token 2, and the histogram
    // of the work queue's counting sort (largest pairs first -- the short ones fill the tail --, empty pairs never queued;
    // the key is the number of rounds a pair needs, clamped: the order among giants is free)
    constexpr int NB = 4096;
    static thread_local std::vector<int32_t> count;
    count.assign(NB + 1, 0);
    auto bucket = [&](int64_t p) {
        const int64_t work = (int64_t)hm[p].nbands * hm[p].nrounds;
        return (int)(NB - 1 - (work < NB ? work : NB - 1));                           // descending
    };
    int64_t maxlen = 0, dwords = 0, mmax = 1;
    for (int64_t p = 0; p < npairs; ++p) {
        const int64_t a = n1_host[p], b = n2_host[p];
        ABN_REQUIRE(a >= 0 && b >= 0, "dtw: negative token length at pair %lld", (long long)p);
        ABN_REQUIRE(off1_host[p] >= 0 && off1_host[p] + a <= rows1 && off2_host[p] >= 0 && off2_host[p] + b <= rows2,
                    "dtw: pair %lld reads outside the feature arrays", (long long)p);
        maxlen = a + b - 1 > maxlen ? a + b - 1 : maxlen;
        hm[p].off1 = off1_host[p]; hm[p].off2 = off2_host[p];
        hm[p].n1 = (int32_t)a; hm[p].n2 = (int32_t)b;
        hm[p].dir_off = dwords;
        hm[p].nbands = (int32_t)((a + BAND - 1) / BAND);
        hm[p].nrounds = (int32_t)((b + 2 * BAND - 1) / BAND);
        if (a > 0 && b > 0) {
            dwords += (int64_t)hm[p].nbands * 2 * hm[p].nrounds * BAND;
            mmax = b > mmax ? b : mmax;
            ++count[bucket(p) + 1];
        }
    }
    ABN_REQUIRE(path_stride >= maxlen, "dtw: path_stride %lld < longest possible path %lld", (long long)path_stride,
                (long long)maxlen);
    ABN_REQUIRE(rows1 * D < (1LL << 62) && rows2 * D < (1LL << 62), "dtw: feature array too large");
    const WsPlan w = plan_ws_of(dwords, mmax, npairs);
    if (ws_bytes < w.total) { set_error("dtw: workspace too small (%lld < %lld bytes)", (long long)ws_bytes, (long long)w.total); return ABN_E_WORKSPACE; }
    int64_t nq = 0;
    {
        for (int b = 0; b < NB; ++b) count[b + 1] += count[b];
        nq = count[NB];
        for (int64_t p = 0; p < npairs; ++p)
            if (n1_host[p] > 0 && n2_host[p] > 0) hord[count[bucket(p)]++] = (int32_t)p;
    }
    // behind the queue: the pairs with an empty token (the traceback launches give them path_len = 0)
    if (nq < npairs) {
        int64_t ne = nq;
        for (int64_t p = 0; p < npairs && ne < npairs; ++p)
            if (!(n1_host[p] > 0 && n2_host[p] > 0)) hord[ne++] = (int32_t)p;
    }
    // metadata and queue in ONE copy (host_stage and the workspace lay them out alike), one memset for flags + counter
    if (w.order_off != meta_bytes ||
        hipMemcpyAsync(base + w.meta_off, hm, (size_t)(meta_bytes + npairs * 4), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemsetAsync(base + w.bad_off, 0, (size_t)(w.counter_off + 256 - w.bad_off), st) != hipSuccess) {
        set_error("dtw: metadata upload failed");
        return ABN_E_LAUNCH;
    }
    const PairMeta* dm = (const PairMeta*)(base + w.meta_off);
    {
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
        P.ynorm = (float*)(base + w.ynorm_off);
        P.done = (int32_t*)(base + w.done_off);
        int64_t nwg = (nq + 1) / 2;
        if (nwg > w.nwg) nwg = w.nwg;
        const bool vec = D % 4 == 0 && aligned16(feats1) && aligned16(feats2);
        const bool pipelined = switches().dtw_f40;     // A/B switch of the 40-d specialisation
        const bool gang = switches().dtw_pc;            // A/B switch: gang form (a producer per slot + one consumer)
        const int32_t* qorder = (const int32_t*)(base + w.order_off);
        const unsigned tb_grid = (unsigned)((nq + TB_SIDE_LANES - 1) / TB_SIDE_LANES), tb_grid_all = (unsigned)((npairs + 63) / 64);
        if (nq == 0) {}
        else if (vec && D == KCH && gang) {
            const int64_t ng = nwg < 256 * switches().dtw_wgs_per_cu ? nwg : 256 * switches().dtw_wgs_per_cu;
            if (side && side != st) {
                // The traceback beside the fill: the second stream's launch starts once the upload and the memset above
                // are done, polls the pairs' flags and walks each pair as it completes; this stream then waits for it
                // and sweeps up whatever it left.  Two events order the streams (kept per thread and device between calls:
                // creating and destroying a pair per call cost 8 us of the call's ~80 on the host).
                hipEvent_t ready = nullptr, traced = nullptr;
                if (!stream_order_events(&ready, &traced)) { set_error("dtw: hipEventCreate failed"); return ABN_E_LAUNCH; }
                bool ok = hipEventRecord(ready, st) == hipSuccess && hipStreamWaitEvent(side, ready, 0) == hipSuccess;
                if (ok) {
                    launch_gang(true, ng, st, P);
                    hipLaunchKernelGGL(dtw_traceback_kernel<1>, dim3(tb_grid), dim3(64), 0, side, dm, qorder, (int)nq, (int)npairs, P.dirs, P.bad,
                                       P.done, path1, path2, path_len, path_stride, total_cost);
                    ok = hipEventRecord(traced, side) == hipSuccess && hipStreamWaitEvent(st, traced, 0) == hipSuccess;
                    hipLaunchKernelGGL(dtw_traceback_kernel<2>, dim3(tb_grid_all), dim3(64), 0, st, dm, qorder, (int)nq, (int)npairs, P.dirs, P.bad,
                                       P.done, path1, path2, path_len, path_stride, total_cost);
                }
                if (!ok) { set_error("dtw: ordering the two streams failed"); return ABN_E_LAUNCH; }
                ABN_CHECK_LAUNCH("dtw");
                return ABN_OK;
            }
            launch_gang(false, ng, st, P);
        }
        else if (vec && D == KCH && pipelined) hipLaunchKernelGGL((dtw_fused_kernel<true, true>), dim3((unsigned)nwg), dim3(64), 0, st, P);
        else if (vec) hipLaunchKernelGGL((dtw_fused_kernel<true, false>), dim3((unsigned)nwg), dim3(64), 0, st, P);
        else hipLaunchKernelGGL((dtw_fused_kernel<false, false>), dim3((unsigned)nwg), dim3(64), 0, st, P);
        hipLaunchKernelGGL(dtw_traceback_kernel<0>, dim3(tb_grid_all), dim3(64), 0, st, dm, qorder, (int)nq, (int)npairs, P.dirs, P.bad,
                           P.done, path1, path2, path_len, path_stride, total_cost);
    }
    ABN_CHECK_LAUNCH("dtw");
    return ABN_OK;
}

extern "C" int abn_dtw_batched(const float* feats1, int64_t rows1, const float* feats2, int64_t rows2,
                               const int64_t* off1_host, const int32_t* n1_host, const int64_t* off2_host,
                               const int32_t* n2_host, int64_t npairs, int64_t D, int32_t* path1,
                               int32_t* path2, int32_t* path_len, int64_t path_stride, double* total_cost,
                               void* ws, int64_t ws_bytes, void* host_stage, int64_t host_stage_bytes,
                               void* stream)
{
    return dtw_batched_impl(feats1, rows1, feats2, rows2, off1_host, n1_host, off2_host, n2_host, npairs, D, path1, path2,
                            path_len, path_stride, total_cost, ws, ws_bytes, host_stage, host_stage_bytes, stream, nullptr);
}

extern "C" int abn_dtw_batched_overlap(const float* feats1, int64_t rows1, const float* feats2, int64_t rows2,
                                       const int64_t* off1_host, const int32_t* n1_host, const int64_t* off2_host,
                                       const int32_t* n2_host, int64_t npairs, int64_t D, int32_t* path1,
                                       int32_t* path2, int32_t* path_len, int64_t path_stride, double* total_cost,
                                       void* ws, int64_t ws_bytes, void* host_stage, int64_t host_stage_bytes,
                                       void* stream, void* side_stream)
{
    return dtw_batched_impl(feats1, rows1, feats2, rows2, off1_host, n1_host, off2_host, n2_host, npairs, D, path1, path2,
                            path_len, path_stride, total_cost, ws, ws_bytes, host_stage, host_stage_bytes, stream, side_stream);
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

#ifdef ABN_TB_STAMPS
extern "C" int abn_debug_tb_stamps(unsigned long long* out, int reset)
{
    if (reset) return hipMemcpyToSymbol(HIP_SYMBOL(abn::g_tb_stamps), out, sizeof(abn::g_tb_stamps)) == hipSuccess ? 0 : -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(abn::g_tb_stamps), sizeof(abn::g_tb_stamps)) == hipSuccess ? 0 : -1;
}
#endif
