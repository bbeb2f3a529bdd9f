// tower_bn_persist.h -- a BatchNorm tower in TRAINING as one launch per direction (north_star's "stacked
// Linear + BatchNorm + Sigmoid towers as a fused GEMM"; abnet3/model.py:139-140,149-150,158-159,194-195).
//
// The batch statistics of a layer span every workgroup's rows, so tower_planes.h cut the tower at every
// BatchNorm: per layer a product launch (bn_fwd_layer_kernel / bn_bwd_layer_kernel) and a finishing launch
// that adds the workgroups' column sums -- 4 + 4 + 8 launches per step at C2, every z_l and d loss / d a_l written
// and read back in between.  Here the grid is RESIDENT -- one workgroup of 32 rows per CU, at most as many
// workgroups as the device has CUs (the host checks; larger batches keep the layer launches) -- and the layers
// are separated by a grid-wide barrier instead of a kernel boundary:
//
//   k-loop of layer l (the chains' register ring, ring_kloop)       z_l stays in the accumulators
//   column sums of the workgroup's 32 rows  -> part[workgroup]      write-through (sc1) stores
//   grid barrier
//   every workgroup finishes ITS columns (c = workgroup, workgroup + grid, ...): the workgroups' sums in the
//   finishing kernels' fixed order, float64 -> mean / invstd / running statistics (forward), s1 / s2 / d gamma /
//   d beta (backward)                                               write-through stores
//   grid barrier
//   every workgroup reads the finished vectors of its call (sc1 loads), normalises its accumulators, applies the
//   activation, and goes on exactly as the chains do: row scales, operand image in LDS, transposed image for the
//   weight-gradient launch.
//
// The hand-overs carry their own flags (MI355X_MICROARCH.md, Valid forms, R2: "the data IS the flag"): a workgroup's column
// sums and a column's finished statistics travel as 16-byte granules [value, value, value, TAG], each written by ONE
// write-through (sc1) 16-byte store and polled with sc1 16-byte loads until the tag is the exchange's -- no counter, no
// fence, no grid barrier: a consumer waits for exactly the granules it reads, one round trip behind their producers (a
// counter barrier in front of every exchange and another behind it took 4 us each: 30 us per layer with the finishing step,
// more than the kernel boundaries they replaced).  Tags are unique per launch and exchange: (launch counter + 1) x 64 +
// exchange index; the launch counter lives in the caller's persistent, once-zeroed sync buffer (abn_tower_desc.sync_ws: also
// the granules' home, so a stale granule always carries an OLDER tag) and is advanced by workgroup 0 when it leaves (every
// workgroup has read it by then: workgroup 0 cannot finish the last exchange before all of them have published).
// Every spin is BOUNDED: a wave that gives up raises a failure word every poller watches, the launch drains, and the kernel
// poisons its outputs with NaN (the step's loss reads NaN: loud, never a hang).  The failure word is sticky: the caller
// zeroes sync_ws again before the buffer can be trusted.
#pragma once
#include "tower_planes.h"

namespace abn {

constexpr float BN_EPS = 1e-5f;
constexpr float BN_MOMENTUM = 0.1f;
constexpr int BN_WG_GROUPS = 16;          // thread groups of the finishing kernels (tower.hip): the order their sums are added in

constexpr unsigned BNP_SPIN_LIMIT = 1u << 21;
#ifndef BNP_POLL_SLEEP
#define BNP_POLL_SLEEP 1                  // x 64 cycles between two polls of a granule that was not there yet
#endif
constexpr int BNP_HDR_BYTES = 256;        // sync_ws: word 0 the launch counter, word 16 the failure word
constexpr int BNP_MAX_CALLS = 8;
constexpr int BNP_MAX_WGS = 1024;         // workgroups a sync buffer serves (abn_tower_sync_ws_bytes)
// sync_ws = header | finished-statistics granules [call][PL_MAXW] | column-sum granules [workgroup][PL_MAXW], 16 bytes each
__host__ __device__ inline int64_t bnp_stat_off() { return BNP_HDR_BYTES; }
__host__ __device__ inline int64_t bnp_part_off() { return BNP_HDR_BYTES + (int64_t)BNP_MAX_CALLS * PL_MAXW * 16; }
__host__ __device__ inline int64_t bnp_sync_bytes(int64_t max_wgs) { return bnp_part_off() + max_wgs * PL_MAXW * 16; }

// 16 bytes to base + byte_off as one write-through store (a buffer store with the sc1 cache bit)
__device__ __forceinline__ void st_sc1_4(const __amdgpu_buffer_rsrc_t& rs, int byte_off, const f32x4& v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, v), rs, byte_off, 0, 16);
}
// 16 bytes at base + byte_off with one sc1 load (a buffer load with the sc1 cache bit: it bypasses this CU's L1)
__device__ __forceinline__ f32x4 ld_sc1_4(const __amdgpu_buffer_rsrc_t& rs, int byte_off)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16));
}

// A workgroup barrier that orders LDS only.  (__syncthreads() also waits for every outstanding global store of the wave --
// s_waitcnt vmcnt(0) -- : behind a burst of 16 MB of z or image stores that is microseconds.)
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

struct GranuleSync {
    __amdgpu_buffer_rsrc_t rs;    // the whole sync_ws
    unsigned* hdr;                // its header (global)
    unsigned base;                // (launch counter + 1) << 6
    int* fail_s;                  // one LDS word: a wave of this workgroup gave up
};
__device__ __forceinline__ GranuleSync granule_open(void* sync_ws, int64_t bytes, int* fail_s)
{
    GranuleSync g;
    g.rs = __builtin_amdgcn_make_buffer_rsrc(sync_ws, 0, (int)bytes, 0x00020000);
    g.hdr = reinterpret_cast<unsigned*>(sync_ws);
    g.base = (g.hdr[0] + 1u) << 6;
    g.fail_s = fail_s;
    if (threadIdx.x == 0) *fail_s = 0;
    return g;
}
// The granule at byte_off once it carries `tag` (active lanes; the others return zeros at once).  Bounded: a wave that gives
// up -- or sees that another has -- raises both failure words and returns garbage; the caller checks *fail_s behind its next
// workgroup barrier.
__device__ __forceinline__ f32x4 granule_wait(const GranuleSync& g, int byte_off, unsigned tag, bool active)
{
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    bool done = !active;
    for (unsigned spins = 0;; ++spins) {
        if (!done) {
            v = ld_sc1_4(g.rs, byte_off);
            done = __float_as_uint(v[3]) == tag;
        }
        if (__all((int)done)) break;
        bool give_up = spins > BNP_SPIN_LIMIT;
        if ((spins & 31u) == 31u) give_up = give_up || __hip_atomic_load(g.hdr + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        if (give_up) {
            if ((threadIdx.x & 63) == 0) {
                __hip_atomic_store(g.hdr + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *g.fail_s = 1;
            }
            break;
        }
        __builtin_amdgcn_s_sleep(BNP_POLL_SLEEP);
    }
    return v;
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
struct BnPersistP {
    void* sync_ws;                       // abn_tower_desc.sync_ws (persistent, zeroed once by the caller)
    int64_t sync_bytes;
    const int* n_valid;                  // a padded batch (abn_tower_desc.n_valid), or null
    int n_calls;
    float* z[ABN_MAX_LAYERS];            // [rows][dims[l + 1]] row-major: what the backward normalises again
    float* mean[ABN_MAX_LAYERS];         // [n_calls][dims[l + 1]]
    float* invstd[ABN_MAX_LAYERS];
    float* var[ABN_MAX_LAYERS];
    float* rm[ABN_MAX_LAYERS];           // running statistics: one momentum update per call, in call order
    float* rv[ABN_MAX_LAYERS];
    long long* nbt[ABN_MAX_LAYERS];      // num_batches_tracked (int64, += n_calls), or null
    float* a_top;                        // [rows][dims[n_layers]] the embeddings, row-major
};

// The columns c = cols_per blockIdx.x + j of layer l are this workgroup's to finish: the workgroups' shifted sums
// [sum (z - c0), sum (z - c0)^2, c0] -> sum z, sum z^2 in float64, added in a fixed order (per call 16 / n_calls groups of
// consecutive workgroups, each in order, then the groups in order: bn_stats_finish_wg_kernel's), then mean, biased variance,
// invstd -- published as granules tagged tag_s --, and the running statistics (unbiased variance; the calls in order).
// Waits for the column-sum granules tagged tag_p.  scratch: PL_PART_BYTES of LDS.  false: a wait gave up (workgroup-uniform).
__device__ __forceinline__ bool bnp_fwd_finish(const PlanesFwdP& p, const BnPersistP& q, const GranuleSync& gs, unsigned tag_p, unsigned tag_s,
                                               int l, int N, int wpc, double* __restrict__ scratch)
{
    const int n_wg = gridDim.x, n_calls = q.n_calls;
    const int cols_per = (N + n_wg - 1) / n_wg;
    const int gpc = n_calls < BN_WG_GROUPS ? BN_WG_GROUPS / n_calls : 1;
    const int per = (wpc + gpc - 1) / gpc;
    int64_t rows_per_call = p.rows_call;
    if (q.n_valid) { const int64_t nv = *q.n_valid; rows_per_call = nv < 1 ? 1 : (nv < rows_per_call ? nv : rows_per_call); }
    // every (column, call, workgroup) term by a thread of its own: all granules in flight together
    const int total = cols_per * n_calls * wpc;              // <= N + grid <= 768 terms (two doubles each: PL_PART_BYTES hold 1024)
    double* const ta_s = scratch, * const tb_s = scratch + total;
    for (int i0 = 0; i0 < total; i0 += PL_NT) {              // (whole waves enter the wait)
        const int i = i0 + threadIdx.x;
        const int k = i % wpc, g = (i / wpc) % n_calls, j = i / (wpc * n_calls);
        const int c = blockIdx.x * cols_per + j;
        const bool live = i < total && c < N;
        const f32x4 v = granule_wait(gs, (int)bnp_part_off() + ((g * wpc + k) * PL_MAXW + (live ? c : 0)) * 16, tag_p, live);
#ifdef ABN_STAMPS
        // (diagnostic: when term i of the first 16 finishing workgroups was SEEN, on the chip-wide clock: rows 256 .. of the stamp buffer)
        if (p.stamps && l == 1 && blockIdx.x < 16 && i < 512) p.stamps[(size_t)(256 + blockIdx.x * 4 + (i >> 7)) * 128 + (i & 127)] = __builtin_amdgcn_s_memrealtime();
#endif
        if (i < total) {
            double a = 0.0, b = 0.0;
            if (live) {
                const double sd = v[0], sq = v[1], cc = v[2];
                const int64_t left = rows_per_call - (int64_t)k * PL_ROWS;
                const double nk = left < PL_ROWS ? (left > 0 ? (double)left : 0.0) : (double)PL_ROWS;      // rows of workgroup k
                a = sd + nk * cc;
                b = sq + 2.0 * cc * sd + nk * cc * cc;
            }
            ta_s[i] = a;
            tb_s[i] = b;
        }
    }
    lds_barrier();
    if (*gs.fail_s) return false;
    // the groups of consecutive workgroups, each in order (its sum replaces its first term) ...
    const int n_items = cols_per * n_calls * gpc;
    for (int item = threadIdx.x; item < n_items; item += PL_NT) {
        const int sub = item % gpc, base = (item / gpc) * wpc;
        const int k0 = sub * per, k1 = min(k0 + per, wpc);
        if (k0 < wpc) {
            double a = 0.0, b = 0.0;
#pragma unroll 8
            for (int k = k0; k < k1; ++k) { a += ta_s[base + k]; b += tb_s[base + k]; }      // (the reads of eight steps in flight; the adds in order)
            ta_s[base + k0] = a;
            tb_s[base + k0] = b;
        }
    }
    lds_barrier();
    // ... then the groups in order, a thread per column: mean, biased variance, invstd, the running statistics (the calls in order)
    const float unb = rows_per_call > 1 ? (float)((double)rows_per_call / (double)(rows_per_call - 1)) : 1.0f;
    for (int j = threadIdx.x; j < cols_per; j += PL_NT) {
        const int c = blockIdx.x * cols_per + j;
        if (c >= N) continue;
        float m_run = q.rm[l][c], v_run = q.rv[l][c];
        for (int g = 0; g < n_calls; ++g) {
            double ta = 0.0, tb = 0.0;
            for (int k0 = 0; k0 < wpc; k0 += per) { ta += ta_s[(j * n_calls + g) * wpc + k0]; tb += tb_s[(j * n_calls + g) * wpc + k0]; }
            const double n = (double)rows_per_call;
            const double m = ta / n;
            double var = tb / n - m * m;
            if (var < 0.0) var = 0.0;
            const int64_t idx = (int64_t)g * N + c;
            const float is = 1.0f / sqrtf((float)var + BN_EPS);
            st_sc1_4(gs.rs, (int)bnp_stat_off() + (g * PL_MAXW + c) * 16, f32x4{(float)m, is, 0.0f, __uint_as_float(tag_s)});      // what the other workgroups wait for
            q.mean[l][idx] = (float)m;                                              // what the backward (a later launch) reads
            q.var[l][idx] = (float)var;
            q.invstd[l][idx] = is;
            m_run = (1.0f - BN_MOMENTUM) * m_run + BN_MOMENTUM * (float)m;
            v_run = (1.0f - BN_MOMENTUM) * v_run + BN_MOMENTUM * ((float)var * unb);
        }
        q.rm[l][c] = m_run;
        q.rv[l][c] = v_run;
    }
    return true;
}

// (experiment, round 6, -DBNP_PREFETCH_NEXT: the NEXT layer's first PL_DEPTH ring steps requested before this layer's exchange --
// the weights do not depend on the statistics --, so that its k-loop starts on operands that are already there)
#ifndef BNP_PREFETCH_NEXT
#define BNP_PREFETCH_NEXT 0
#endif
template <int NP, int BPW, int KS>
__device__ __forceinline__ void bnp_prefetch_ring(WeightRing<NP>& ring, const char* wp, const char* wbase, int nblk, int nsteps, int wave, int lane)
{
    const WaveShare<BPW, KS> ws(wave, nblk, nsteps);
    if (!ws.active) return;
    const int ioff = (int)(wp - wbase);
#pragma unroll
    for (int i = 0; i < PL_DEPTH; ++i)
#pragma unroll
        for (int j = 0; j < BPW; ++j) {
            const int blk = ws.blk0 + j < nblk ? ws.blk0 + j : nblk - 1;
            const int wv = ioff + (blk * nsteps + ws.s_first) * (NP * 1024) + lane * 16;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                ring.wq[i][j][pl] = __builtin_amdgcn_raw_buffer_load_b128(ring.rs, wv, (i * NP + pl) * 1024, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    ring.filled = true;
}

// One layer of the resident forward.  img: the layer's input fragments (all pl_steps(K) steps); on return the output's
// (unless it is the last layer).  Returns false when a grid barrier gave up.
template <int NP, int BPW, int KS>
__device__ __forceinline__ bool bnp_fwd_layer(const PlanesFwdP& p, const BnPersistP& q, int l, char* __restrict__ img, float* __restrict__ part,
                                              const bf16x8* idf, int wave, int lane, int row0, int row_end, int call, int wpc, float& ainv,
                                              WeightRing<NP>& ring, const GranuleSync& gs)
{
    const int K = p.dims[l], N = p.dims[l + 1];
    const unsigned tag_p = gs.base + 2u * (unsigned)l + 1u, tag_s = tag_p + 1u;
    const int nsteps = pl_steps(K), nblk = (N + 31) / 32;
    const int r = lane & 31, h = lane >> 5;
    const WaveShare<BPW, KS> ws(wave, nblk, nsteps);
    const int blk0 = ws.blk0;
    const bool last = l + 1 == p.n_layers;
    const bool own = ws.active && ws.khalf == 0;          // this wave finishes blocks blk0 .. blk0 + BPW - 1

    f32x16 acc[BPW];
#pragma unroll
    for (int j = 0; j < BPW; ++j)
#pragma unroll
        for (int x = 0; x < 16; ++x) acc[j][x] = 0.0f;
    float cinv[BPW];
#pragma unroll
    for (int j = 0; j < BPW; ++j) cinv[j] = NP == 2 && ws.active ? packed_inv(p.wp[l], nblk, nsteps, blk0 + j) * ainv : 1.0f;
    // one value per lane of each per-feature vector (feature 32 blk0 + lane of the wave's up to 64): requested here,
    // parked in the wave's LDS slots, read back as the 16-byte pieces the accumulator layout wants
    const int nfeat = 32 * blk0 + lane;
    const int nfc = nfeat < N ? nfeat : N - 1;
    const bool flane = own && lane < 32 * BPW;
    float bias_lane = 0.0f, ga_lane = 1.0f, be_lane = 0.0f;
    if (flane) {
        bias_lane = p.b[l][nfc];
        ga_lane = p.bn_w[l][nfc];
        be_lane = p.bn_b[l][nfc];
    }
    if (ws.active) {
        if (!BNP_PREFETCH_NEXT) ring.filled = false;
        const int ioff = (int)(p.wp[l] - p.wbase);
        int wv[BPW], dnext[BPW];
#pragma unroll
        for (int j = 0; j < BPW; ++j) {
            const int blk = blk0 + j < nblk ? blk0 + j : nblk - 1;
            wv[j] = ioff + (blk * nsteps + ws.s_first) * (NP * 1024) + lane * 16;
            dnext[j] = 0;
        }
        ring_kloop<NP, BPW>(acc, ring, wv, dnext, false, img, ws.s_first, ws.my_steps, lane);
    }
    ring.filled = false;
    PSTAMPF(2 + 8 * l);
    float* const sc = part + PL_PART_BYTES / 4;
    float* const slot_s = part + (PL_PART_BYTES + PL_SC_BYTES) / 4 + wave * 64;      // + v * (PL_BIAS_BYTES / 4): bias, mean, invstd, gamma, beta
    constexpr int VS = PL_BIAS_BYTES / 4;
    slot_s[lane] = bias_lane;
    slot_s[3 * VS + lane] = ga_lane;
    slot_s[4 * VS + lane] = be_lane;
    if (KS == 2) {
        if (ws.active && ws.khalf == 1) {
#pragma unroll
            for (int x = 0; x < 16; ++x) part[((wave & 3) * 16 + x) * 64 + lane] = acc[0][x];
        }
        lds_barrier();
        if (own) {
#pragma unroll
            for (int x = 0; x < 16; ++x) acc[0][x] += part[(wave * 16 + x) * 64 + lane];
        }
        lds_barrier();                                              // (the buffer is written again below: the column sums)
    } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");        // (this wave's own LDS accesses complete in order)
        __builtin_amdgcn_wave_barrier();
    }
    const float* __restrict__ mask = p.mask[l];
    const DropGen drop = make_drop(mask ? nullptr : p.drop_seed, p.drop_p, l);
    const int gr = row0 + r;
    const bool row_ok = gr < row_end;
    const int call_end = (call + 1) * p.rows_call;         // rows of the call as allocated (a padded batch: row_end <= call_end)
    // z = (acc x scales + bias) x dropout: Linear -> Dropout -> BatchNorm (abnet3/model.py:136-140)
    if (own) {
        const float* mrow = mask ? mask + (int64_t)(row_ok ? gr : row0) * N : nullptr;
        auto z_loop = [&](auto mode) {                    // (a straight-line copy per dropout mode: bnp_bwd_layer)
            constexpr int MODE = decltype(mode)::value;   // 0 none, 1 mask tensor, 2 the per-forward seed hash
#pragma unroll
            for (int j = 0; j < BPW; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = 32 * (blk0 + j) + 4 * h + 8 * g;
                    const bool live = n < N;              // N % 4 == 0: four features in or out together
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(slot_s + 32 * j + 4 * h + 8 * g);
                    f32x4 m4 = {1.f, 1.f, 1.f, 1.f};
                    if constexpr (MODE == 1) m4 = *reinterpret_cast<const f32x4*>(mrow + (live ? n : N - 4));
                    if constexpr (MODE == 2) m4 = drop4(drop, gr, n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = (NP == 2 ? acc[j][4 * g + e] * cinv[j] : acc[j][4 * g + e]) + b4[e];
                        if constexpr (MODE != 0) v *= m4[e];
                        acc[j][4 * g + e] = live ? v : 0.0f;
                    }
                }
        };
        if (mrow) z_loop(std::integral_constant<int, 1>{});
        else if (drop.on) z_loop(std::integral_constant<int, 2>{});
        else z_loop(std::integral_constant<int, 0>{});
        // column statistics of this workgroup's rows, shifted by its first row's value (the float32 sums stay the size of
        // the variance), and z itself for the backward
        // (staged in the K-split buffer: [feature][sum (z - c) | sum (z - c)^2 | c | -])
#pragma unroll
        for (int j = 0; j < BPW; ++j) {
            if (blk0 + j >= nblk) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 sd, sq, cc;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float zv = acc[j][4 * g + e];
                    const float c0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(zv), 0));
                    const float c1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(zv), 32));
                    const float c = h ? c1 : c0;
                    const float d = row_ok ? zv - c : 0.0f;
                    sd[e] = d;
                    sq[e] = d * d;
                    cc[e] = c;
                }
                half_wave_sum4(sd);
                half_wave_sum4(sq);
                const int n = 32 * (blk0 + j) + 8 * g + 4 * h;
                if (r == 16 && n < PL_MAXW) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) *reinterpret_cast<f32x4*>(part + 4 * (n + e)) = f32x4{sd[e], sq[e], cc[e], __uint_as_float(tag_p)};
                }
            }
        }
    }
    // the workgroup's sums leave as ONE 16-byte write-through granule per thread, 8 KB contiguous (a store per lane and
    // value -- a thousand small write-through transactions per workgroup -- took 7 us to drain)
    lds_barrier();
    if ((int)threadIdx.x < N)
        st_sc1_4(gs.rs, (int)bnp_part_off() + ((int)blockIdx.x * PL_MAXW + (int)threadIdx.x) * 16, *reinterpret_cast<const f32x4*>(part + 4 * threadIdx.x));
    PSTAMPF(3 + 8 * l);
    PSTAMPR(9 + 8 * l);                                // (the chip-wide clock: when this workgroup's sums were out)
    if (BNP_PREFETCH_NEXT && !last) {
        const int nblk2 = (p.dims[l + 2] + 31) / 32, nsteps2 = pl_steps(p.dims[l + 1]);
        if (nblk2 > PL_WAVES) bnp_prefetch_ring<NP, 2, 1>(ring, p.wp[l + 1], p.wbase, nblk2, nsteps2, wave, lane);
        else if (nblk2 > PL_WAVES / 2 || nsteps2 % (2 * PL_DEPTH) != 0) bnp_prefetch_ring<NP, 1, 1>(ring, p.wp[l + 1], p.wbase, nblk2, nsteps2, wave, lane);
        else bnp_prefetch_ring<NP, 1, 2>(ring, p.wp[l + 1], p.wbase, nblk2, nsteps2, wave, lane);
    }
    lds_barrier();                                     // (the staging buffer is the finishing step's scratch)
    PSTAMPF(4 + 8 * l);
    if (!bnp_fwd_finish(p, q, gs, tag_p, tag_s, l, N, wpc, reinterpret_cast<double*>(part))) return false;
    PSTAMPF(5 + 8 * l);

    // the finished statistics of this workgroup's call -> the wave's LDS slots
    {
        const f32x4 st = granule_wait(gs, (int)bnp_stat_off() + (call * PL_MAXW + nfc) * 16, tag_s, flane);
        if (flane) {
            slot_s[VS + lane] = st[0];
            slot_s[2 * VS + lane] = st[1];
        }
    }
    // z itself, for the backward (a later launch: plain stores).  Here, behind the last load this layer waits for: whatever
    // is loaded next (vector memory operations retire in order) is a layer away.
    if (own) {
        float* const zrow = q.z[l] + (int64_t)gr * N;
#pragma unroll
        for (int j = 0; j < BPW; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = 32 * (blk0 + j) + 8 * g + 4 * h;
                if (row_ok && n < N) *reinterpret_cast<f32x4*>(zrow + n) = f32x4{acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
            }
    }
    lds_barrier();                                     // (also: this wave's own LDS slots are complete)
    if (*gs.fail_s) return false;
    PSTAMPF(6 + 8 * l);
    if (own) {
        with_act(p.act[l], [&](auto tag) {
            constexpr int ACT = decltype(tag)::value;
#pragma unroll
            for (int j = 0; j < BPW; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bool live = 32 * (blk0 + j) + 4 * h + 8 * g < N;
                    int so = 32 * j + 4 * h + 8 * g;
                    asm volatile("" : "+v"(so));         // (keeps each group's reads where they are used: tower_planes.h)
                    const float* const slot = slot_s + so;
                    const f32x4 mu4 = *reinterpret_cast<const f32x4*>(slot + VS), is4 = *reinterpret_cast<const f32x4*>(slot + 2 * VS);
                    const f32x4 ga4 = *reinterpret_cast<const f32x4*>(slot + 3 * VS), be4 = *reinterpret_cast<const f32x4*>(slot + 4 * VS);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = ((acc[j][4 * g + e] - mu4[e]) * is4[e]) * ga4[e] + be4[e];
                        acc[j][4 * g + e] = live && row_ok ? act_apply(v, ACT) : 0.0f;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        });
    }
    PSTAMPF(7 + 8 * l);
    if (last) {
        // the embeddings, row-major (zeros behind a padded call's real rows)
        if (own && gr < call_end) {
            float* const orow = q.a_top + (int64_t)gr * N;
#pragma unroll
            for (int j = 0; j < BPW; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = 32 * (blk0 + j) + 4 * h + 8 * g;
                    if (n < N) *reinterpret_cast<f32x4*>(orow + n) = f32x4{acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
                }
        }
        return true;
    }
    // as the chains' epilogue: the rows' scales (fp16 x 2), the operand image, the transposed image [a_l | 1]
    float osc = 1.0f;
    if constexpr (NP == 2) {
        float m = 0.0f;
        if (own) {
#pragma unroll
            for (int j = 0; j < BPW; ++j)
#pragma unroll
                for (int x = 0; x < 16; ++x) m = fmaxf(m, fabsf(acc[j][x]));
        }
        sc[wave * 64 + lane] = m;
        lds_barrier();
        float oinv;
        m = row_scales(sc, wave, lane, osc, oinv);
        ainv = oinv;
        if (p.tp[l + 1] && wave == 0) store_amax_rows(p.amax[l + 1] + (int64_t)blockIdx.x * PL_AMAX, m, 1.0f, lane);
    } else {
        lds_barrier();                               // every wave is done reading img
    }
    const float* const inv_tab = sc + PL_WAVES * 64 + wave * 32;
    char* const tp = p.tp[l + 1];
    if (own) {
#pragma unroll
        for (int j = 0; j < BPW; ++j) {
            const int blk = blk0 + j;
            if (blk < nblk) {
                Frag<NP> f[2];
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    const f32x4 v0 = {acc[j][8 * t2], acc[j][8 * t2 + 1], acc[j][8 * t2 + 2], acc[j][8 * t2 + 3]};
                    const f32x4 v1 = {acc[j][8 * t2 + 4], acc[j][8 * t2 + 5], acc[j][8 * t2 + 6], acc[j][8 * t2 + 7]};
                    f[t2] = make_frag<NP>(v0, v1, osc);
                    store_frag<NP>(img + (int64_t)(2 * blk + t2) * (NP * 1024) + lane * 16, f[t2]);
                }
                if (tp)
                    emit_planes<NP>(tp + ((int64_t)blk * p.tp_steps + 2 * blockIdx.x) * tile_bytes<NP>(), f, idf, lane,
                                    blk == N / 32 ? N % 32 : -1, row_end - row0, inv_tab);
            }
        }
    }
    if (tp && N % 32 == 0 && wave == PL_WAVES - 1) {       // the column of ones opens a block of its own
        Frag<NP> zf[2] = {};
        emit_planes<NP>(tp + ((int64_t)nblk * p.tp_steps + 2 * blockIdx.x) * tile_bytes<NP>(), zf, idf, lane, 0, row_end - row0, inv_tab);
    }
    {
        const int next_steps = pl_steps(N);
        const bf16x8 zz = {};
        for (int s = 2 * nblk + wave; s < next_steps; s += PL_WAVES)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<bf16x8*>(img + ((int64_t)s * NP + pl) * 1024 + lane * 16) = zz;
    }
    lds_barrier();
    PSTAMPF(8 + 8 * l);
    return true;
}

template <int NP>
__global__ __launch_bounds__(PL_NT) void bn_fwd_tower_kernel(PlanesFwdP p, BnPersistP q)
{
    extern __shared__ __attribute__((aligned(16))) char pl_smem[];
    char* const img = pl_smem;
    float* const part = reinterpret_cast<float*>(pl_smem + PL_MAXSTEPS * NP * 1024);
    int* const flag_s = reinterpret_cast<int*>(pl_smem + pl_lds_bytes(NP) + 4 * PL_BIAS_BYTES);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // workgroups never straddle two forward_once calls (bn_fwd_layer_kernel's map)
    const int wpc = (p.rows_call + PL_ROWS - 1) / PL_ROWS;
    const int call = blockIdx.x / wpc;
    const int row0 = call * p.rows_call + (blockIdx.x - call * wpc) * PL_ROWS;
    const int row_end = call * p.rows_call + (q.n_valid ? min(max(*q.n_valid, 0), p.rows_call) : p.rows_call);
    bf16x8 idf[2];
    make_identity<NP>(idf, lane);
    float ainv = 1.0f;
    WeightRing<NP> ring;
    ring_open(ring, p.wbase, p.wbytes);
    const GranuleSync gs = granule_open(q.sync_ws, q.sync_bytes, flag_s);

    PSTAMPF(0);
    planes_input_stage<NP, false>(p, img, part, idf, wave, lane, row0, ainv, row_end);
    lds_barrier();
    PSTAMPF(1);
    bool ok = true;
    for (int l = 0; l < p.n_layers && ok; ++l) {
        const int nblk = (p.dims[l + 1] + 31) / 32;
        if (nblk > PL_WAVES) ok = bnp_fwd_layer<NP, 2, 1>(p, q, l, img, part, idf, wave, lane, row0, row_end, call, wpc, ainv, ring, gs);
        else if (nblk > PL_WAVES / 2 || pl_steps(p.dims[l]) % (2 * PL_DEPTH) != 0) ok = bnp_fwd_layer<NP, 1, 1>(p, q, l, img, part, idf, wave, lane, row0, row_end, call, wpc, ainv, ring, gs);
        else ok = bnp_fwd_layer<NP, 1, 2>(p, q, l, img, part, idf, wave, lane, row0, row_end, call, wpc, ainv, ring, gs);
    }
    if (!ok) {
        // a grid barrier gave up (the grid was not resident): nothing of this launch may pass for a result
        const int NT = p.dims[p.n_layers];
        for (int i = threadIdx.x; i < PL_ROWS * NT; i += PL_NT) {
            const int gr = row0 + i / NT;
            if (gr < (call + 1) * p.rows_call) q.a_top[(int64_t)gr * NT + i % NT] = __builtin_nanf("");
        }
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x < p.n_layers && q.nbt[threadIdx.x]) *q.nbt[threadIdx.x] += q.n_calls;
    if (blockIdx.x == 0 && threadIdx.x == 0) gs.hdr[0] = (gs.base >> 6);      // the launch counter, advanced (every workgroup read it long ago)
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
struct BnBwdTowerP {
    int n_layers, rows, rows_call, n_calls;
    int dims[ABN_MAX_LAYERS + 1];
    int act[ABN_MAX_LAYERS];               // the activation behind BatchNorm l
    const float* z[ABN_MAX_LAYERS];        // [rows][dims[l + 1]] the forward's pre-normalisation values
    const float* mean[ABN_MAX_LAYERS];     // [n_calls][dims[l + 1]]
    const float* invstd[ABN_MAX_LAYERS];
    const float* gamma[ABN_MAX_LAYERS];
    const float* beta[ABN_MAX_LAYERS];
    float* dgamma[ABN_MAX_LAYERS];
    float* dbeta[ABN_MAX_LAYERS];
    const float* mask[ABN_MAX_LAYERS];
    const unsigned long long* drop_seed;
    float drop_p;
    const char* wpt[ABN_MAX_LAYERS];       // packed W_l^T
    const char* wbase;
    int64_t wbytes;
    char* dzp[ABN_MAX_LAYERS];             // out: transposed images of dz_l (the weight-gradient launch's operand)
    float* amax_dz[ABN_MAX_LAYERS];
    int64_t tp_steps;
    float* dx;                             // optional [rows][dims[0]]
    const float* d_out;                    // [rows][dims[n_layers]] d loss / d a of the output layer (null: the pair loss rides along)
    const float* a_top;                    // [rows][dims[n_layers]] the embeddings (pair loss)
    void* sync_ws;                         // abn_tower_desc.sync_ws: the column-sum and finished-sum granules
    int64_t sync_bytes;
    const int* n_valid;
    float n_stat;                          // rows the statistics of a call span
    // the pair loss riding along (loss_kind < 0: d_out is given): rows are [call 0 = tower 1 | call 1 = tower 2]
    int loss_kind, y_dtype;
    const void* y;
    double margin, scale;
    double* loss_partial;
    unsigned* loss_counter;
    float* loss_out;
    double* loss_accum;
};

// s1 = sum dy, s2 = sum dy xhat per (call, column) of this workgroup's columns from the workgroups' sums (granules tagged
// tag_p), in bn_bwd_finish_wg_kernel's order, published as granules tagged tag_s; d gamma = sum over the calls of s2, d beta
// of s1.  false: a wait gave up (workgroup-uniform).
__device__ __forceinline__ bool bnp_bwd_finish(const BnBwdTowerP& q, const GranuleSync& gs, unsigned tag_p, unsigned tag_s, int li, int C, int wpc,
                                               double* __restrict__ scratch)
{
    const int n_wg = gridDim.x, n_calls = q.n_calls;
    const int cols_per = (C + n_wg - 1) / n_wg;
    const int gpc = n_calls < BN_WG_GROUPS ? BN_WG_GROUPS / n_calls : 1;
    const int per = (wpc + gpc - 1) / gpc;
    const int total = cols_per * n_calls * wpc;              // a thread per term (bnp_fwd_finish)
    double* const ta_s = scratch, * const tb_s = scratch + total;
    for (int i0 = 0; i0 < total; i0 += PL_NT) {
        const int i = i0 + threadIdx.x;
        const int k = i % wpc, g = (i / wpc) % n_calls, j = i / (wpc * n_calls);
        const int c = blockIdx.x * cols_per + j;
        const bool live = i < total && c < C;
        const f32x4 v = granule_wait(gs, (int)bnp_part_off() + ((g * wpc + k) * PL_MAXW + (live ? c : 0)) * 16, tag_p, live);
        if (i < total) {
            ta_s[i] = live ? (double)v[0] : 0.0;
            tb_s[i] = live ? (double)v[1] : 0.0;
        }
    }
    lds_barrier();
    if (*gs.fail_s) return false;
    const int n_items = cols_per * n_calls * gpc;
    for (int item = threadIdx.x; item < n_items; item += PL_NT) {
        const int sub = item % gpc, base = (item / gpc) * wpc;
        const int k0 = sub * per, k1 = min(k0 + per, wpc);
        if (k0 < wpc) {
            double a = 0.0, b = 0.0;
#pragma unroll 8
            for (int k = k0; k < k1; ++k) { a += ta_s[base + k]; b += tb_s[base + k]; }      // (the reads of eight steps in flight; the adds in order)
            ta_s[base + k0] = a;
            tb_s[base + k0] = b;
        }
    }
    lds_barrier();
    for (int j = threadIdx.x; j < cols_per; j += PL_NT) {
        const int c = blockIdx.x * cols_per + j;
        if (c >= C) continue;
        float sg = 0.0f, sbeta = 0.0f;
        for (int g = 0; g < n_calls; ++g) {
            double ta = 0.0, tb = 0.0;
            for (int k0 = 0; k0 < wpc; k0 += per) { ta += ta_s[(j * n_calls + g) * wpc + k0]; tb += tb_s[(j * n_calls + g) * wpc + k0]; }
            st_sc1_4(gs.rs, (int)bnp_stat_off() + (g * PL_MAXW + c) * 16, f32x4{(float)ta, (float)tb, 0.0f, __uint_as_float(tag_s)});
            sg += (float)tb;
            sbeta += (float)ta;
        }
        q.dgamma[li][c] = sg;
        q.dbeta[li][c] = sbeta;
    }
    return true;
}

// d loss / d a_{l-1} = dz_l W_l (dz_l in img), then -- l >= 1 -- BatchNorm l - 1 backwards in the accumulators: dy = da act'(a),
// the workgroup's sums of dy and dy xhat, the two grid barriers around the finishing step, dz_{l-1} = gamma invstd / n
// (n dy - s1 - xhat s2) mask -> the operand image and the transposed image of the weight-gradient launch.
template <int NP, int BPW, int KS>
__device__ __forceinline__ bool bnp_bwd_layer(const BnBwdTowerP& q, int l, char* __restrict__ img, float* __restrict__ part, const bf16x8* idf,
                                              int wave, int lane, int row0, int row_end, int call, int wpc, float nf, float& ainv,
                                              WeightRing<NP>& ring, const GranuleSync& gs, unsigned tag_p)
{
    const unsigned tag_s = tag_p + 1u;
    const int N = q.dims[l + 1], K = q.dims[l];           // sum over N, K output features
    const int nsteps = pl_steps(N), nblk = (K + 31) / 32;
    const int r = lane & 31, h = lane >> 5;
    const WaveShare<BPW, KS> ws(wave, nblk, nsteps);
    const int blk0 = ws.blk0;
    const bool own = ws.active && ws.khalf == 0;
    f32x16 acc[BPW];
#pragma unroll
    for (int j = 0; j < BPW; ++j)
#pragma unroll
        for (int x = 0; x < 16; ++x) acc[j][x] = 0.0f;
    if (ws.active) {
        ring.filled = false;
        const int ioff = (int)(q.wpt[l] - q.wbase);
        int wv[BPW], dnext[BPW];
#pragma unroll
        for (int j = 0; j < BPW; ++j) {
            const int blk = blk0 + j < nblk ? blk0 + j : nblk - 1;
            wv[j] = ioff + (blk * nsteps + ws.s_first) * (NP * 1024) + lane * 16;
            dnext[j] = 0;
        }
        ring_kloop<NP, BPW>(acc, ring, wv, dnext, false, img, ws.s_first, ws.my_steps, lane);
    }
    if (KS == 2) {
        if (ws.active && ws.khalf == 1) {
#pragma unroll
            for (int x = 0; x < 16; ++x) part[((wave & 3) * 16 + x) * 64 + lane] = acc[0][x];
        }
        lds_barrier();
        if (own) {
#pragma unroll
            for (int x = 0; x < 16; ++x) acc[0][x] += part[(wave * 16 + x) * 64 + lane];
        }
        lds_barrier();                                              // (the buffer is written again below: the column sums)
    }
    if constexpr (NP == 2) {
        if (own) {
#pragma unroll
            for (int j = 0; j < BPW; ++j) {
                const float cinv = packed_inv(q.wpt[l], nblk, nsteps, blk0 + j) * ainv;
#pragma unroll
                for (int x = 0; x < 16; ++x) acc[j][x] *= cinv;
            }
        }
    }
    const int gr = row0 + r;
    const bool row_ok = gr < row_end;
    const int grc = row_ok ? gr : row0;
    if (l == 0) {                                         // d loss / d input: the plain product
        if (own && row_ok) {
            float* const orow = q.dx + (int64_t)gr * K;
#pragma unroll
            for (int j = 0; j < BPW; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = 32 * (blk0 + j) + 8 * g + 4 * h;
                    if (n < K) *reinterpret_cast<f32x4*>(orow + n) = f32x4{acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
                }
        }
        return true;
    }
    const int lp = l - 1;                                 // the BatchNorm layer in hand: K features
    f32x16 xh[BPW];
    if (own) {
        float* const pw = part;                           // staged in the K-split buffer: [feature][sum dy | sum dy xhat]
        with_act(q.act[lp], [&](auto tag) {
            constexpr int ACT = decltype(tag)::value;
#pragma unroll
            for (int j = 0; j < BPW; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = 32 * (blk0 + j) + 8 * g + 4 * h;
                    const bool live = n < K;              // K % 4 == 0: four features in or out together
                    const int nc = live ? n : K - 4;
                    const f32x4 z4 = *reinterpret_cast<const f32x4*>(q.z[lp] + (int64_t)grc * K + nc);
                    const f32x4 mu = *reinterpret_cast<const f32x4*>(q.mean[lp] + (int64_t)call * K + nc);
                    const f32x4 is = *reinterpret_cast<const f32x4*>(q.invstd[lp] + (int64_t)call * K + nc);
                    const f32x4 ga = *reinterpret_cast<const f32x4*>(q.gamma[lp] + nc), be = *reinterpret_cast<const f32x4*>(q.beta[lp] + nc);
                    f32x4 sd, sq;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = (z4[e] - mu[e]) * is[e];
                        const float a = act_apply(x * ga[e] + be[e], ACT);
                        const float dy = live && row_ok ? acc[j][4 * g + e] * act_grad(a, ACT) : 0.0f;
                        acc[j][4 * g + e] = dy;
                        xh[j][4 * g + e] = x;
                        sd[e] = dy;
                        sq[e] = dy * x;
                    }
                    half_wave_sum4(sd);
                    half_wave_sum4(sq);
                    if (r == 16 && live) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) *reinterpret_cast<f32x4*>(pw + 4 * (n + e)) = f32x4{sd[e], sq[e], 0.0f, __uint_as_float(tag_p)};
                    }
                }
        });
    }
    // one 16-byte write-through granule per thread, 8 KB contiguous (bnp_fwd_layer)
    lds_barrier();
    if ((int)threadIdx.x < K)
        st_sc1_4(gs.rs, (int)bnp_part_off() + ((int)blockIdx.x * PL_MAXW + (int)threadIdx.x) * 16, *reinterpret_cast<const f32x4*>(part + 4 * threadIdx.x));
    lds_barrier();                                     // (the staging buffer is the finishing step's scratch)
    if (!bnp_bwd_finish(q, gs, tag_p, tag_s, lp, K, wpc, reinterpret_cast<double*>(part))) return false;

    // the finished sums of this workgroup's call and k = gamma invstd / n, one value per lane, through the wave's LDS slots
    float* const sc = part + PL_PART_BYTES / 4;
    float* const slot_s = part + (PL_PART_BYTES + PL_SC_BYTES) / 4 + wave * 64;
    constexpr int VS = PL_BIAS_BYTES / 4;
    {
        const bool flane = own && lane < 32 * BPW;
        const int nfeat = 32 * blk0 + lane;
        const int nfc = nfeat < K ? nfeat : K - 1;
        const f32x4 st = granule_wait(gs, (int)bnp_stat_off() + (call * PL_MAXW + nfc) * 16, tag_s, flane);
        if (flane) {
            slot_s[lane] = st[0];
            slot_s[VS + lane] = st[1];
            slot_s[2 * VS + lane] = q.gamma[lp][nfc] * q.invstd[lp][(int64_t)call * K + nfc] / nf;
        }
    }
    lds_barrier();                                     // (also: this wave's own LDS slots are complete)
    if (*gs.fail_s) return false;
    const float* __restrict__ mask = q.mask[lp];
    const DropGen drop = make_drop(mask ? nullptr : q.drop_seed, q.drop_p, lp);
    if (own) {
        // (one straight-line copy per dropout mode: with the mode's branches INSIDE the unrolled loop the register allocator
        // spilled around every one of them, and a spill's reload here is a round trip to HBM)
        auto dz_loop = [&](auto mode) {
            constexpr int MODE = decltype(mode)::value;   // 0 none, 1 mask tensor, 2 the forward's seed hash
#pragma unroll
            for (int j = 0; j < BPW; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = 32 * (blk0 + j) + 8 * g + 4 * h;
                    const bool live = n < K;
                    const int so = 32 * j + 4 * h + 8 * g;
                    const f32x4 a1 = *reinterpret_cast<const f32x4*>(slot_s + so), a2 = *reinterpret_cast<const f32x4*>(slot_s + VS + so);
                    const f32x4 k4 = *reinterpret_cast<const f32x4*>(slot_s + 2 * VS + so);
                    f32x4 m4 = {1.f, 1.f, 1.f, 1.f};
                    if constexpr (MODE == 1) m4 = *reinterpret_cast<const f32x4*>(mask + (int64_t)grc * K + (live ? n : K - 4));
                    if constexpr (MODE == 2) m4 = drop4(drop, gr, n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = k4[e] * (nf * acc[j][4 * g + e] - a1[e] - xh[j][4 * g + e] * a2[e]) * m4[e];
                        acc[j][4 * g + e] = live && row_ok ? v : 0.0f;
                    }
                }
        };
        if (mask) dz_loop(std::integral_constant<int, 1>{});
        else if (drop.on) dz_loop(std::integral_constant<int, 2>{});
        else dz_loop(std::integral_constant<int, 0>{});
    }
    float osc = 1.0f;
    if constexpr (NP == 2) {
        float m = 0.0f;
        if (own) {
#pragma unroll
            for (int j = 0; j < BPW; ++j)
#pragma unroll
                for (int x = 0; x < 16; ++x) m = fmaxf(m, fabsf(acc[j][x]));
        }
        sc[wave * 64 + lane] = m;
        lds_barrier();
        float oinv;
        m = row_scales(sc, wave, lane, osc, oinv);
        ainv = oinv;
        if (wave == 0) store_amax_rows(q.amax_dz[lp] + (int64_t)blockIdx.x * PL_AMAX, m, 0.0f, lane);
    }
    const float* const inv_tab = sc + PL_WAVES * 64 + wave * 32;
    if (own) {
#pragma unroll
        for (int j = 0; j < BPW; ++j) {
            const int blk = blk0 + j;
            if (blk < nblk) {
                Frag<NP> f[2];
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    const f32x4 v0 = {acc[j][8 * t2], acc[j][8 * t2 + 1], acc[j][8 * t2 + 2], acc[j][8 * t2 + 3]};
                    const f32x4 v1 = {acc[j][8 * t2 + 4], acc[j][8 * t2 + 5], acc[j][8 * t2 + 6], acc[j][8 * t2 + 7]};
                    f[t2] = make_frag<NP>(v0, v1, osc);
                    store_frag<NP>(img + (int64_t)(2 * blk + t2) * (NP * 1024) + lane * 16, f[t2]);
                }
                emit_planes<NP>(q.dzp[lp] + ((int64_t)blk * q.tp_steps + 2 * blockIdx.x) * tile_bytes<NP>(), f, idf, lane, -1, 0, inv_tab);
            }
        }
    }
    {
        const int next_steps = pl_steps(K);
        const bf16x8 zz = {};
        for (int s = 2 * nblk + wave; s < next_steps; s += PL_WAVES)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<bf16x8*>(img + ((int64_t)s * NP + pl) * 1024 + lane * 16) = zz;
    }
    lds_barrier();
    return true;
}

template <int NP>
__global__ __launch_bounds__(PL_NT) void bn_bwd_tower_kernel(BnBwdTowerP q)
{
    extern __shared__ __attribute__((aligned(16))) char pl_smem[];
    char* const img = pl_smem;
    float* const part = reinterpret_cast<float*>(pl_smem + PL_MAXSTEPS * NP * 1024);
    int* const flag_s = reinterpret_cast<int*>(pl_smem + pl_lds_bytes(NP) + 4 * PL_BIAS_BYTES);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wpc = (q.rows_call + PL_ROWS - 1) / PL_ROWS;        // (the forward's workgroup -> rows map)
    const int call = blockIdx.x / wpc;
    const int row0 = call * q.rows_call + (blockIdx.x - call * wpc) * PL_ROWS;
    const int nv = q.n_valid ? min(max(*q.n_valid, 0), q.rows_call) : q.rows_call;
    const int row_end = call * q.rows_call + nv;
    const float nf = q.n_valid ? (float)(nv > 0 ? nv : 1) : q.n_stat;
    const int top = q.n_layers - 1;
    const int NT = q.dims[top + 1];
    bf16x8 idf[2];
    make_identity<NP>(idf, lane);
    const GranuleSync gs = granule_open(q.sync_ws, q.sync_bytes, flag_s);

    // the output layer's per-feature vectors of this workgroup's call, parked in the (idle) K-split buffer; behind them
    // the pair loss's per-row coefficients
    float* const mu_s = part, * const is_s = part + PL_MAXW, * const ga_s = part + 2 * PL_MAXW, * const be_s = part + 3 * PL_MAXW,
               * const s1_s = part + 4 * PL_MAXW, * const s2_s = part + 5 * PL_MAXW, * const k_s = part + 6 * PL_MAXW;
    double* const coef = reinterpret_cast<double*>(part + 7 * PL_MAXW);      // [32][2]
    double* const term_s = coef + 64;                                        // [32]
    static_assert(7 * PL_MAXW * 4 + 96 * 8 <= PL_PART_BYTES, "the K-split buffer holds the vectors and the coefficients");
    for (int c = threadIdx.x; c < NT; c += PL_NT) {
        mu_s[c] = q.mean[top][(int64_t)call * NT + c];
        is_s[c] = q.invstd[top][(int64_t)call * NT + c];
        ga_s[c] = q.gamma[top][c];
        be_s[c] = q.beta[top][c];
    }
    const bool with_loss = q.loss_kind >= 0;
    const int B = q.rows_call;
    double lscale = 1.0;
    if (with_loss) {
        const int Bv = q.n_valid ? *q.n_valid : B;
        lscale = q.n_valid && q.scale != 1.0 ? 1.0 / (double)(Bv > 0 ? Bv : 1) : q.scale;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int lr = 4 * wave + 2 * it + (lane >> 5), ll = lane & 31;
            const int g = row0 + lr;
            double inv = 0.0, kself = 0.0, term = 0.0;
            const int pi_ = g < row_end ? g - call * B : 0;
            const bool ok = g < row_end && pi_ < Bv;
            const float* a = q.a_top + (int64_t)pi_ * NT;           // e1[pair], e2[pair]: the order loss.hip sums in
            const float* b = q.a_top + (int64_t)(B + pi_) * NT;
            double dot = 0.0, s11 = 0.0, s22 = 0.0;
            for (int c = ll; c < NT / 4; c += 32) {
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
                switch (q.y_dtype) {
                    case ABN_Y_I8: v = ((const int8_t*)q.y)[pi_]; break;
                    case ABN_Y_I32: v = ((const int32_t*)q.y)[pi_]; break;
                    case ABN_Y_I64: v = (double)((const int64_t*)q.y)[pi_]; break;
                    case ABN_Y_F32: v = ((const float*)q.y)[pi_]; break;
                    default: v = ((const double*)q.y)[pi_]; break;
                }
                const int code = v == 1.0 ? 1 : (v == -1.0 ? -1 : 0);
                double dcos;
                if (q.loss_kind == ABN_LOSS_COSCOS2) {
                    if (code == 1) { term = (1.0 - cs) * 0.5; dcos = -0.5; }
                    else if (code == -1) { term = cs * cs; dcos = 2.0 * cs; }
                    else { term = cs; dcos = 1.0; }
                } else {
                    if (code == 1) { term = 1.0 - cs; dcos = -1.0; }
                    else if (code == -1) { const double hh = cs - q.margin; term = hh > 0.0 ? hh : 0.0; dcos = hh >= 0.0 ? 1.0 : 0.0; }
                    else { term = cs; dcos = 1.0; }
                }
                dcos *= lscale;
                inv = dcos / (c1 * c2);
                const double k1 = n1 > 0.0 ? dcos * cs / (c1 * n1) : 0.0;
                const double k2 = n2 > 0.0 ? dcos * cs / (c2 * n2) : 0.0;
                kself = call ? k2 : k1;
                if (call) term = 0.0;                                // a pair's term counts once
            }
            if (ll == 0) { coef[2 * lr] = inv; coef[2 * lr + 1] = kself; term_s[lr] = term; }
        }
    }
    lds_barrier();

    // dy = da act'(a) of the output layer for this lane's row, in the staging layout (a wave's blocks wave, wave + 8; two
    // 16-feature steps each), and the workgroup's sums of dy and dy xhat
    const int steps_t = pl_steps(NT), blocks_t = steps_t / 2;
    const int gr = row0 + r;
    const bool row_ok = gr < row_end;
    const int grc = row_ok ? gr : row0;
    const float* const zrow = q.z[top] + (int64_t)grc * NT;
    const float* const self_row = q.a_top + (int64_t)grc * NT;
    const float* const partner_row = q.a_top + (int64_t)(with_loss ? (call ? grc - B : grc + B) : grc) * NT;
    const float* const drow = q.d_out ? q.d_out + (int64_t)grc * NT : nullptr;
    const double my_inv = with_loss ? coef[2 * r] : 0.0, my_k = with_loss ? coef[2 * r + 1] : 0.0;
    f32x4 dyv[2][2][2], xhv[2][2][2];
    {
        float* const pw = reinterpret_cast<float*>(img);      // staged in the (still idle) operand image: [feature][sum dy | sum dy xhat]
        with_act(q.act[top], [&](auto tag) {
            constexpr int ACT = decltype(tag)::value;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const int kb = wave + PL_WAVES * u;
                        const int c = 16 * (2 * kb + t2) + 4 * h + 8 * hf;
                        f32x4 dy4 = {0.f, 0.f, 0.f, 0.f}, x4 = dy4;
                        const bool live = kb < blocks_t && c < NT;
                        if (live) {
                            f32x4 da4;
                            if (with_loss) {
                                const f32x4 es = *reinterpret_cast<const f32x4*>(self_row + c);
                                const f32x4 ep = *reinterpret_cast<const f32x4*>(partner_row + c);
#pragma unroll
                                for (int e = 0; e < 4; ++e) da4[e] = (float)(ep[e] * my_inv - es[e] * my_k);
                            } else {
                                da4 = *reinterpret_cast<const f32x4*>(drow + c);
                            }
                            const f32x4 z4 = *reinterpret_cast<const f32x4*>(zrow + c);
                            const f32x4 mu = *reinterpret_cast<const f32x4*>(mu_s + c), is = *reinterpret_cast<const f32x4*>(is_s + c);
                            const f32x4 ga = *reinterpret_cast<const f32x4*>(ga_s + c), be = *reinterpret_cast<const f32x4*>(be_s + c);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                x4[e] = (z4[e] - mu[e]) * is[e];
                                const float a = act_apply(x4[e] * ga[e] + be[e], ACT);
                                dy4[e] = row_ok ? da4[e] * act_grad(a, ACT) : 0.0f;
                            }
                        }
                        dyv[u][t2][hf] = dy4;
                        xhv[u][t2][hf] = x4;
                        if (kb < blocks_t) {             // (wave-uniform: the sums run over whole half waves)
                            f32x4 sd, sq;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                sd[e] = dy4[e];
                                sq[e] = dy4[e] * x4[e];
                            }
                            half_wave_sum4(sd);
                            half_wave_sum4(sq);
                            if (r == 16 && c < NT) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) *reinterpret_cast<f32x4*>(pw + 4 * (c + e)) = f32x4{sd[e], sq[e], 0.0f, __uint_as_float(gs.base + 1u)};
                            }
                        }
                    }
        });
    }
    if (with_loss) {
        // the loss terms: a ticket per workgroup, the last workgroup to arrive adds all partials in a fixed order
        if (wave == 0) {
            int lastw = 0;
            if (lane == 0) {
                double sum = 0.0;
                for (int i = 0; i < 32; ++i) sum += term_s[i];
                lastw = abn_ticket_publish(&q.loss_partial[blockIdx.x], sum, q.loss_counter, gridDim.x);      // (common.h)
            }
            lastw = __shfl(lastw, 0, 64);
            if (lastw) {
                double sum = 0.0;
                for (int i = lane; i < (int)gridDim.x; i += 64) sum += abn_ticket_partial(&q.loss_partial[i]);
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 64);
                if (lane == 0) {
                    const float lv = (float)(sum * lscale);
                    *q.loss_out = lv;
                    if (q.loss_accum) *q.loss_accum += (double)lv;        // (one thread of one workgroup per call, calls in stream order)
                    __hip_atomic_store(q.loss_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
    lds_barrier();
    if ((int)threadIdx.x < NT)
        st_sc1_4(gs.rs, (int)bnp_part_off() + ((int)blockIdx.x * PL_MAXW + (int)threadIdx.x) * 16, *reinterpret_cast<const f32x4*>(img + 16 * threadIdx.x));
    lds_barrier();
    // (the finishing step takes the K-split buffer: the vectors parked there are read again from memory below)
    bool ok = bnp_bwd_finish(q, gs, gs.base + 1u, gs.base + 2u, top, NT, wpc, reinterpret_cast<double*>(part));
    float ainv = 1.0f;
    if (ok) {
        {
            const int c = threadIdx.x < (unsigned)NT ? (int)threadIdx.x : 0;         // (NT <= 512: a column per thread)
            const f32x4 st = granule_wait(gs, (int)bnp_stat_off() + (call * PL_MAXW + c) * 16, gs.base + 2u, (int)threadIdx.x < NT);
            if ((int)threadIdx.x < NT) {
                s1_s[c] = st[0];
                s2_s[c] = st[1];
                k_s[c] = q.gamma[top][c] * q.invstd[top][(int64_t)call * NT + c] / nf;
            }
        }
        lds_barrier();
        ok = *gs.fail_s == 0;
    }
    if (ok) {
        const float* const mrow = q.mask[top] ? q.mask[top] + (int64_t)grc * NT : nullptr;
        const DropGen drop = make_drop(q.mask[top] ? nullptr : q.drop_seed, q.drop_p, top);
        float* const sc = part + PL_PART_BYTES / 4;
        float m = 0.0f;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int kb = wave + PL_WAVES * u;
                    const int c = 16 * (2 * kb + t2) + 4 * h + 8 * hf;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (kb < blocks_t && c < NT && row_ok) {
                        const f32x4 k4 = *reinterpret_cast<const f32x4*>(k_s + c);
                        const f32x4 a1 = *reinterpret_cast<const f32x4*>(s1_s + c), a2 = *reinterpret_cast<const f32x4*>(s2_s + c);
                        f32x4 m4 = {1.f, 1.f, 1.f, 1.f};
                        if (mrow) m4 = *reinterpret_cast<const f32x4*>(mrow + c);
                        else if (drop.on) m4 = drop4(drop, gr, c);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = k4[e] * (nf * dyv[u][t2][hf][e] - a1[e] - xhv[u][t2][hf][e] * a2[e]) * m4[e];
                    }
                    dyv[u][t2][hf] = v;
                }
        float osc = 1.0f;
        if constexpr (NP == 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) m = fmaxf(m, absmax8(dyv[u][t2][0], dyv[u][t2][1]));
            sc[wave * 64 + lane] = m;
            lds_barrier();
            m = row_scales(sc, wave, lane, osc, ainv);
            if (wave == 0) store_amax_rows(q.amax_dz[top] + (int64_t)blockIdx.x * PL_AMAX, m, 0.0f, lane);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int kb = wave + PL_WAVES * u;
            if (kb < blocks_t) {
                Frag<NP> f[2];
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    f[t2] = make_frag<NP>(dyv[u][t2][0], dyv[u][t2][1], osc);
                    store_frag<NP>(img + (int64_t)(2 * kb + t2) * (NP * 1024) + lane * 16, f[t2]);
                }
                if (kb < pl_blocks(NT))
                    emit_planes<NP>(q.dzp[top] + ((int64_t)kb * q.tp_steps + 2 * blockIdx.x) * tile_bytes<NP>(), f, idf, lane, -1, 0,
                                    sc + PL_WAVES * 64 + wave * 32);
            }
        }
        lds_barrier();
        WeightRing<NP> ring;
        ring_open(ring, q.wbase, q.wbytes);
        for (int l = top; l >= (q.dx ? 0 : 1) && ok; --l) {
            const int nblk = (q.dims[l] + 31) / 32;
            const unsigned tag_p = gs.base + 3u + 2u * (unsigned)(top - l);      // (exchanges 1, 2: the output layer's above)
            if (nblk > PL_WAVES) ok = bnp_bwd_layer<NP, 2, 1>(q, l, img, part, idf, wave, lane, row0, row_end, call, wpc, nf, ainv, ring, gs, tag_p);
            else if (nblk > PL_WAVES / 2 || pl_steps(q.dims[l + 1]) % (2 * PL_DEPTH) != 0) ok = bnp_bwd_layer<NP, 1, 1>(q, l, img, part, idf, wave, lane, row0, row_end, call, wpc, nf, ainv, ring, gs, tag_p);
            else ok = bnp_bwd_layer<NP, 1, 2>(q, l, img, part, idf, wave, lane, row0, row_end, call, wpc, nf, ainv, ring, gs, tag_p);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) gs.hdr[0] = (gs.base >> 6);      // the launch counter, advanced (every workgroup read it long ago)
    if (!ok && blockIdx.x == 0 && threadIdx.x == 0) {
        // a grid barrier gave up (the grid was not resident): the step's gradients are garbage, say so where the caller looks
        if (q.loss_out) *q.loss_out = __builtin_nanf("");
        q.dgamma[top][0] = __builtin_nanf("");
    }
}

}  // namespace abn
