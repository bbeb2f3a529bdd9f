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
    int k_chunk;            // split-K: slice s owns k in [s*k_chunk, +k_chunk)
    int splits;             // number of k slices (grid = tiles * splits blocks)
    const float* bias;      // FWD: [N] or nullptr
    int act;                // FWD: activation applied; DGRAD: act of `aux`
    const float* aux;       // DGRAD: activation OUTPUT a_prev[M][N] (ld = ldaux)
    int64_t ldaux;
    const float* mask;      // FWD: dropout multiplier [M][N] (ld = ldc) applied before `act`;
                            // DGRAD: multiplier of the layer that produced `aux` (ld = ldaux)
    float* C2;              // WGRAD: destination of the ones column (bias grad)
    int64_t slab_stride;    // WGRAD: floats between split-K slabs of C / C2
    int ones_col;           // WGRAD: column of B that is identically 1, or -1
    int a_vec, b_vec;       // 16-byte global loads allowed for A / B
    int c_vec;              // 16-byte epilogue accesses allowed (C, bias, aux)
    int bf16;               // 0 fp32 MFMA, 1 bf16 operands, 2 bf16 x 3 (split per fragment or, vectorised tiles, per LDS commit)
#ifdef ABN_STAMPS
    unsigned long long* stamps;   // diagnostic build only: [block][128] s_memtime stamps
#endif
};

// sigmoid on the hardware transcendental units: v_exp_f32 and v_rcp_f32 are
// each accurate to ~1 ulp, so the result is within ~3e-7 relative of the IEEE
// expf + division form (parity bar: 1e-5) at a quarter of its instruction
// count -- the forward epilogue evaluates 4.1 M of them per 500-wide layer and
// was spending 6-10 us per layer there (s_memtime stamps).
__device__ __forceinline__ float fast_sigmoid(float z)
{
    // (v_rcp_f32: 1 ulp; the correctly rounded __frcp_rn is a ten-instruction division sequence, a third of this
    // function -- and a sigmoid is read to 1e-5 here)
    return __builtin_amdgcn_rcpf(1.0f + __expf(-z));
}

__device__ __forceinline__ float act_apply(float z, int act)
{
    if (act == ACT_SIGMOID) return fast_sigmoid(z);
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

// Global -> register stage of one operand tile, BRANCH-FREE: every lane issues
// all of its loads unconditionally from clamped (always valid) addresses and
// nothing consumes the results here, so the loads stay in flight behind the
// MFMA phase that follows.  Validity is re-derived and applied in tile_commit,
// after that phase.  (Per-unit `if (m < M)` guards compile to exec-mask
// branches with a vmcnt(0) at every join: measured 2.6x slower.)
// mn0: first row (M/N index) of the tile, k0: first k, k_end: exclusive k bound.
template <int BMN, bool KCONTIG, bool VEC, bool INTERIOR>
__device__ __forceinline__ void tile_issue(f32x4* r, const float* __restrict__ X, int64_t ld,
                                           int MN, int mn0, int k0, int k_end)
{
    constexpr int PT = TileShape<BMN, KCONTIG>::per_thread;
    const int t = threadIdx.x;
    if constexpr (VEC && INTERIOR) {   // whole tile inside the matrix: no clamping at all
        static_assert(!INTERIOR, "interior tiles go through tile_issue_fast");
    } else if constexpr (VEC) {  // units are entirely inside or entirely outside the matrix
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int u = t + 256 * i;
            int m, k;
            if constexpr (KCONTIG) { m = mn0 + (u >> 3); k = k0 + 4 * (u & 7); }
            else { constexpr int UPR = BMN / 4; k = k0 + u / UPR; m = mn0 + 4 * (u % UPR); }
            const bool ok = KCONTIG ? (m < MN && k < k_end) : (k < k_end && m < MN);
            const int64_t off = KCONTIG ? ((int64_t)m * ld + k) : ((int64_t)k * ld + m);
            r[i] = *reinterpret_cast<const f32x4*>(X + (ok ? off : 0));
        }
    } else {
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int u = t + 256 * i;
            int m, k;
            if constexpr (KCONTIG) { m = mn0 + (u >> 3); k = k0 + 4 * (u & 7); }
            else { constexpr int UPR = BMN / 4; k = k0 + u / UPR; m = mn0 + 4 * (u % UPR); }
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int me = KCONTIG ? m : m + e, ke = KCONTIG ? k + e : k;
                const bool ok = me < MN && ke < k_end;
                const int64_t off = KCONTIG ? ((int64_t)me * ld + ke) : ((int64_t)ke * ld + me);
                v[e] = X[ok ? off : 0];
            }
            r[i] = v;
        }
    }
}

// Interior tiles (the common case): the per-lane element offsets inside the tile
// never change, so they are computed ONCE (32-bit: the host checks that every
// operand spans < 2^31 floats) and each k-tile only moves a wave-uniform base
// pointer -- no 64-bit integer VALU work in the loop (it measured ~1400 cycles
// per tile when recomputed each time).
template <int BMN, bool KCONTIG>
__device__ __forceinline__ void tile_offsets(uint32_t* voff, int64_t ld)
{
    constexpr int PT = TileShape<BMN, KCONTIG>::per_thread;
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int u = t + 256 * i;
        if constexpr (KCONTIG) voff[i] = (uint32_t)((u >> 3) * (uint32_t)ld + 4 * (u & 7));
        else { constexpr int UPR = BMN / 4; voff[i] = (uint32_t)((u / UPR) * (uint32_t)ld + 4 * (u % UPR)); }
    }
}

template <int BMN, bool KCONTIG>
__device__ __forceinline__ void tile_issue_fast(f32x4* r, const float* __restrict__ base, const uint32_t* voff)
{
    constexpr int PT = TileShape<BMN, KCONTIG>::per_thread;
#pragma unroll
    for (int i = 0; i < PT; ++i) r[i] = *reinterpret_cast<const f32x4*>(base + voff[i]);
}

// Register -> LDS stage: zero what lies outside the matrix, synthesise the
// all-ones bias column of wgrad, store 16 bytes per unit.
template <int BMN, bool KCONTIG, bool INTERIOR>
__device__ __forceinline__ void tile_commit(const f32x4* r, float* __restrict__ lds, int MN, int mn0,
                                            int k0, int k_end, int ones_col)
{
    constexpr int PT = TileShape<BMN, KCONTIG>::per_thread;
    constexpr int ST = TileShape<BMN, KCONTIG>::stride;
    const int t = threadIdx.x;
    if constexpr (INTERIOR) {
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int u = t + 256 * i;
            int off;
            if constexpr (KCONTIG) off = (u >> 3) * ST + 4 * (u & 7);
            else { constexpr int UPR = BMN / 4; off = (u / UPR) * ST + 4 * (u % UPR); }
            *reinterpret_cast<f32x4*>(lds + off) = r[i];
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int u = t + 256 * i;
        int m, k, off;
        if constexpr (KCONTIG) {
            m = mn0 + (u >> 3); k = k0 + 4 * (u & 7);
            off = (u >> 3) * ST + 4 * (u & 7);
        } else {
            constexpr int UPR = BMN / 4;
            k = k0 + u / UPR; m = mn0 + 4 * (u % UPR);
            off = (u / UPR) * ST + 4 * (u % UPR);
        }
        f32x4 v = r[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int me = KCONTIG ? m : m + e, ke = KCONTIG ? k + e : k;
            float x = (me < MN && ke < k_end) ? v[e] : 0.0f;
            if (!KCONTIG && me == ones_col && ke < k_end) x = 1.0f;   // ones_col = -1: never
            v[e] = x;
        }
        *reinterpret_cast<f32x4*>(lds + off) = v;
    }
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// bf16 x 3: an fp32 value as the sum of three bf16 values (8 + 8 + 8 significant bits):
//   hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid)     (both differences exact in fp32)
// A product x y is then summed from the six bf16 products whose weight reaches 2^-16 of it,
//   x y ~ hi.hi + hi.mid + mid.hi + mid.mid + hi.lo + lo.hi       (each one exact in the fp32 accumulator's
//                                                                  input, dropped terms <= 3 * 2^-24 |x y|),
// on v_mfma_f32_32x32x16_bf16: 6 MFMAs of 32 cycles per 16 k and block instead of 8 of 64.
// The result differs from the exact-fp32 MFMA chain like one fp32 summation order from another
// (~1e-7 relative), far inside the 1e-5 parity bar -- it is NOT bit-identical to it.
struct bf16x8x3 { bf16x8 hi, mid, lo; };
__device__ __forceinline__ bf16x8x3 split_bf16x3(const f32x4& a, const f32x4& b)
{
    bf16x8x3 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float v = i < 4 ? a[i] : b[i - 4];
        const __bf16 h = (__bf16)v;
        const float r1 = v - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        r.hi[i] = h; r.mid[i] = m; r.lo[i] = (__bf16)r2;
    }
    return r;
}

// Two fp32 fragments (k-groups g, g+1: 8 k values per lane) -> one bf16 MFMA operand
// (v_cvt_pk_bf16_f32, round to nearest even).  Which k a lane holds does not matter as
// long as A and B agree: the 32x32x16 MFMA sums over all 16 (lane-half, slot) positions.
__device__ __forceinline__ bf16x8 pack_bf16(const f32x4& a, const f32x4& b)
{
    bf16x8 r;
    r[0] = (__bf16)a[0]; r[1] = (__bf16)a[1]; r[2] = (__bf16)a[2]; r[3] = (__bf16)a[3];
    r[4] = (__bf16)b[0]; r[5] = (__bf16)b[1]; r[6] = (__bf16)b[2]; r[7] = (__bf16)b[3];
    return r;
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


// ---------------------------------------------------------------------------------------------
// PREC 3 = bf16 x 3 with the split done ONCE per tile, when the tile is committed to LDS (PREC 2
// splits at fragment time: every wave re-splits what it reads, and the kernel turns VALU-bound).
// An operand tile [BMN rows][32 k] lives in LDS as rows of 208 bytes: the row's 32 k as bf16 three
// times over -- [hi 64 B | mid 64 B | lo 64 B] -- and 16 bytes of padding.  A lane's MFMA operand
// (8 consecutive k of one row and plane) is one ds_read_b128; 16 consecutive rows start at 16
// distinct multiples of 16 bytes modulo 256 (208 r mod 256), so the reads are conflict-free, and
// rows 4 apart alternate between the two halves of the 128-byte window the writes see (the first
// image -- three separate planes of 64-byte rows -- spent half its LDS cycles in bank conflicts:
// SQ_LDS_BANK_CONFLICT 9.6 M of SQ_LDS_IDX_ACTIVE 19.4 M per launch).
// k-major operands (dgrad's W, both wgrad operands) are transposed on the way in: a thread owns
// 4 (or 2) consecutive k of 4 consecutive rows, splits the values and writes, per row and plane,
// its k as 8 (4) bytes; eight lanes with consecutive k-groups fill one row's 64 bytes.
// ---------------------------------------------------------------------------------------------
constexpr int T3_ROW = 208;              // bytes per row of the image: 3 x 64 + 16
template <int BMN> struct Tile3 {
    static constexpr int plane_bytes = 64;              // hi -> mid -> lo inside a row
    static constexpr int bytes = BMN * T3_ROW;
    static constexpr int units = BMN * BK / 4;
    static constexpr int per_thread = units / 256;
};
__device__ __forceinline__ int t3_off(int row, int chunk) { return row * T3_ROW + (chunk << 4); }

// two fp32 -> (hi, mid, lo) bf16 pairs, each pair in one dword: v_cvt_pk_bf16_f32 converts both at
// once, a bf16 widens back to fp32 by a shift / a mask, and both differences are exact
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& mid, uint32_t& lo)
{
    const f32x2 v = {a, b};
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
    const f32x2 r1 = {a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u)};
    mid = __builtin_bit_cast(uint32_t, __builtin_convertvector(r1, bf16x2));
    const f32x2 r2 = {r1.x - __uint_as_float(mid << 16), r1.y - __uint_as_float(mid & 0xffff0000u)};
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r2, bf16x2));
}

// four fp32 -> 4 x (hi, mid, lo) bf16, each quadruple packed into 8 bytes
__device__ __forceinline__ void split4(const float v0, const float v1, const float v2, const float v3, uint2& hi, uint2& mid,
                                       uint2& lo)
{
    split2(v0, v1, hi.x, mid.x, lo.x);
    split2(v2, v3, hi.y, mid.y, lo.y);
}

// unit -> (row, k) of a thread's i-th float4.  K-contiguous: the fp32 path's map.  k-major: a
// thread's units are 4 (BMN = 128) or 2 (BMN = 64) CONSECUTIVE k of the same 4 rows.
template <int BMN, bool KCONTIG>
__device__ __forceinline__ void unit3(int t, int i, int& row, int& k)
{
    if constexpr (KCONTIG) { const int u = t + 256 * i; row = u >> 3; k = 4 * (u & 7); }
    else {
        constexpr int PT = Tile3<BMN>::per_thread, KG = BK / PT;        // k-groups of PT along the tile's 32 k
        row = 4 * (t / KG);                                             // consecutive lanes: consecutive k-groups of
        k = PT * (t % KG) + i;                                          // the same 4 rows (one row's bytes side by side)
    }
}

template <int BMN, bool KCONTIG>
__device__ __forceinline__ void tile3_offsets(uint32_t* voff, int64_t ld)
{
    constexpr int PT = Tile3<BMN>::per_thread;
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        int row, k;
        unit3<BMN, KCONTIG>((int)threadIdx.x, i, row, k);
        voff[i] = KCONTIG ? (uint32_t)(row * (uint32_t)ld + k) : (uint32_t)(k * (uint32_t)ld + row);
    }
}

// edge tiles: clamped loads (always valid addresses), validity applied at commit
template <int BMN, bool KCONTIG>
__device__ __forceinline__ void tile3_issue(f32x4* r, const float* __restrict__ X, int64_t ld, int MN, int mn0, int k0, int k_end)
{
    constexpr int PT = Tile3<BMN>::per_thread;
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        int row, k;
        unit3<BMN, KCONTIG>((int)threadIdx.x, i, row, k);
        const int m = mn0 + row, kk = k0 + k;
        const bool ok = m < MN && kk < k_end;          // 16-byte units are entirely inside or outside (VEC)
        const int64_t off = KCONTIG ? ((int64_t)m * ld + kk) : ((int64_t)kk * ld + m);
        r[i] = *reinterpret_cast<const f32x4*>(X + (ok ? off : 0));
    }
}

template <int BMN, bool KCONTIG, bool INTERIOR>
__device__ __forceinline__ void tile3_commit(const f32x4* r, char* __restrict__ lds, int MN, int mn0, int k0, int k_end, int ones_col)
{
    constexpr int PT = Tile3<BMN>::per_thread, PB = Tile3<BMN>::plane_bytes;
    const int t = threadIdx.x;
    if constexpr (KCONTIG) {
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            int row, k;
            unit3<BMN, true>(t, i, row, k);
            f32x4 v = r[i];
            if constexpr (!INTERIOR) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (mn0 + row < MN && k0 + k + e < k_end) ? v[e] : 0.0f;
            }
            uint2 hi, mid, lo;
            split4(v[0], v[1], v[2], v[3], hi, mid, lo);
            char* dst = lds + t3_off(row, k >> 3) + (k & 4) * 2;
            *reinterpret_cast<uint2*>(dst) = hi;
            *reinterpret_cast<uint2*>(dst + PB) = mid;
            *reinterpret_cast<uint2*>(dst + 2 * PB) = lo;
        }
    } else {
        int row, k;
        unit3<BMN, false>(t, 0, row, k);            // k of unit 0; units 1 .. PT-1 follow at k + 1 ...
        f32x4 v[PT];
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            v[i] = r[i];
            if constexpr (!INTERIOR) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int me = mn0 + row + e, ke = k0 + k + i;
                    float x = (me < MN && ke < k_end) ? v[i][e] : 0.0f;
                    if (me == ones_col && ke < k_end) x = 1.0f;       // wgrad's bias column (ones_col = -1: never)
                    v[i][e] = x;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {               // row (row + e): PT consecutive k
            if constexpr (PT == 4) {
                uint2 hi, mid, lo;
                split4(v[0][e], v[1][e], v[2][e], v[3][e], hi, mid, lo);
                char* dst = lds + t3_off(row + e, k >> 3) + (k & 4) * 2;
                *reinterpret_cast<uint2*>(dst) = hi;
                *reinterpret_cast<uint2*>(dst + PB) = mid;
                *reinterpret_cast<uint2*>(dst + 2 * PB) = lo;
            } else {
                static_assert(PT == 2 || PT == 4, "tile3: 64- or 128-row tiles");
                uint2 hi, mid, lo;
                split4(v[0][e], v[1][e], 0.0f, 0.0f, hi, mid, lo);
                char* dst = lds + t3_off(row + e, k >> 3) + (k & 7) * 2;
                *reinterpret_cast<uint32_t*>(dst) = hi.x;
                *reinterpret_cast<uint32_t*>(dst + PB) = mid.x;
                *reinterpret_cast<uint32_t*>(dst + 2 * PB) = lo.x;
            }
        }
    }
}

// MFMA operand of row block `row0` for the 16-deep slab `sl` of the tile: lane (r, h) takes k = 16 sl + 8 h .. + 7
__device__ __forceinline__ bf16x8 frag3(const char* __restrict__ plane, int row0, int sl, int lane)
{
    const int r = row0 + (lane & 31);
    return *reinterpret_cast<const bf16x8*>(plane + t3_off(r, 2 * sl + (lane >> 5)));
}

template <int BM, int BN>
constexpr size_t gemm_lds_bytes3()
{
    constexpr size_t tiles = 2 * (size_t)(Tile3<BM>::bytes + Tile3<BN>::bytes);
    constexpr size_t stage = sizeof(float) * BM * (BN + 4);
    return tiles > stage ? tiles : stage;
}

// XCD-aware tile order: blocks b, b+8, b+16, ... share an XCD (and its L2), so
// hand XCD x the contiguous tile range [x*n/8, (x+1)*n/8) when n % 8 == 0.
__device__ __forceinline__ int xcd_tile_index(int b, int n)
{
    if ((n & 7) != 0) return b;
    return (b & 7) * (n >> 3) + (b >> 3);
}

#ifdef ABN_STAMPS
// stamps stay in SGPR/LDS-free registers of lane 0 until the end of the kernel
#define ABN_STAMP()                                                                          \
    do {                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        unsigned long long t_;                                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        if (nst < 80) stbuf[nst] = t_;                                                       \
        ++nst;                                                                               \
    } while (0)
#define ABN_STAMP_FLUSH()                                                                    \
    do {                                                                                     \
        if (p.stamps && threadIdx.x == 0)                                                    \
            for (int q_ = 0; q_ < 80; ++q_) p.stamps[(size_t)block_id * 128 + q_] = q_ < nst ? stbuf[q_] : 0; \
    } while (0)
#else
#define ABN_STAMP() do {} while (0)
#define ABN_STAMP_FLUSH() do {} while (0)
#endif

// BF16 = the opt-in throughput mode: same tiles, loaders, LDS image (fp32) and
// epilogues; only the inner product changes -- pairs of k-groups are rounded to bf16
// and fed to v_mfma_f32_32x32x16_bf16 (fp32 accumulate), 2 MFMAs of 32 cycles per
// 32-deep tile and block instead of 16 of 64.  NOT the parity path (~3 significant
// digits); the kernel is then bound by its global -> LDS traffic, not by the MFMA.
template <int BM, int BN, bool A_KC, bool B_KC, int EPI, bool VEC, int BF16 = 0>
__device__ __forceinline__ void gemm_body(const GemmP& p, const int block_id)
{
#ifdef ABN_STAMPS
    int nst = 0;
    unsigned long long stbuf[80];
#endif
    ABN_STAMP();
    static_assert(BM % 64 == 0 && BN % 64 == 0, "tile must split over 2x2 waves of 32x32 MFMAs");
    constexpr int TM = BM / 64, TN = BN / 64;
    using TA = TileShape<BM, A_KC>;
    using TB = TileShape<BN, B_KC>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As = smem;                       // two A stages, then two B stages
    float* const Bs = smem + 2 * TA::floats;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    // Block -> (tile, k-slice).  Blocks b, b+8, ... share an XCD (its L2).  Without
    // split-K every XCD gets a contiguous run of tiles (neighbours share A rows).
    // With split-K an XCD owns whole k-SLICES instead, for all tiles: its L2 then
    // fetches each operand row of its slices from HBM exactly once (measured
    // FETCH_SIZE of the 500x500 wgrad: 113 MB tile-major vs 33 MB algorithmic).
    const int ntiles = tiles_m * tiles_n;
    int tile, slice;
    if (p.splits > 1 && (p.splits & 7) == 0) {
        const int xcd = block_id & 7, q = block_id >> 3, spx = p.splits >> 3;
        slice = xcd * spx + q / ntiles;
        tile = q % ntiles;
    } else if (p.splits > 1) {
        slice = block_id / ntiles;
        tile = block_id % ntiles;
    } else {
        slice = 0;
        tile = xcd_tile_index(block_id, ntiles);
    }
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int kb = slice * p.k_chunk;
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

    if constexpr (BF16 == 3) {
        // ---- bf16 x 3, split at commit time (see Tile3) ----
        static_assert(VEC, "the plane image is filled from 16-byte units");
        char* const cs = reinterpret_cast<char*>(smem);
        constexpr int SA = Tile3<BM>::bytes, SB = Tile3<BN>::bytes, STG = SA + SB;
        constexpr int PA = Tile3<BM>::per_thread, PBt = Tile3<BN>::per_thread;
        // TWO register stages: with six 32-cycle MFMAs per block and slab a tile's matrix work
        // (~770 cycles) is shorter than a global load's latency, so the loads run two tiles
        // ahead of the MFMAs (one tile in LDS, the next in registers, the one after in flight).
        f32x4 ra[2][PA], rb[2][PBt];
        const int ones = (EPI == EPI_WGRAD) ? p.ones_col : -1;
        const int nB = (ones >= 0) ? ones : p.N;
        const bool a_in = m0 + BM <= p.M, b_in = n0 + BN <= nB;
        uint32_t voa[PA], vob[PBt];
        tile3_offsets<BM, A_KC>(voa, p.lda);
        tile3_offsets<BN, B_KC>(vob, p.ldb);
        const float* const a_org = p.A + (A_KC ? (int64_t)m0 * p.lda : (int64_t)m0);
        const float* const b_org = p.B + (B_KC ? (int64_t)n0 * p.ldb : (int64_t)n0);
        auto issue = [&](int k0, f32x4* qa, f32x4* qb) {
            const bool k_in = k0 + BK <= ke;
            if (a_in && k_in) {
                const float* an = a_org + (A_KC ? (int64_t)k0 : (int64_t)k0 * p.lda);
#pragma unroll
                for (int i = 0; i < PA; ++i) qa[i] = *reinterpret_cast<const f32x4*>(an + voa[i]);
            } else tile3_issue<BM, A_KC>(qa, p.A, p.lda, p.M, m0, k0, ke);
            if (b_in && k_in) {
                const float* bn = b_org + (B_KC ? (int64_t)k0 : (int64_t)k0 * p.ldb);
#pragma unroll
                for (int i = 0; i < PBt; ++i) qb[i] = *reinterpret_cast<const f32x4*>(bn + vob[i]);
            } else tile3_issue<BN, B_KC>(qb, p.B, p.ldb, nB, n0, k0, ke);
        };
        auto commit = [&](int k0, char* st, const f32x4* qa, const f32x4* qb) {
            const bool k_in = k0 + BK <= ke;
            if (a_in && k_in) tile3_commit<BM, A_KC, true>(qa, st, p.M, m0, k0, ke, -1);
            else tile3_commit<BM, A_KC, false>(qa, st, p.M, m0, k0, ke, -1);
            if (b_in && k_in) tile3_commit<BN, B_KC, true>(qb, st + SA, nB, n0, k0, ke, ones);
            else tile3_commit<BN, B_KC, false>(qb, st + SA, nB, n0, k0, ke, ones);
        };
        auto compute = [&](const char* as, const char* bs) {
#pragma unroll
            for (int sl = 0; sl < BK / 16; ++sl) {
                bf16x8 fa[TM][3], fb[TN][3];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) fa[i][pl] = frag3(as + pl * Tile3<BM>::plane_bytes, wm0 + 32 * i, sl, lane);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) fb[j][pl] = frag3(bs + pl * Tile3<BN>::plane_bytes, wn0 + 32 * j, sl, lane);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {       // small terms first
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], acc[i][j], 0, 0, 0);
                    }
            }
        };
        if (nkt > 0) {
            issue(kb, ra[0], rb[0]);
            if (nkt > 1) issue(kb + BK, ra[1], rb[1]);
            commit(kb, cs, ra[0], rb[0]);
        }
        __syncthreads();
        // iteration kt: tile kt in LDS stage kt & 1, tile kt+1 in register set (kt+1) & 1; the loads of
        // tile kt+2 go into register set kt & 1 (free since its tile was committed).  Unrolled by two
        // so that the register sets are addressed statically.
        for (int kt = 0; kt < nkt; kt += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = kt + h;
                if (k >= nkt) break;
                const char* as = cs + h * STG;
#ifndef ABN_EXP_NOLOAD
                if (k + 2 < nkt) issue(kb + (k + 2) * BK, ra[h], rb[h]);
#endif
#ifndef ABN_EXP_NOMFMA
                compute(as, as + SA);
#endif
#ifndef ABN_EXP_NOCOMMIT
                if (k + 1 < nkt) commit(kb + (k + 1) * BK, cs + (h ^ 1) * STG, ra[h ^ 1], rb[h ^ 1]);
#endif
                __syncthreads();
            }
        }
    } else {
    f32x4 ra[TA::per_thread], rb[TB::per_thread];
    const int ones = (EPI == EPI_WGRAD) ? p.ones_col : -1;
    // the ones column is synthesised, never read: B really has `ones` columns
    const int nB = (ones >= 0) ? ones : p.N;
    // wave-uniform "this operand tile lies fully inside the matrix" flags pick a
    // clamp-free issue and a mask-free commit (the common case: 15 of 16 k-tiles)
    const bool a_in = VEC && (m0 + BM <= p.M), b_in = VEC && (n0 + BN <= nB);
    uint32_t voa[TA::per_thread], vob[TB::per_thread];
    tile_offsets<BM, A_KC>(voa, p.lda);
    tile_offsets<BN, B_KC>(vob, p.ldb);
    // element (mn0, k = 0) of each operand; a k-tile adds k0 (K-contig) or k0*ld
    const float* const a_org = p.A + (A_KC ? (int64_t)m0 * p.lda : (int64_t)m0);
    const float* const b_org = p.B + (B_KC ? (int64_t)n0 * p.ldb : (int64_t)n0);
    auto issue = [&](int k0) {
        const bool k_in = k0 + BK <= ke;
        if (a_in && k_in) tile_issue_fast<BM, A_KC>(ra, a_org + (A_KC ? (int64_t)k0 : (int64_t)k0 * p.lda), voa);
        else tile_issue<BM, A_KC, VEC, false>(ra, p.A, p.lda, p.M, m0, k0, ke);
        if (b_in && k_in) tile_issue_fast<BN, B_KC>(rb, b_org + (B_KC ? (int64_t)k0 : (int64_t)k0 * p.ldb), vob);
        else tile_issue<BN, B_KC, VEC, false>(rb, p.B, p.ldb, nB, n0, k0, ke);
    };
    auto commit = [&](int k0, float* as, float* bs) {
        const bool k_in = k0 + BK <= ke;
        if (a_in && k_in) tile_commit<BM, A_KC, true>(ra, as, p.M, m0, k0, ke, -1);
        else tile_commit<BM, A_KC, false>(ra, as, p.M, m0, k0, ke, -1);
        if (b_in && k_in) tile_commit<BN, B_KC, true>(rb, bs, nB, n0, k0, ke, ones);
        else tile_commit<BN, B_KC, false>(rb, bs, nB, n0, k0, ke, ones);
    };
    if (nkt > 0) {
        issue(kb);
        commit(kb, As, Bs);
    }
    __syncthreads();
    ABN_STAMP();

    auto k_group = [&](const float* as, const float* bs, int g) {
        f32x4 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = frag_read<BM, A_KC>(as, wm0 + 32 * i, g, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = frag_read<BN, B_KC>(bs, wn0 + 32 * j, g, lane);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
    };

    auto k_group_bf16 = [&](const float* as, const float* bs, int g2) {
        if constexpr (BF16 == 2) {               // bf16 x 3: fp32-grade products from six bf16 MFMAs
            bf16x8x3 pa[TM], pb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                pa[i] = split_bf16x3(frag_read<BM, A_KC>(as, wm0 + 32 * i, 2 * g2, lane),
                                     frag_read<BM, A_KC>(as, wm0 + 32 * i, 2 * g2 + 1, lane));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                pb[j] = split_bf16x3(frag_read<BN, B_KC>(bs, wn0 + 32 * j, 2 * g2, lane),
                                     frag_read<BN, B_KC>(bs, wn0 + 32 * j, 2 * g2 + 1, lane));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {   // small terms first
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[i].lo, pb[j].hi, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[i].hi, pb[j].lo, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[i].mid, pb[j].mid, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[i].mid, pb[j].hi, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[i].hi, pb[j].mid, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[i].hi, pb[j].hi, acc[i][j], 0, 0, 0);
                }
        } else {
        bf16x8 pa[TM], pb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
            pa[i] = pack_bf16(frag_read<BM, A_KC>(as, wm0 + 32 * i, 2 * g2, lane),
                              frag_read<BM, A_KC>(as, wm0 + 32 * i, 2 * g2 + 1, lane));
#pragma unroll
        for (int j = 0; j < TN; ++j)
            pb[j] = pack_bf16(frag_read<BN, B_KC>(bs, wn0 + 32 * j, 2 * g2, lane),
                              frag_read<BN, B_KC>(bs, wn0 + 32 * j, 2 * g2 + 1, lane));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[i], pb[j], acc[i][j], 0, 0, 0);
        }
    };

    // One global load of the next tile (unit u: A units first, then B units).
    auto issue_unit = [&](int u, const float* an, const float* bn) {
        if (u < TA::per_thread) ra[u] = *reinterpret_cast<const f32x4*>(an + voa[u]);
        else rb[u - TA::per_thread] = *reinterpret_cast<const f32x4*>(bn + vob[u - TA::per_thread]);
    };
    constexpr int UNITS = TA::per_thread + TB::per_thread;
    constexpr int UPG = (UNITS + 1) / 2;         // loads issued behind each of the first two k-groups

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nkt;
        const int knext = kb + (kt + 1) * BK;
        const float* as = As + cur * TA::floats;
        const float* bs = Bs + cur * TB::floats;
        // Interior next tile (the common case): its loads are NOT issued in one
        // burst at the top -- all 8 waves of a CU doing that right after the
        // barrier queue ~50 KB on the CU's address unit and every wave's MFMA
        // phase starts 800-1700 cycles late (s_memtime stamps) -- but spread
        // behind the first two k-groups' MFMAs.  Edge tiles keep the simple order.
        const bool fast = more && a_in && b_in && (knext + BK <= ke);
        if (more && !fast) issue(knext);
        ABN_STAMP();
        const float* an = a_org + (A_KC ? (int64_t)knext : (int64_t)knext * p.lda);
        const float* bn = b_org + (B_KC ? (int64_t)knext : (int64_t)knext * p.ldb);
        // tiles are zero-filled past k_end, so every tile runs all BK/8 groups
        // (at most 31 wasted k per GEMM) and the loop body stays branch-free
        if constexpr (BF16) {
#pragma unroll
            for (int g2 = 0; g2 < BK / 16; ++g2) {
                k_group_bf16(as, bs, g2);
                if (fast) {
#pragma unroll
                    for (int u = g2 * UPG; u < (g2 + 1) * UPG && u < UNITS; ++u) issue_unit(u, an, bn);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            k_group(as, bs, g);
            if (g < 2 && fast) {
#pragma unroll
                for (int u = g * UPG; u < (g + 1) * UPG && u < UNITS; ++u) issue_unit(u, an, bn);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        ABN_STAMP();
        if (more) commit(knext, As + (cur ^ 1) * TA::floats, Bs + (cur ^ 1) * TB::floats);
        ABN_STAMP();
        __syncthreads();
        ABN_STAMP();
    }

    }
    // Epilogue.  The accumulators go through LDS (free after the k loop) so that
    // every global access of the epilogue -- C, the bias, the dgrad's saved
    // activations -- is a full-row 16-byte access: storing straight from the
    // MFMA layout (32 lanes x 4 B per row segment) measured 10 us per 16 MB
    // output, this form ~4.  Accumulator register r of lane l holds
    //   row = (r&3) + 8*(r>>2) + 4*(l>>5), col = l&31   of its 32x32 block.
    constexpr int SST = BN + 4;
    float* const stage = smem;
    {
        const int col_l = lane & 31, rsub = 4 * (lane >> 5);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    stage[(wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + rsub) * SST + wn0 + 32 * j + col_l] = acc[i][j][r];
    }
    __syncthreads();
    ABN_STAMP();
    constexpr int C4 = BN / 4;
    const int n_plain = (EPI == EPI_WGRAD && p.ones_col >= 0) ? p.ones_col : p.N;   // columns that live in C
    const int64_t slab = (EPI == EPI_WGRAD) ? (int64_t)slice * p.slab_stride : 0;
#pragma unroll 2
    for (int u = threadIdx.x; u < BM * C4; u += 256) {
        const int row = u / C4, c4 = u % C4;
        const int m = m0 + row, n = n0 + 4 * c4;
        if (m >= p.M || n >= p.N) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * SST + 4 * c4);
        if (p.c_vec && n + 3 < n_plain) {
            if constexpr (EPI == EPI_FWD) {
                if (p.bias) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n);
                    v += bv;
                }
                if (p.mask) v *= *reinterpret_cast<const f32x4*>(p.mask + (int64_t)m * p.ldc + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], p.act);
            } else if constexpr (EPI == EPI_DGRAD) {
                if (p.aux) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(p.aux + (int64_t)m * p.ldaux + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= act_grad(av[e], p.act);
                }
                if (p.mask) v *= *reinterpret_cast<const f32x4*>(p.mask + (int64_t)m * p.ldaux + n);
            }
            *reinterpret_cast<f32x4*>(p.C + slab + (int64_t)m * p.ldc + n) = v;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ne = n + e;
                if (ne >= p.N) break;
                float x = v[e];
                if constexpr (EPI == EPI_FWD) {
                    x += p.bias ? p.bias[ne] : 0.0f;
                    if (p.mask) x *= p.mask[(int64_t)m * p.ldc + ne];
                    p.C[(int64_t)m * p.ldc + ne] = act_apply(x, p.act);
                } else if constexpr (EPI == EPI_DGRAD) {
                    if (p.aux) x *= act_grad(p.aux[(int64_t)m * p.ldaux + ne], p.act);
                    if (p.mask) x *= p.mask[(int64_t)m * p.ldaux + ne];
                    p.C[(int64_t)m * p.ldc + ne] = x;
                } else {
                    if (ne == p.ones_col) p.C2[slab + m] = x;
                    else p.C[slab + (int64_t)m * p.ldc + ne] = x;
                }
            }
        }
    }
    ABN_STAMP();
    ABN_STAMP_FLUSH();
}

template <int BM, int BN, bool A_KC, bool B_KC, int EPI, bool VEC, int BF16 = 0>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmP p)
{
    gemm_body<BM, BN, A_KC, B_KC, EPI, VEC, BF16>(p, (int)blockIdx.x);
}

// One launch for the two GEMMs of a backward layer, which both consume dz_l and are
// independent of each other: workgroups [0, n0) are the wgrad (dW_l = dz_l^T a_{l-1},
// split-K slabs), the rest the dgrad (d a_{l-1} = dz_l W_l).  Launched back to back, the
// second kernel cannot start before the first one's last workgroup has drained its
// epilogue; in one grid its workgroups take over the CUs as they free up.  n0 must be a
// multiple of 8 so that the XCD-affine tile maps of both parts still see `b & 7` = XCD.
template <int WM, int WN, int BF16>
__global__ __launch_bounds__(256) void gemm_bwd_pair_kernel(GemmP pw, int n0, GemmP pd)
{
    if ((int)blockIdx.x < n0) gemm_body<WM, WN, false, false, EPI_WGRAD, true, BF16>(pw, (int)blockIdx.x);
    else gemm_body<128, 64, true, false, EPI_DGRAD, true, BF16>(pd, (int)blockIdx.x - n0);
}

template <int BM, int BN, bool A_KC, bool B_KC>
constexpr size_t gemm_lds_bytes()
{
    constexpr size_t tiles = sizeof(float) * 2 * (TileShape<BM, A_KC>::floats + TileShape<BN, B_KC>::floats);
    constexpr size_t stage = sizeof(float) * BM * (BN + 4);       // epilogue staging tile
    return tiles > stage ? tiles : stage;
}

}  // namespace abn
