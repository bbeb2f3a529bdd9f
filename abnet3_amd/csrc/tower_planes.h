// tower_planes.h -- the tower forward AND backward on 16-bit MFMA operands that are split ONCE, widths <= 512:
// the template parameter NP is the number of operand planes -- 2: fp16 x 2 (abn_tower_desc.precision 3, the
// default: each fp32 operand times a power of two, split into hi + lo fp16, three v_mfma_f32_32x32x16_f16 per
// operand pair, the scales taken out again in the epilogues), 3: bf16 x 3 (precision 2: hi + mid + lo, six
// v_mfma_f32_32x32x16_bf16), 1: plain bf16 (precision 1).  Without BatchNorm: one launch for the forward, one for
// the data-gradient chain, one for every layer's weight gradient.  With BatchNorm: inference in the same single
// launch (running statistics in the epilogue); training one launch per layer each way (bn_fwd_layer_kernel,
// bn_bwd_layer_kernel), the same weight-gradient launch.  tower_wide.h holds the layer-per-launch kernels for
// small batches on the same operand images.
//
// What the stamps and the knock-out builds said about tower_fused.h / gemm_f32.h:
// the forward's floor was the per-wave weight stream (LDS-DMA of 64-byte row pieces: 80 us with
// every MFMA and fragment read removed), and in the bf16 x 3 arithmetic every wave re-split the same
// fp32 fragments on the VALU (activations 8 times per workgroup, weights 256 times per step).
// Here nothing on a k-loop touches the VALU:
//
//   * pack_planes_kernel turns each weight matrix (and its transpose, for the backward) into MFMA
//     operand fragments once per step:
//       image[block of 32 operand rows][step of 16 along the sum][plane hi|mid|lo][lane][8 bf16]
//     1 KB per (block, step, plane), consecutive steps contiguous: a wave streams its block's
//     weights with one 16-byte load per lane, plane and step STRAIGHT INTO REGISTERS (whole cache
//     lines, no LDS round trip; the ring of PL_DEPTH steps lives in VGPRs);
//   * the chain products are transposed, Y^T = W X^T and dZ_in^T = W^T dZ_out^T: the weights are the
//     MFMA's A operand, the 32 batch rows of the workgroup its B operand.  The accumulator then has
//     the batch row on the lane and 16 features in its registers -- which IS the B-operand layout
//     of the next product of the chain (guide: "an accumulator tile as the next MFMA's operand",
//     k order 16s + 8(j>>2) + 4h + (j&3) inside a step; the packed weights carry the same
//     permutation).  The epilogue splits each value once and writes whole fragments, 16 bytes per
//     lane and plane, into the LDS image  img[step][plane][lane][8 bf16]  that all eight waves read
//     back lane-linearly (conflict free);
//   * the weight gradient dW = dZ^T A sums over the batch rows, the lane index of both chains'
//     accumulators.  Each chain epilogue therefore also hands its tile to the MFMA once more, as
//     the A operand against a (permuted) identity: the result is the tile transposed -- feature on
//     the lane, batch rows in the registers, exact, because every plane is a bf16 value times 1 --
//     and goes to HBM in operand-fragment order
//       timage[block of 32 features][step of 16 batch rows][lane][8 values],
//     the three planes added up again (exactly) to one fp32 value per element for bf16 x 3 (4 bytes
//     instead of 6: these images are the step's largest HBM streams -- they are also the ONLY copy
//     of the hidden activations: the data-gradient chain gathers act'(a) from them), one bf16 for
//     the bf16 mode.  wgrad_planes_kernel streams both operands in that form (bf16: LDS-DMA,
//     lane-linear; bf16 x 3: fetched into registers, split once by the fetching wave, planes into an LDS
//     ring) and otherwise only issues MFMAs.  A column of ones appended to the activation image yields
//     the bias gradient.  A slab's tiles are placed on one XCD (wgrad_group_count): its L2 serves the re-reads.
#pragma once
#include <type_traits>

#include "gemm_f32.h"

namespace abn {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int PL_ROWS = 32;
constexpr int PL_WAVES = 8;
constexpr int PL_NT = 64 * PL_WAVES;
constexpr int PL_MAXW = 512;
constexpr int PL_MAXSTEPS = PL_MAXW / 16;
#ifdef PL_DEPTH_OVERRIDE
constexpr int PL_DEPTH = PL_DEPTH_OVERRIDE;
#else
constexpr int PL_DEPTH = 4;                       // weight steps in flight per wave (registers)
#endif
constexpr int PL_PART_BYTES = 4 * 16 * 64 * 4;    // K-split hand-over: 4 waves x one accumulator block
// fp16 x 2 (NP == 2): per-lane partial maxima of the workgroup's 32 rows [8 waves][64 lanes], then one table of
// the rows' inverse scales per wave [8][32]
constexpr int PL_SC_BYTES = PL_WAVES * 64 * 4 + PL_WAVES * 32 * 4;
constexpr int PL_AMAX = 32;                       // floats per 32-row block in an image's table of maxima (wgrad's scales)

__host__ __device__ inline int pl_steps(int64_t contraction) { return ((int)((contraction + 15) / 16) + PL_DEPTH - 1) / PL_DEPTH * PL_DEPTH; }
__host__ __device__ inline int pl_blocks(int64_t features) { return (int)((features + 31) / 32); }
// (np == 2: one inverse scale per 32-row block behind the fragments)
static inline int64_t pl_image_bytes(int64_t features, int64_t contraction, int np)
{
    return (int64_t)pl_blocks(features) * pl_steps(contraction) * np * 1024 + (np == 2 ? (pl_blocks(features) * 4 + 255) / 256 * 256 : 0);
}
// batch-row steps of the transposed (weight-gradient) images: two per 32-row workgroup
static inline int64_t pl_row_steps(int64_t rows) { return (rows + PL_ROWS - 1) / PL_ROWS * 2; }
// bytes of one (block, row step) tile of a transposed image: 64 lanes x 8 fp32 (bf16 x 3, fp16 x 2) or 8 bf16 (bf16)
template <int NP> constexpr int tile_bytes() { return NP >= 2 ? 2048 : 1024; }
static inline int64_t pl_timage_bytes(int64_t features, int64_t rows, int np) { return (int64_t)pl_blocks(features) * pl_row_steps(rows) * (np >= 2 ? 2048 : 1024); }
// floats of an image's table of maxima: PL_AMAX per 32-row block
static inline int64_t pl_amax_floats(int64_t rows) { return pl_row_steps(rows) / 2 * PL_AMAX; }
constexpr int PL_BIAS_BYTES = PL_WAVES * 64 * 4;   // a wave's 64 bias values, parked in LDS across its k-loop
__host__ __device__ inline size_t pl_lds_bytes(int np) { return (size_t)PL_MAXSTEPS * np * 1024 + PL_PART_BYTES + PL_SC_BYTES + PL_BIAS_BYTES; }
// the inference forward of a BatchNorm tower parks four more per-feature vectors beside the bias
static inline size_t pl_lds_bytes_bn(int np) { return pl_lds_bytes(np) + 4 * PL_BIAS_BYTES; }

#ifdef ABN_STAMPS
#define PSTAMPF(slot) do { if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 128 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
// (s_memtime counts from a base of its own per CU group: a stamp that is compared BETWEEN workgroups takes the constant 100 MHz clock)
#define PSTAMPR(slot) do { if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 128 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
// every WAVE's clock at the end of layer l's k-loop: slots 64 + 8 l + wave (tools/planes_wave_stamps.py)
#define PSTAMPW(l) do { if (p.stamps && (threadIdx.x & 63) == 0 && (l) < 8) p.stamps[(size_t)blockIdx.x * 128 + 64 + 8 * (l) + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PSTAMPW(l) do {} while (0)
#define PSTAMPF(slot) do {} while (0)
#define PSTAMPR(slot) do {} while (0)
#endif

// ---------------------------------------------------------------------------------------------
// fragments
// ---------------------------------------------------------------------------------------------
template <int NP> struct Frag { bf16x8 p[NP]; };        // (NP == 2: the 16 bytes are eight fp16 values)

// fp16 x 2 (NP == 2): an fp32 value u = s x (s a power of two that puts the largest |x| of the operand row in
// [2^14, 2^15)) as hi + lo, hi = fp16(u), lo = fp16(u - hi): 22 significant bits, the difference exact in fp32.
// A product is summed from lo.hi + hi.lo + hi.hi (dropped: lo.lo <= 2^-22 of it) on v_mfma_f32_32x32x16_f16 --
// three MFMAs per 16 k and block where bf16 x 3 issues six, two planes to stream instead of three -- and the
// accumulator is multiplied by the two inverse scales (exact) in the epilogue.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// scale of an operand row whose largest magnitude is m (>= 0), and its inverse
__device__ __forceinline__ void scale_of(float m, float& s, float& inv)
{
    unsigned e = __float_as_uint(m) >> 23;
    e = e > 240u ? 240u : e;
    const bool tiny = e < 27u;                          // zero rows, and magnitudes whose products are below fp32 anyway
    s = tiny ? 1.0f : __uint_as_float((268u - e) << 23);
    inv = tiny ? 1.0f : __uint_as_float((e - 14u) << 23);
}

// Two values at once, two instructions per value: hi = fp16(s x) is ONE v_fma_mix (the product s x is exact -- s is a
// power of two -- so the fused multiply-add rounds once, to fp16: the same number as (_Float16)(s * x)), lo =
// fp16(s x - hi) another one (the difference is exact in fp32), both written straight into their half of the packed
// register.  (The compiler finds this form for some of the call sites and a seven-instruction one -- multiply,
// packed convert, two conversions back, packed fma, packed convert -- for others; the weight-gradient launch, whose
// vector instructions compete with its MFMAs for the same issue cycles, converts twelve values per lane and step.)
typedef uint32_t u32x4p __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split_f16x2_pair(float s, float a, float b, uint32_t& hi, uint32_t& lo)
{
    uint32_t h, l;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(s), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(s), "v"(b));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(s), "v"(a), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(s), "v"(b), "v"(h));
    hi = h; lo = l;
}

template <int NP>
__device__ __forceinline__ f32x16 pl_mfma(const bf16x8& a, const bf16x8& b, const f32x16& c)
{
    if constexpr (NP == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// the products of one k-step, smallest terms first: plane of the A operand, plane of the B operand
template <int NP> struct Products;
template <> struct Products<1> { static constexpr int N = 1; static constexpr int A[1] = {0}, B[1] = {0}; };
template <> struct Products<2> { static constexpr int N = 3; static constexpr int A[3] = {1, 0, 0}, B[3] = {0, 1, 0}; };
template <> struct Products<3> { static constexpr int N = 6; static constexpr int A[6] = {2, 0, 1, 1, 0, 0}, B[6] = {0, 2, 1, 0, 1, 0}; };

template <int NP>
__device__ __forceinline__ Frag<NP> make_frag(const f32x4& v0, const f32x4& v1, float scale = 1.0f)
{
    Frag<NP> f;
    if constexpr (NP == 3) {
        const bf16x8x3 s = split_bf16x3(v0, v1);
        f.p[0] = s.hi; f.p[1] = s.mid; f.p[2] = s.lo;
    } else if constexpr (NP == 2) {
        u32x4p hi, lo;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a = i < 2 ? v0[2 * i] : v1[2 * i - 4], b = i < 2 ? v0[2 * i + 1] : v1[2 * i - 3];
            uint32_t h, l;
            split_f16x2_pair(scale, a, b, h, l);
            hi[i] = h; lo[i] = l;
        }
        f.p[0] = __builtin_bit_cast(bf16x8, hi); f.p[1] = __builtin_bit_cast(bf16x8, lo);
    } else {
        f.p[0] = pack_bf16(v0, v1);
    }
    return f;
}

template <int NP>
__device__ __forceinline__ void store_frag(char* dst, const Frag<NP>& f)
{
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<bf16x8*>(dst + pl * 1024) = f.p[pl];
}

template <int NP>
__device__ __forceinline__ void write_frag(char* dst, const f32x4& v0, const f32x4& v1, float scale = 1.0f)
{
    store_frag<NP>(dst, make_frag<NP>(v0, v1, scale));
}

// largest magnitude of a lane's eight values
__device__ __forceinline__ float absmax8(const f32x4& v0, const f32x4& v1)
{
    float m = fmaxf(fmaxf(fabsf(v0[0]), fabsf(v0[1])), fmaxf(fabsf(v0[2]), fabsf(v0[3])));
    return fmaxf(m, fmaxf(fmaxf(fabsf(v1[0]), fabsf(v1[1])), fmaxf(fabsf(v1[2]), fabsf(v1[3]))));
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// The fp16 x 2 scales of a workgroup's 32 rows.  Every lane has parked the largest magnitude it holds of row
// lane & 31 in sc[wave][lane] (a workgroup barrier since): the row's scale and inverse for this lane, the inverse
// also in this wave's table (emit_planes reads it back as the 16-byte pieces the transposed tile wants).
__device__ __forceinline__ float row_scales(float* __restrict__ sc, int wave, int lane, float& s, float& inv)
{
    const int r = lane & 31;
    float m = 0.0f;
#pragma unroll
    for (int w = 0; w < PL_WAVES; ++w) m = fmaxf(m, fmaxf(sc[w * 64 + r], sc[w * 64 + 32 + r]));
    scale_of(m, s, inv);
    float* const tab = sc + PL_WAVES * 64 + wave * 32;
    tab[r] = inv;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");        // (this wave's own LDS accesses complete in order)
    __builtin_amdgcn_wave_barrier();
    return m;
}
// An image's maxima for the weight-gradient launch: PL_AMAX floats per 32-row block -- here the block's one maximum
// (every feature of the image) and zeros.  m: this lane's row maximum; floor: 1 for the images with a column of ones.
__device__ __forceinline__ void store_amax_rows(float* __restrict__ amax_rb, float m, float floor_, int lane)
{
    const float M = fmaxf(wave_max(m), floor_);
    if (lane < PL_AMAX) amax_rb[lane] = lane == 0 ? M : 0.0f;
}

// B operand of the transposing product: element j of lane (c, h) in step s is 1 where the (permuted)
// k index 16 s + 8 (j >> 2) + 4 h + (j & 3) equals the column c
template <int NP>
__device__ __forceinline__ void make_identity(bf16x8 idf[2], int lane)
{
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if constexpr (NP == 2) {
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (_Float16)((16 * s + 8 * (j >> 2) + 4 * h + (j & 3)) == c ? 1.0f : 0.0f);
            idf[s] = __builtin_bit_cast(bf16x8, o);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) idf[s][j] = (__bf16)((16 * s + 8 * (j >> 2) + 4 * h + (j & 3)) == c ? 1.0f : 0.0f);
        }
    }
}

// One 32-feature block of one workgroup (32 batch rows), held as the two k-steps' fragments f[0], f[1]
// (lane = batch row), written transposed: dst -> timage[block][this workgroup's first row step].
// ones_c >= 0: that column of the block is the appended column of ones (rows below rows_left only).
// inv_tab (fp16 x 2): the 32 rows' inverse scales (LDS) -- the image holds the values themselves, hi + lo unscaled.
template <int NP>
__device__ __forceinline__ void emit_planes(char* dst, const Frag<NP> f[2], const bf16x8 idf[2], int lane, int ones_c, int rows_left,
                                            const float* __restrict__ inv_tab = nullptr)
{
    const int c = lane & 31, h = lane >> 5;
    f32x16 t;                  // t[q] of lane (c, h) = feature c of batch row (q & 3) + 8 (q >> 2) + 4 h
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
        f32x16 u;
#pragma unroll
        for (int q = 0; q < 16; ++q) u[q] = 0.0f;
        u = pl_mfma<NP>(f[0].p[pl], idf[0], u);
        u = pl_mfma<NP>(f[1].p[pl], idf[1], u);
        if (pl == 0) t = u;
        else t += u;           // hi + mid, then + lo: exact, the terms do not overlap
    }
    if constexpr (NP == 2) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 iv = *reinterpret_cast<const f32x4*>(inv_tab + 8 * g + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) t[4 * g + e] *= iv[e];
        }
    }
    if (c == ones_c) {
#pragma unroll
        for (int q = 0; q < 16; ++q) t[q] = (q & 3) + 8 * (q >> 2) + 4 * h < rows_left ? 1.0f : 0.0f;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if constexpr (NP >= 2) {       // tile = [elements 0..3 | 4..7][lane][4 floats]: two lane-linear 1 KB halves
            char* o = dst + s * 2048 + lane * 16;
            *reinterpret_cast<f32x4*>(o) = f32x4{t[8 * s], t[8 * s + 1], t[8 * s + 2], t[8 * s + 3]};
            *reinterpret_cast<f32x4*>(o + 1024) = f32x4{t[8 * s + 4], t[8 * s + 5], t[8 * s + 6], t[8 * s + 7]};
        } else {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (__bf16)t[8 * s + j];
            *reinterpret_cast<bf16x8*>(dst + s * 1024 + lane * 16) = o;
        }
    }
}

// The reverse gather for one lane of a chain accumulator (batch row r of the workgroup, registers =
// features (q & 3) + 8 (q >> 2) + 4 h of a block): byte offset of its row inside a (block, 2 row steps)
// tile pair, to which feature c adds c * (8 elements).
template <int NP>
__device__ __forceinline__ int tgather_row_offset(int r)
{
    // r = 16 s' + 8 (j >> 2) + 4 h' + (j & 3)
    const int sp = r >> 4, jh = (r >> 3) & 1, jl = r & 3, hp = (r >> 2) & 1;
    if constexpr (NP >= 2) return sp * 2048 + jh * 1024 + hp * 32 * 16 + jl * 4;      // + c * 16
    else return sp * 1024 + hp * 32 * 16 + (4 * jh + jl) * 2;                         // + c * 16
}
template <int NP>
__device__ __forceinline__ float tgather(const char* tile_pair, int row_off, int c)
{
    if constexpr (NP >= 2) return *reinterpret_cast<const float*>(tile_pair + row_off + c * 16);
    else return (float)*reinterpret_cast<const __bf16*>(tile_pair + row_off + c * 16);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// ---------------------------------------------------------------------------------------------
// weights -> operand fragments
// ---------------------------------------------------------------------------------------------
struct PackJob {
    const float* W;        // [N][K] row-major (nn.Linear.weight)
    int N, K;
    int transposed;        // 0: operand rows are output features, sum over k (forward)
                           // 1: operand rows are input features, sum over n (backward, W^T)
    int nblk, nsteps;
    int tile0;             // first tile (block, step) of this job in the launch
    int blk0;              // first 32-row block of this job in the launch (pack_planes_scaled_kernel: a workgroup per block)
    int64_t dst;           // byte offset of the image
};
struct PackTable {
    int n_jobs, n_tiles, n_blocks;
    char* base;
    PackJob job[2 * ABN_MAX_LAYERS];
};

template <int NP>
__global__ __launch_bounds__(256) void pack_planes_kernel(PackTable t)
{
    const int tile = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (tile >= t.n_tiles) return;
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    int jn = 0;
    while (jn + 1 < t.n_jobs && tile >= t.job[jn + 1].tile0) ++jn;
    const PackJob& J = t.job[jn];
    const int local = tile - J.tile0;
    const int nb = local / J.nsteps, s = local % J.nsteps;
    const int a = 32 * nb + r;                       // operand row
    const int rows_a = J.transposed ? J.K : J.N;     // operand rows in the matrix
    const int len_c = J.transposed ? J.N : J.K;      // length of the sum
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
    if (a < rows_a) {
        const int c0 = 16 * s + 4 * h, c1 = c0 + 8;
        if (!J.transposed) {                         // K % 4 == 0: four k in or out together
            if (c0 < len_c) v0 = *reinterpret_cast<const f32x4*>(J.W + (int64_t)a * J.K + c0);
            if (c1 < len_c) v1 = *reinterpret_cast<const f32x4*>(J.W + (int64_t)a * J.K + c1);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (c0 + e < len_c) v0[e] = J.W[(int64_t)(c0 + e) * J.K + a];
                if (c1 + e < len_c) v1[e] = J.W[(int64_t)(c1 + e) * J.K + a];
            }
        }
    }
    write_frag<NP>(t.base + J.dst + ((int64_t)local * NP) * 1024 + lane * 16, v0, v1);
}

// fp16 x 2: a workgroup per 32-row block of an image.  Its waves take the block's steps (at most four each: K <=
// 512) into registers, agree on the block's largest magnitude through LDS, and write the fragments scaled by the
// block's power of two; the inverse goes behind the image's fragments (one float per block), where the chains'
// epilogues pick it up.  One pass over the weights, no launch of its own for the maxima.
__global__ __launch_bounds__(PL_NT) void pack_planes_scaled_kernel(PackTable t)
{
    __shared__ float wmax[PL_WAVES];
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x;
    int jn = 0;
    while (jn + 1 < t.n_jobs && b >= t.job[jn + 1].blk0) ++jn;
    const PackJob& J = t.job[jn];
    const int nb = b - J.blk0;
    const int rows_a = J.transposed ? J.K : J.N;     // operand rows in the matrix
    const int len_c = J.transposed ? J.N : J.K;      // length of the sum (a multiple of 4)
    const int a = 32 * nb + r;                       // operand row
    const int ac = a < rows_a ? a : rows_a - 1;
    constexpr int SPW = PL_MAXSTEPS / PL_WAVES;      // steps per wave
    // (every load from a clamped address, the padding selected away afterwards: conditional stores into v would
    // carry the whole array through each branch)
    f32x4 v[SPW][2];
    float m = 0.0f;
#pragma unroll
    for (int i = 0; i < SPW; ++i) {
        const int s = wave + PL_WAVES * i;
        const bool live = s < J.nsteps && a < rows_a;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int c = 16 * s + 4 * h + 8 * u;
            const int cc = c < len_c ? c : len_c - 4;
            f32x4 x;
            if (!J.transposed) {
                x = *reinterpret_cast<const f32x4*>(J.W + (int64_t)ac * J.K + cc);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = J.W[(int64_t)(cc + e) * J.K + ac];
            }
            const bool ok = live && c < len_c;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][u][e] = ok ? x[e] : 0.0f;
        }
        m = fmaxf(m, absmax8(v[i][0], v[i][1]));
    }
    m = wave_max(m);
    if (lane == 0) wmax[wave] = m;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < PL_WAVES; ++w) m = fmaxf(m, wmax[w]);
    float sc, inv;
    scale_of(m, sc, inv);
    char* const image = t.base + J.dst;
#pragma unroll
    for (int i = 0; i < SPW; ++i) {
        const int s = wave + PL_WAVES * i;
        if (s < J.nsteps) write_frag<2>(image + ((int64_t)(nb * J.nsteps + s) * 2) * 1024 + lane * 16, v[i][0], v[i][1], sc);
    }
    if (threadIdx.x == 0) reinterpret_cast<float*>(image + (int64_t)J.nblk * J.nsteps * 2048)[nb] = inv;
}
// the inverse scale of block blk of a packed image (fp16 x 2)
__device__ __forceinline__ float packed_inv(const char* __restrict__ image, int nblk, int nsteps, int blk)
{
    return reinterpret_cast<const float*>(image + (int64_t)nblk * nsteps * 2048)[blk < nblk ? blk : nblk - 1];
}

// ---------------------------------------------------------------------------------------------
// the chains' k-loop
// ---------------------------------------------------------------------------------------------
// The weight ring of a wave: PL_DEPTH steps of (up to) two blocks, in registers.  It is filled with raw buffer loads:
// the compiler counts their vmcnt itself, and -- unlike plain loads, which InstCombine sinks through the loop's phi
// right in front of their MFMAs -- they stay where the pipeline puts them, PL_DEPTH steps ahead.
// A chain keeps ONE ring through its layers (rs spans the whole packed image, offsets are relative to it): where the
// next layer gives the wave the same share (two blocks, all steps), the last PL_DEPTH refills of a layer fetch the
// next layer's first steps instead of repeats nobody reads -- they are in flight through the epilogue, OLDER than its
// image stores, so the next k-loop starts without a round trip (`filled`).
// (measured at C2, fp16 x 2: 0.1851-0.1866 ms / step with the hand-over against 0.1848-0.1850 without -- the k-loops
// are bound by what the 32 CUs of an XCD pull from its L2, a round trip less per layer changes nothing, and the ring
// held through the epilogue spills: off unless built with -DPL_HANDOVER_ON)
#ifdef PL_HANDOVER_ON
constexpr bool PL_HANDOVER = true;
#else
constexpr bool PL_HANDOVER = false;
#endif
#ifndef PL_THROTTLE_LAG
#define PL_THROTTLE_LAG 0
#endif
#ifndef PL_THROTTLE_NAP
#define PL_THROTTLE_NAP 2
#endif
template <int NP>
struct WeightRing {
    __amdgpu_buffer_rsrc_t rs;
    v4i wq[PL_DEPTH][2][NP];
    bool filled;
};
template <int NP>
__device__ __forceinline__ void ring_open(WeightRing<NP>& ring, const char* base, int64_t bytes)
{
    ring.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)bytes, 0x00020000);
    ring.filled = false;
}

// acc[j] += block j x img over my_steps steps (a multiple of PL_DEPTH).  wv[j]: byte offset (from the ring's base) of
// block j's first step for this lane; dnext[j] (wave-uniform): what to add to reach the block the wave takes in the
// next layer (chain_next), whose first PL_DEPTH steps the ring holds on return.
template <int NP, int BPW>
__device__ __forceinline__ void ring_kloop(f32x16* acc, WeightRing<NP>& ring, const int* wv, const int* dnext, bool chain_next,
                                           const char* __restrict__ img, int s_first, int my_steps, int lane)
{
    const char* ab = img + (int64_t)s_first * (NP * 1024) + lane * 16;
    const __amdgpu_buffer_rsrc_t rs = ring.rs;
    if (!ring.filled) {
#pragma unroll
        for (int i = 0; i < PL_DEPTH; ++i)
#pragma unroll
            for (int j = 0; j < BPW; ++j)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
                    // (fenced one by one: the in-order vmcnt the compiler derives for the loop is the worst of
                    // the loop's own order and this one)
                    ring.wq[i][j][pl] = __builtin_amdgcn_raw_buffer_load_b128(rs, wv[j], (i * NP + pl) * 1024, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
    }
    // activation fragments: read one step ahead, into alternating register sets
    bf16x8 af[2][NP];
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) af[0][pl] = *reinterpret_cast<const bf16x8*>(ab + pl * 1024);
#if PL_THROTTLE_LAG > 0
    // (experiment, round 6: the waves of a workgroup kept within PL_THROTTLE_LAG steps of the slowest -- a wave that is
    // further ahead naps: its weight requests leave the CU's in-order L1 to the others, who would otherwise finish
    // thousands of cycles behind it with only their own few loads in flight)
    __shared__ int pl_prog[PL_WAVES];
    const int my_wave = threadIdx.x >> 6;
#endif
    for (int s0 = 0; s0 < my_steps; s0 += PL_DEPTH) {
#if PL_THROTTLE_LAG > 0
        {
            if (lane == 0) pl_prog[my_wave] = s0;
            int mn = 1 << 20;
#pragma unroll
            for (int w = 0; w < PL_WAVES; ++w) {
                const unsigned v = (unsigned)pl_prog[w];
                mn = v < (unsigned)mn ? (int)v : mn;
            }
            if (s0 - mn > PL_THROTTLE_LAG) __builtin_amdgcn_s_sleep(PL_THROTTLE_NAP);
        }
#endif
#pragma unroll
        for (int i = 0; i < PL_DEPTH; ++i) {
            const int s = s0 + i;
#if defined(PL_EXP_PRIO) && PL_EXP_PRIO == 3
            // (experiment: the two waves of a SIMD take turns at the higher issue priority, step by step)
            if (((threadIdx.x >> 8) ^ i) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
            // (the three regions are fenced: left alone, the machine scheduler issues a refill before
            // the slot's last MFMA -- a second register set and copies that wait for the loads at the
            // loop's end -- or gathers all refills there)
            const int s1 = s + 1 < my_steps ? s + 1 : s;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) af[(i + 1) & 1][pl] = *reinterpret_cast<const bf16x8*>(ab + (s1 * NP + pl) * 1024);
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8* a = af[i & 1];
            // smallest terms first (gemm_f32.h); the blocks alternate so that consecutive MFMAs are independent
#pragma unroll
            for (int t = 0; t < Products<NP>::N; ++t)
#pragma unroll
                for (int j = 0; j < BPW; ++j)
                    acc[j] = pl_mfma<NP>(__builtin_bit_cast(bf16x8, ring.wq[i][j][Products<NP>::A[t]]), a[Products<NP>::B[t]], acc[j]);
            // (pure MFMA nodes float across a sched_barrier at instruction selection: the empty asm
            // ties the accumulators, and with them every MFMA of the step, in front of the refills)
#pragma unroll
            for (int j = 0; j < BPW; ++j) asm volatile("" : "+v"(acc[j]) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // refill the slot for step s + PL_DEPTH; past this layer's last step: the next layer's step i (chain_next),
            // else a repeat nobody reads
            const bool past = s + PL_DEPTH >= my_steps;
#pragma unroll
            for (int j = 0; j < BPW; ++j) {
                const int so = past ? (chain_next ? dnext[j] + i * (NP * 1024) : (my_steps - 1) * (NP * 1024)) : (s + PL_DEPTH) * (NP * 1024);
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) ring.wq[i][j][pl] = __builtin_amdgcn_raw_buffer_load_b128(rs, wv[j], so + pl * 1024, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#if PL_THROTTLE_LAG > 0
    if (lane == 0) pl_prog[my_wave] = 1 << 20;          // (done: nobody waits for this wave)
#endif
    ring.filled = chain_next;
}

// one layer's k-loop on a ring of its own (the launches that hold one layer: BatchNorm's data gradient)
template <int NP, int BPW>
__device__ __forceinline__ void planes_kloop(f32x16* acc, const char* __restrict__ image, int nblk, int nsteps,
                                             const char* __restrict__ img, int blk0, int s_first, int my_steps, int lane)
{
    WeightRing<NP> ring;
    ring_open(ring, image, (int64_t)nblk * nsteps * (NP * 1024));
    int wv[BPW], dnext[BPW];
#pragma unroll
    for (int j = 0; j < BPW; ++j) {
        const int blk = blk0 + j < nblk ? blk0 + j : nblk - 1;        // an odd block count leaves the last wave half idle
        wv[j] = (blk * nsteps + s_first) * (NP * 1024) + lane * 16;
        dnext[j] = 0;
    }
    ring_kloop<NP, BPW>(acc, ring, wv, dnext, false, img, s_first, my_steps, lane);
}

// which blocks and steps a wave sums in a layer with `nblk` output blocks (KS = 2, at most 4 blocks:
// waves 4..7 sum the second half of the steps for the blocks of waves 0..3 and hand their
// accumulators over through LDS)
template <int BPW, int KS>
struct WaveShare {
    int khalf, blk0, s_first, my_steps;
    bool active;
    __device__ __forceinline__ WaveShare(int wave, int nblk, int nsteps)
    {
        khalf = KS == 2 ? wave >> 2 : 0;
        blk0 = KS == 2 ? (wave & 3) : wave * BPW;
        s_first = KS == 2 ? khalf * (nsteps / 2) : 0;
        my_steps = KS == 2 ? nsteps / 2 : nsteps;          // a multiple of PL_DEPTH (KS = 2 only if nsteps % (2 PL_DEPTH) == 0)
        active = blk0 < nblk;
    }
};

// f(std::integral_constant<int, act>) for the runtime act: the activation's switch stays out of the
// unrolled loops (inside them it was most of the kernel's code, run once, from a cold instruction cache)
template <class F>
__device__ __forceinline__ void with_act(int act, F&& f)
{
    switch (act) {
        case ACT_SIGMOID: f(std::integral_constant<int, ACT_SIGMOID>{}); break;
        case ACT_RELU: f(std::integral_constant<int, ACT_RELU>{}); break;
        case ACT_TANH: f(std::integral_constant<int, ACT_TANH>{}); break;
        default: f(std::integral_constant<int, ACT_NONE>{}); break;
    }
}

// ---------------------------------------------------------------------------------------------
// dropout drawn in the kernels: multiplier of (layer, batch row, feature) = a hash of them and of the
// call's seed -- 0 with probability p (16 bits), else 1 / (1 - p).  The forward epilogue and the
// backward's act' step evaluate the same function: no mask tensor is drawn, stored or read (with the
// reference's default p = 0.1 those were 40 us of a 220 us step: 13 M bernoulli draws, 52 MB
// written, scaled, read twice).
// ---------------------------------------------------------------------------------------------
struct DropGen {
    uint32_t key, thr;
    float scale;
    bool on;
};
__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ DropGen make_drop(const unsigned long long* seedp, float p, int layer)
{
    DropGen g = {0u, 0u, 1.0f, false};
    if (!seedp) return g;
    const unsigned long long seed = *seedp;
    g.key = hash32((uint32_t)seed ^ (0x9e3779b9u * (uint32_t)(layer + 1))) ^ (uint32_t)(seed >> 32);
    const float t = p * 65536.0f + 0.5f;
    g.thr = t >= 65536.0f ? 65536u : (uint32_t)t;
    g.scale = p < 1.0f ? 1.0f / (1.0f - p) : 0.0f;
    g.on = true;
    return g;
}
// the four multipliers of features n .. n + 3 (n % 4 == 0, n < 512) of batch row gr (< 2^20)
__device__ __forceinline__ f32x4 drop4(const DropGen& g, int gr, int n)
{
    const uint32_t c = ((uint32_t)gr * 128u + (uint32_t)(n >> 2)) * 2u;
    const uint32_t h0 = hash32(c ^ g.key), h1 = hash32((c + 1u) ^ g.key);
    f32x4 m;
    m[0] = (h0 & 0xffffu) >= g.thr ? g.scale : 0.0f;
    m[1] = (h0 >> 16) >= g.thr ? g.scale : 0.0f;
    m[2] = (h1 & 0xffffu) >= g.thr ? g.scale : 0.0f;
    m[3] = (h1 >> 16) >= g.thr ? g.scale : 0.0f;
    return m;
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
struct PlanesFwdP {
    int n_layers;
    int rows;                      // total rows (both towers)
    int rows_call;                 // rows per forward_once call (x1 | x2 split)
    int dims[ABN_MAX_LAYERS + 1];
    int act[ABN_MAX_LAYERS];
    const float* x1;
    const float* x2;               // may be null: all rows in x1
    float* x_copy;                 // [rows, dims[0]] concatenated copy for the backward (may be null)
    const char* wp[ABN_MAX_LAYERS];   // packed forward image of layer l
    const char* wbase;                // the packed images' common buffer (wp[l] - wbase fits an int) and its size
    int64_t wbytes;
    const float* b[ABN_MAX_LAYERS];
    const float* mask[ABN_MAX_LAYERS];
    float* out[ABN_MAX_LAYERS];    // [rows, dims[l+1]] post-activation outputs, row-major: the last layer's (the others' may be null)
    // the layer inputs as transposed images with a column of ones appended (what the backward reads, and
    // the only copy of the hidden activations): tp[0] the inputs (dims[0] + 1 features), tp[l + 1] the
    // outputs of layer l < n_layers - 1
    char* tp[ABN_MAX_LAYERS];
    int64_t tp_steps;              // row steps of those images (pl_row_steps(rows))
    float* amax[ABN_MAX_LAYERS];   // fp16 x 2: tp[l]'s maxima, PL_AMAX floats per 32-row block (the weight-gradient launch's scales)
    const unsigned long long* drop_seed;   // in-kernel dropout for the layers without a mask tensor (null: off)
    float drop_p;
    // BatchNorm with running statistics (the BN instantiation only: inference), after the bias and before the
    // activation, the arithmetic of bn_apply_kernel's eval branch (tower.hip): ((z - rm) / sqrt(rv + eps)) gamma + beta
    const float* bn_rm[ABN_MAX_LAYERS];
    const float* bn_rv[ABN_MAX_LAYERS];
    const float* bn_w[ABN_MAX_LAYERS];
    const float* bn_b[ABN_MAX_LAYERS];
    float bn_eps;
    // BatchNorm in training (bn_fwd_layer_kernel: one launch per layer, out[l] = the pre-normalisation z):
    // per-workgroup column statistics of z, [workgroup][3][PL_MAXW] floats -- sum (z - c), sum (z - c)^2, c
    float* bn_part;
#ifdef ABN_STAMPS
    unsigned long long* stamps;
#endif
};

// One layer for one workgroup.  img holds the input fragments of all pl_steps(K) steps (zero in
// the padding); on return it holds this layer's output the same way, for pl_steps(N) steps.
// MODE: what the forward is for.  The inference instantiations compile the dropout and the transposed
// images out (their loop-invariant address and mask registers are what fills the register file).
// PL_BN_TRAIN: one layer of a BatchNorm tower in training (the kernel ends after it: the batch statistics
// span every workgroup's rows): the host asks for no activation and no transposed image; nothing is left in
// img; the column statistics of the workgroup's 32 rows go to bn_part.
enum { PL_TRAIN = 0, PL_INFER = 1, PL_INFER_BN = 2, PL_BN_TRAIN = 3 };

// sum over the 32 lanes of a half wave, valid in its lanes 16..31 (DPP: two quad permutes, two row rotations,
// row_bcast15 into the odd rows)
__device__ __forceinline__ float half_wave_sum(float v)
{
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));     // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));     // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, false));    // row_ror:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false));    // row_ror:8
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));    // row_bcast15 -> rows 1, 3
    return v;
}

// The same sums (same steps, same order: the same bits) of FOUR values at once, each add carrying its DPP operand itself
// (v_add_f32_dpp).  What the compiler makes of update_dpp + add when it sees several at once is a zeroed temporary, a
// v_mov_b32_dpp and half a packed add per step and value -- 2.5 instructions where one does; the column sums of a BatchNorm
// layer are 64 such sums per lane.  Four independent chains keep every DPP read three instructions behind the write of its
// register (the hardware asks for two wait states; nothing inserts them inside an asm statement); s_nop 1 covers the
// values' producers.
__device__ __forceinline__ void half_wave_sum4(f32x4& v)
{
    float a = v[0], b = v[1], c = v[2], d = v[3];
#define ABN_DPP_STEP(ctrl)                                      \
    "v_add_f32_dpp %0, %0, %0 " ctrl " bank_mask:0xf\n\t"     \
    "v_add_f32_dpp %1, %1, %1 " ctrl " bank_mask:0xf\n\t"     \
    "v_add_f32_dpp %2, %2, %2 " ctrl " bank_mask:0xf\n\t"     \
    "v_add_f32_dpp %3, %3, %3 " ctrl " bank_mask:0xf\n\t"
    asm volatile("s_nop 1\n\t"
                 ABN_DPP_STEP("quad_perm:[1,0,3,2] row_mask:0xf")
                 ABN_DPP_STEP("quad_perm:[2,3,0,1] row_mask:0xf")
                 ABN_DPP_STEP("row_ror:4 row_mask:0xf")
                 ABN_DPP_STEP("row_ror:8 row_mask:0xf")
                 ABN_DPP_STEP("row_bcast:15 row_mask:0xa")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef ABN_DPP_STEP
    v = f32x4{a, b, c, d};
}

// ---------------------------------------------------------------------------------------------
// Staggered layers.  A chain's wide layers (more than eight output blocks: two per wave) are bound by the operand
// stream -- a CU pulls the packed layer through its L1 at ~45 B/clk whether four or eight waves ask for it
// (tools/planes_stamps.py with PL_EXP_HALFWAVES) -- and between two k-loops ~18 k cycles of epilogue, image writes and
// barriers pass with no load in flight.  Where two such layers follow each other the workgroup therefore runs as TWO
// GROUPS of four waves (one per SIMD each: waves 0-3 own blocks 0-7, waves 4-7 the rest) that drift half a layer apart:
// a group leaves its blocks in the OTHER operand image (two images: fp16 x 2 and bf16 only), publishes its rows' scales
// and a counter, and goes on; the next layer's k-loop walks the lower group's sixteen steps first, then the upper
// group's, waiting for each half's counter and rescaling the accumulators in between (each half has its own power of
// two per row: exact).  One group's epilogue then runs under the other group's stream.  No workgroup barrier inside a
// run; the last layer of a run ends with the ordinary epilogue (uniform scales, barriers).
// ---------------------------------------------------------------------------------------------
// MEASURED (C2, fp16 x 2, same box, tools/step_ab.py): forward chain staggered 0.1744-0.1757 ms / step against 0.1762-0.1772
// in phase; with the upper group held back until the lower one is through its first half (the lag that makes one group's
// epilogue fall under the other's stream) 0.1768-0.1779: a group streaming alone gets 45 B/clk through the CU's L1, both
// together 58 -- what the overlap hides, the lone stream loses.  Off unless built with -DPL_STAGGER_ON (the forward chain
// only; every planes / timed-path test passes with it on).
#ifdef PL_STAGGER_ON
constexpr bool PL_STAGGER = true;
#else
constexpr bool PL_STAGGER = false;
#endif
constexpr int PL_HS_BYTES = 2 * 2 * 2 * 32 * 4;                       // [layer parity][half][scale | inverse][row]
constexpr int PL_CNT_BYTES = (ABN_MAX_LAYERS + 1) * 2 * 2 * 4;        // [layer][group][maxima parked | blocks in the image]
// LDS of a kernel whose layers may stagger: the second image, the halves' scales and the counters behind everything else
static inline size_t pl_lds_bytes_stag(int np) { return pl_lds_bytes(np) + (PL_STAGGER && np <= 2 ? (size_t)PL_MAXSTEPS * np * 1024 + PL_HS_BYTES + PL_CNT_BYTES : 0); }
struct Stag {
    char* img0;            // the two operand images (selected, not indexed: an indexed member puts the struct in scratch)
    char* img1;
    int cur;               // which of them holds the next layer's input
    __device__ __forceinline__ char* in() const { return cur ? img1 : img0; }
    __device__ __forceinline__ char* out() const { return cur ? img0 : img1; }
    float* hs;
    int* cnt;
    bool in_halves;        // img[cur] was left by a staggered epilogue: per-half scales and counters
};
__device__ __forceinline__ void stag_open(Stag& st, char* smem, int np, int tid)
{
    st.img0 = smem;
    st.img1 = smem + pl_lds_bytes(np);
    st.hs = reinterpret_cast<float*>(st.img1 + PL_MAXSTEPS * np * 1024);
    st.cnt = reinterpret_cast<int*>(reinterpret_cast<char*>(st.hs) + PL_HS_BYTES);
    st.cur = 0;
    st.in_halves = false;
    if (tid < PL_CNT_BYTES / 4) st.cnt[tid] = 0;       // (a workgroup barrier follows before the first layer)
}
// one wave's arrival at a counter / waiting for `need` arrivals (the waves of a workgroup are co-resident: a spin is safe)
__device__ __forceinline__ void stag_arrive(int* c, int lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");            // (this wave's LDS writes first)
    if (lane == 0) __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void stag_wait(int* c, int need)
{
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < need) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// waves of group g that own a block of a layer with nblk output blocks (two blocks per wave)
__device__ __forceinline__ int stag_group_waves(int nblk, int g)
{
    const int w = (nblk + 1) / 2 - 4 * g;
    return w < 0 ? 0 : (w > 4 ? 4 : w);
}

template <int NP, int BPW, int KS, int MODE>
__device__ __forceinline__ void planes_layer(const PlanesFwdP& p, int l, char* __restrict__ img,
                                             float* __restrict__ part, const bf16x8* idf, int wave, int lane, int row0,
                                             float& ainv, WeightRing<NP>& ring, int row_end, Stag& st, bool stag_on, bool out_halves);
// (a layer outside any staggered run)
template <int NP, int BPW, int KS, int MODE = PL_TRAIN>
__device__ __forceinline__ void planes_layer(const PlanesFwdP& p, int l, char* __restrict__ img,
                                             float* __restrict__ part, const bf16x8* idf, int wave, int lane, int row0,
                                             float& ainv, WeightRing<NP>& ring, int row_end = -1)
{
    Stag none = {};
    planes_layer<NP, BPW, KS, MODE>(p, l, img, part, idf, wave, lane, row0, ainv, ring, row_end, none, false, false);
}

// ainv (fp16 x 2): the inverse scale of this lane's row in img -- in: the layer's input, out: its output.
// st / out_halves: staggered layers (two blocks per wave only) -- img is st.in(); out_halves: this layer's
// consumer is staggered too (the output goes to the other image with per-half scales, no barrier).
template <int NP, int BPW, int KS, int MODE>
__device__ __forceinline__ void planes_layer(const PlanesFwdP& p, int l, char* __restrict__ img,
                                             float* __restrict__ part, const bf16x8* idf, int wave, int lane, int row0,
                                             float& ainv, WeightRing<NP>& ring, int row_end, Stag& st, bool stag_on, bool out_halves)
{
    constexpr bool BN = MODE == PL_INFER_BN, INFER = MODE == PL_INFER || MODE == PL_INFER_BN, BNT = MODE == PL_BN_TRAIN;
    constexpr bool STAGC = PL_STAGGER && BPW == 2 && KS == 1 && NP <= 2 && (MODE == PL_TRAIN || MODE == PL_INFER);
    const int rows_lim = row_end >= 0 ? row_end : p.rows;      // (BatchNorm training: the end of the workgroup's forward_once call)
    const int K = p.dims[l], N = p.dims[l + 1];
    const int nsteps = pl_steps(K), nblk = (N + 31) / 32;
    const int r = lane & 31, h = lane >> 5;
    const WaveShare<BPW, KS> ws(wave, nblk, nsteps);
    const int blk0 = ws.blk0;
    PSTAMPF(2 + 5 * l);

    f32x16 acc[BPW];
#pragma unroll
    for (int j = 0; j < BPW; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][q] = 0.0f;

    // All loads of the epilogue are issued as one batch from clamped addresses (a load under a
    // per-group `if` waits for its own round trip: eight of them in a row were 8.5 k cycles per layer).
    // The bias: ONE value per lane (feature 32 blk0 + lane of the wave's up to 64), requested before the
    // k-loop, parked in a per-wave LDS slot after it and read back as the 16-byte pieces the
    // accumulator layout wants (32 registers held across the k-loop spilled the bf16 x 3 kernel).
    float bias_lane = 0.0f, bn_lane[4] = {0.0f, 1.0f, 1.0f, 0.0f};      // running mean, running variance, gamma, beta
    float cinv[BPW];                                   // fp16 x 2: what turns an accumulator into the product
#pragma unroll
    for (int j = 0; j < BPW; ++j) cinv[j] = NP == 2 && ws.active ? packed_inv(p.wp[l], nblk, nsteps, blk0 + j) : 1.0f;
    {
        const float* __restrict__ bias = p.b[l];
        const int n = 32 * blk0 + lane;
        if (bias && ws.active && lane < 32 * BPW) bias_lane = bias[n < N ? n : N - 1];
        if (BN && ws.active && lane < 32 * BPW) {
            const int nc = n < N ? n : N - 1;
            bn_lane[0] = p.bn_rm[l][nc]; bn_lane[1] = p.bn_rv[l][nc]; bn_lane[2] = p.bn_w[l][nc]; bn_lane[3] = p.bn_b[l][nc];
        }
    }
    {
        // the ring is handed on where the next layer gives every wave the same share (two blocks, all steps)
        bool chain_next = false;
        int nblk2 = 1, nsteps2 = 0, ioff2 = 0;
        if constexpr (PL_HANDOVER && NP <= 2 && BPW == 2 && KS == 1 && !BNT) {
            if (l + 1 < p.n_layers) {
                nblk2 = (p.dims[l + 2] + 31) / 32;
                nsteps2 = pl_steps(N);
                ioff2 = (int)(p.wp[l + 1] - p.wbase);
                chain_next = nblk2 > PL_WAVES;
            }
        }
        if (!ws.active) ring.filled = false;
        const int ioff = (int)(p.wp[l] - p.wbase);
        int wv[BPW], dnext[BPW];
#pragma unroll
        for (int j = 0; j < BPW; ++j) {
            const int blk = blk0 + j < nblk ? blk0 + j : nblk - 1;        // an odd block count leaves the last wave half idle
            const int o = ioff + (blk * nsteps + ws.s_first) * (NP * 1024);
            const int blk2 = blk0 + j < nblk2 ? blk0 + j : nblk2 - 1;
            wv[j] = o + lane * 16;
            dnext[j] = ioff2 + blk2 * nsteps2 * (NP * 1024) - o;
        }
        float cin = ainv;                          // the inverse scale the sum ends up in
        bool done = false;
        if constexpr (STAGC) {
            if (stag_on && st.in_halves) {
                // the input's two halves: the lower group's steps, then the upper group's, each behind its counter
                done = true;
                if (ws.active) {
                    constexpr int SPLIT = 2 * PL_WAVES;                       // steps the lower group's eight blocks cover
                    const float* const hsin = st.hs + (l & 1) * 128;
                    int* const cnt = st.cnt + l * 4;
                    const int nb_in = (K + 31) / 32;
                    int dn[BPW], wv2[BPW];
#pragma unroll
                    for (int j = 0; j < BPW; ++j) { dn[j] = SPLIT * (NP * 1024); wv2[j] = wv[j] + SPLIT * (NP * 1024); }
                    stag_wait(&cnt[1], stag_group_waves(nb_in, 0));
                    const float inv0 = hsin[32 + r];
                    ring_kloop<NP, BPW>(acc, ring, wv, dn, true, img, 0, SPLIT, lane);
                    stag_wait(&cnt[3], stag_group_waves(nb_in, 1));
                    if constexpr (NP == 2) {
                        const float f = hsin[64 + r] * inv0;                  // (powers of two: exact)
#pragma unroll
                        for (int j = 0; j < BPW; ++j)
#pragma unroll
                            for (int q = 0; q < 16; ++q) acc[j][q] *= f;
                        cin = hsin[96 + r];
                    }
                    ring_kloop<NP, BPW>(acc, ring, wv2, dnext, chain_next, img, SPLIT, nsteps - SPLIT, lane);
                }
            }
        }
#ifdef PL_EXP_HALFWAVES
        if (!done && ws.active && (wave < 4 || BPW != 2)) ring_kloop<NP, BPW>(acc, ring, wv, dnext, chain_next, img, ws.s_first, ws.my_steps, lane);
#else
        if (!done && ws.active) ring_kloop<NP, BPW>(acc, ring, wv, dnext, chain_next, img, ws.s_first, ws.my_steps, lane);
#endif
#pragma unroll
        for (int j = 0; j < BPW; ++j) cinv[j] *= cin;
    }
    PSTAMPW(l);
    PSTAMPF(3 + 5 * l);

    // epilogue.  Register q of block j is output feature 32 (blk0 + j) + (q & 3) + 8 (q >> 2) + 4 h of
    // batch row r: registers 4g .. 4g+3 are four consecutive features (one 16-byte piece of the
    // row-major output), registers 8t .. 8t+7 the lane's operand of step 2 blk + t of the next layer.
    const float* __restrict__ mask = INFER ? nullptr : p.mask[l];
    const DropGen drop = make_drop(INFER || mask ? nullptr : p.drop_seed, p.drop_p, l);
    const bool masked = mask || drop.on;
    const int gr = row0 + r;
    const bool row_ok = gr < rows_lim;
    float* const sc = part + PL_PART_BYTES / 4;                // fp16 x 2: row maxima and inverse scales
    float* const bias_s = part + (PL_PART_BYTES + PL_SC_BYTES) / 4 + wave * 64;
    bias_s[lane] = bias_lane;
    if (BN) {
        bias_s[PL_BIAS_BYTES / 4 + lane] = bn_lane[0];
        bias_s[2 * (PL_BIAS_BYTES / 4) + lane] = 1.0f / sqrtf(bn_lane[1] + p.bn_eps);
        bias_s[3 * (PL_BIAS_BYTES / 4) + lane] = bn_lane[2];
        bias_s[4 * (PL_BIAS_BYTES / 4) + lane] = bn_lane[3];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");        // (this wave's own LDS accesses complete in order)
    __builtin_amdgcn_wave_barrier();
    f32x4 mv[BPW][4];
    if (mask && ws.active && ws.khalf == 0) {
        const float* mrow = mask + (int64_t)(row_ok ? gr : rows_lim - 1) * N;
#pragma unroll
        for (int j = 0; j < BPW; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = 32 * (blk0 + j) + 4 * h + 8 * g;
                mv[j][g] = *reinterpret_cast<const f32x4*>(mrow + (n < N ? n : N - 4));
            }
    }
    auto finish = [&]() {
        with_act(p.act[l], [&](auto tag) {
            constexpr int ACT = decltype(tag)::value;
#pragma unroll
            for (int j = 0; j < BPW; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bool live = 32 * (blk0 + j) + 4 * h + 8 * g < N;      // N % 4 == 0: four features in or out together
                    f32x4 m4 = {1.f, 1.f, 1.f, 1.f};
                    if (mask) m4 = mv[j][g];
                    else if (drop.on) m4 = drop4(drop, gr, 32 * (blk0 + j) + 4 * h + 8 * g);
                    int so = 32 * j + 4 * h + 8 * g;
                    // (the BN vectors of all eight groups, hoisted above the activation switch as common code,
                    // are 160 registers: an opaque offset keeps each group's reads where they are used)
                    if (BN) asm volatile("" : "+v"(so));
                    const float* const slot = bias_s + so;
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(slot);
                    f32x4 mu4 = {}, is4 = {}, ga4 = {}, be4 = {};
                    if (BN) {
                        mu4 = *reinterpret_cast<const f32x4*>(slot + PL_BIAS_BYTES / 4);
                        is4 = *reinterpret_cast<const f32x4*>(slot + 2 * (PL_BIAS_BYTES / 4));
                        ga4 = *reinterpret_cast<const f32x4*>(slot + 3 * (PL_BIAS_BYTES / 4));
                        be4 = *reinterpret_cast<const f32x4*>(slot + 4 * (PL_BIAS_BYTES / 4));
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = (NP == 2 ? acc[j][4 * g + e] * cinv[j] : acc[j][4 * g + e]) + b4[e];
                        if (BN) v = ((v - mu4[e]) * is4[e]) * ga4[e] + be4[e];
                        if (masked) v *= m4[e];
                        acc[j][4 * g + e] = live ? act_apply(v, ACT) : 0.0f;
                    }
                    if (BN) __builtin_amdgcn_sched_barrier(0);
                }
        });
    };
    // fp16 x 2: a layer whose output is multiplied again agrees on its rows' scales -- every lane parks the largest
    // magnitude it holds of its row in front of the barrier (K-split layers: behind it, and a second barrier)
    const bool rescale = NP == 2 && !BNT && l + 1 < p.n_layers;
    auto park_max = [&]() {
        float m = 0.0f;
        if (ws.active && ws.khalf == 0) {
#pragma unroll
            for (int j = 0; j < BPW; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) m = fmaxf(m, fabsf(acc[j][q]));
        }
        sc[wave * 64 + lane] = m;
    };
    if constexpr (STAGC) {
        if (stag_on && out_halves) {
            // staggered epilogue: this group's blocks go to the other image under scales of the group's own, no barrier
            const int g = wave >> 2;
            char* const img_out = st.out();
            int* const cnt_out = st.cnt + (l + 1) * 4 + 2 * g;
            float* const hs_out = st.hs + ((l + 1) & 1) * 128 + 64 * g;
            float* const tab = sc + PL_WAVES * 64 + wave * 32;
            char* const tp_s = !INFER ? p.tp[l + 1] : nullptr;
            if (ws.active) {
                finish();
                float osc_s = 1.0f;
                if constexpr (NP == 2) {
                    float m = 0.0f;
#pragma unroll
                    for (int j = 0; j < BPW; ++j)
#pragma unroll
                        for (int q = 0; q < 16; ++q) m = fmaxf(m, fabsf(acc[j][q]));
                    sc[wave * 64 + lane] = m;
                    const int gw = stag_group_waves(nblk, g);
                    stag_arrive(&cnt_out[0], lane);
                    stag_wait(&cnt_out[0], gw);
                    m = 0.0f;
                    for (int w = 4 * g; w < 4 * g + gw; ++w) m = fmaxf(m, fmaxf(sc[w * 64 + r], sc[w * 64 + 32 + r]));
                    float oinv;
                    scale_of(m, osc_s, oinv);
                    tab[r] = oinv;
                    if ((wave & 3) == 0) {
                        hs_out[r] = osc_s;
                        hs_out[32 + r] = oinv;
                        if (tp_s) {                    // the image's maxima: one slot per group (the column of ones: >= 1)
                            const float M = fmaxf(wave_max(m), 1.0f);
                            float* const am = p.amax[l + 1] + (int64_t)blockIdx.x * PL_AMAX;
                            if (g == 0) { if (lane < PL_AMAX && lane != 1) am[lane] = lane == 0 ? M : 0.0f; }
                            else if (lane == 1) am[1] = M;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                }
#pragma unroll
                for (int j = 0; j < BPW; ++j) {
                    const int blk = blk0 + j;
                    if (blk < nblk) {
                        Frag<NP> f[2];
#pragma unroll
                        for (int t2 = 0; t2 < 2; ++t2) {
                            const f32x4 v0 = {acc[j][8 * t2], acc[j][8 * t2 + 1], acc[j][8 * t2 + 2], acc[j][8 * t2 + 3]};
                            const f32x4 v1 = {acc[j][8 * t2 + 4], acc[j][8 * t2 + 5], acc[j][8 * t2 + 6], acc[j][8 * t2 + 7]};
                            f[t2] = make_frag<NP>(v0, v1, osc_s);
                            store_frag<NP>(img_out + (int64_t)(2 * blk + t2) * (NP * 1024) + lane * 16, f[t2]);
                        }
                        if (tp_s)
                            emit_planes<NP>(tp_s + ((int64_t)blk * p.tp_steps + 2 * blockIdx.x) * tile_bytes<NP>(), f, idf, lane,
                                            blk == N / 32 ? N % 32 : -1, rows_lim - row0, tab);
                    }
                }
                if (wave == PL_WAVES / 2) {            // the upper group's first wave: the next layer's padding steps
                    const bf16x8 z = {};
                    for (int s2 = 2 * nblk; s2 < pl_steps(N); ++s2)
#pragma unroll
                        for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<bf16x8*>(img_out + ((int64_t)s2 * NP + pl) * 1024 + lane * 16) = z;
                }
                stag_arrive(&cnt_out[1], lane);
            }
            if (tp_s && N % 32 == 0 && wave == PL_WAVES - 1) {       // the column of ones opens a block of its own
                Frag<NP> z[2] = {};
                if (!ws.active) {                  // (zeros times this wave's table: it must hold numbers)
                    tab[r] = 1.0f;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                }
                emit_planes<NP>(tp_s + ((int64_t)nblk * p.tp_steps + 2 * blockIdx.x) * tile_bytes<NP>(), z, idf, lane, 0, rows_lim - row0, tab);
            }
            st.cur ^= 1;
            st.in_halves = true;
            PSTAMPF(6 + 5 * l);
            return;
        }
    }
    if (KS == 1) {
        if (ws.active) finish();
        if (rescale) park_max();
    } else if (ws.active && ws.khalf == 1) {
#pragma unroll
        for (int q = 0; q < 16; ++q) part[((wave & 3) * 16 + q) * 64 + lane] = acc[0][q];
    }
    PSTAMPF(4 + 5 * l);
    __syncthreads();                               // every wave is done reading img
    PSTAMPF(5 + 5 * l);
    if constexpr (STAGC) { if (stag_on) st.in_halves = false; }
    if (KS == 2 && ws.active && ws.khalf == 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[0][q] += part[(wave * 16 + q) * 64 + lane];
        finish();
    }
    float osc = 1.0f;                              // this lane's row: the scale of the output fragments
    if (rescale) {
        if (KS == 2) { park_max(); __syncthreads(); }
        float oinv;
        const float m = row_scales(sc, wave, lane, osc, oinv);
        ainv = oinv;
        if (!INFER && p.tp[l + 1] && wave == 0) store_amax_rows(p.amax[l + 1] + (int64_t)blockIdx.x * PL_AMAX, m, 1.0f, lane);      // (a forward nothing is kept of: no images)
    }
    const float* const inv_tab = sc + PL_WAVES * 64 + wave * 32;
    float* __restrict__ out = p.out[l];
    char* const tp = !INFER && !BNT && l + 1 < p.n_layers ? p.tp[l + 1] : nullptr;
    if (BNT && ws.active && ws.khalf == 0) {
        // Column statistics of this workgroup's (up to) 32 rows (all of one forward_once call), shifted by the first
        // row's value so that the float32 sums are of the variance's size, not the mean's: the finishing
        // kernel (tower.hip) rebuilds sum z and sum z^2 in float64 and adds the workgroups in a fixed order.
        float* const pw = p.bn_part + (int64_t)blockIdx.x * (3 * PL_MAXW);
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
                    const float d = row_ok ? zv - c : 0.0f;           // (a last workgroup of a call may hold fewer rows)
                    sd[e] = d;
                    sq[e] = d * d;
                    cc[e] = c;
                }
                half_wave_sum4(sd);
                half_wave_sum4(sq);
                if (r == 16) {
                    const int n = 32 * (blk0 + j) + 8 * g + 4 * h;
                    *reinterpret_cast<f32x4*>(pw + n) = sd;
                    *reinterpret_cast<f32x4*>(pw + PL_MAXW + n) = sq;
                    *reinterpret_cast<f32x4*>(pw + 2 * PL_MAXW + n) = cc;
                }
            }
        }
    }
    if (ws.active && ws.khalf == 0) {
#pragma unroll
        for (int j = 0; j < BPW; ++j) {
            const int blk = blk0 + j;
            if (blk < nblk) {
                Frag<NP> f[2];
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    const f32x4 v0 = {acc[j][8 * t2], acc[j][8 * t2 + 1], acc[j][8 * t2 + 2], acc[j][8 * t2 + 3]};
                    const f32x4 v1 = {acc[j][8 * t2 + 4], acc[j][8 * t2 + 5], acc[j][8 * t2 + 6], acc[j][8 * t2 + 7]};
                    if (!BNT && (NP != 2 || rescale)) {
                        f[t2] = make_frag<NP>(v0, v1, osc);
                        store_frag<NP>(img + (int64_t)(2 * blk + t2) * (NP * 1024) + lane * 16, f[t2]);
                    }
                    const int n = 32 * blk + 16 * t2 + 4 * h;
                    if (out && row_ok && n < N) *reinterpret_cast<f32x4*>(out + (int64_t)gr * N + n) = v0;
                    if (out && row_ok && n + 8 < N) *reinterpret_cast<f32x4*>(out + (int64_t)gr * N + n + 8) = v1;
                }
                if (tp)
                    emit_planes<NP>(tp + ((int64_t)blk * p.tp_steps + 2 * blockIdx.x) * tile_bytes<NP>(), f, idf, lane,
                                    blk == N / 32 ? N % 32 : -1, rows_lim - row0, inv_tab);
            }
        }
    }
    if (tp && N % 32 == 0 && wave == PL_WAVES - 1) {       // the column of ones opens a block of its own
        Frag<NP> z[2] = {};
        emit_planes<NP>(tp + ((int64_t)nblk * p.tp_steps + 2 * blockIdx.x) * tile_bytes<NP>(), z, idf, lane, 0, rows_lim - row0, inv_tab);
    }
    PSTAMPF(6 + 5 * l);
    // steps of the next layer's padding that no block of this layer covers
    if (!BNT && l + 1 < p.n_layers) {
        const int next_steps = pl_steps(N);
        const bf16x8 z = {};
        for (int s = 2 * nblk + wave; s < next_steps; s += PL_WAVES)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<bf16x8*>(img + ((int64_t)s * NP + pl) * 1024 + lane * 16) = z;
    }
    __syncthreads();
}

// The input rows of a workgroup -> operand fragments in img (+ the concatenated copy, + the transposed image
// for the weight gradient).
template <int NP, bool INFER>
__device__ __forceinline__ void planes_input_stage(const PlanesFwdP& p, char* __restrict__ img, float* __restrict__ part,
                                                   const bf16x8* idf, int wave, int lane, int row0, float& ainv, int row_end = -1)
{
    const int rows_lim = row_end >= 0 ? row_end : p.rows;
    const int r = lane & 31, h = lane >> 5;
    const int D0 = p.dims[0];
    // input rows -> operand fragments (+ the concatenated copy for the backward): lane (r, h) of
    // step s holds x[row r][16 s + 4 h + 0..3] and x[row r][16 s + 8 + 4 h + 0..3]; a wave takes
    // whole 32-feature blocks (two steps), which it also writes transposed for the weight gradient
    const int steps0 = pl_steps(D0), blocks0 = steps0 / 2;     // PL_DEPTH is even
    const int gr = row0 + r;
    const float* src = nullptr;
    if (gr < rows_lim) src = (p.x2 && gr >= p.rows_call) ? p.x2 + (int64_t)(gr - p.rows_call) * D0 : p.x1 + (int64_t)gr * D0;
    float* const sc = part + PL_PART_BYTES / 4;
    const float* const inv_tab = sc + PL_WAVES * 64 + wave * 32;
    if constexpr (NP == 2) {
        // fp16 x 2: a wave's (up to two) blocks wait in registers until the workgroup has agreed on the rows' scales
        f32x4 v[2][2][2];                               // [block of this wave][step of the block][half]
        float m = 0.0f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int kb = wave + PL_WAVES * u;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const int c0 = 16 * (2 * kb + t2) + 4 * h, c1 = c0 + 8;
                v[u][t2][0] = v[u][t2][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (src && kb < blocks0 && c0 < D0) {
                    v[u][t2][0] = *reinterpret_cast<const f32x4*>(src + c0);
                    if (p.x_copy) *reinterpret_cast<f32x4*>(p.x_copy + (int64_t)gr * D0 + c0) = v[u][t2][0];
                }
                if (src && kb < blocks0 && c1 < D0) {
                    v[u][t2][1] = *reinterpret_cast<const f32x4*>(src + c1);
                    if (p.x_copy) *reinterpret_cast<f32x4*>(p.x_copy + (int64_t)gr * D0 + c1) = v[u][t2][1];
                }
                m = fmaxf(m, absmax8(v[u][t2][0], v[u][t2][1]));
            }
        }
        sc[wave * 64 + lane] = m;
        __syncthreads();
        float osc;
        m = row_scales(sc, wave, lane, osc, ainv);
        if (!INFER && p.tp[0] && wave == 0) store_amax_rows(p.amax[0] + (int64_t)blockIdx.x * PL_AMAX, m, 1.0f, lane);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int kb = wave + PL_WAVES * u;
            if (kb < blocks0) {
                Frag<NP> f[2];
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    f[t2] = make_frag<NP>(v[u][t2][0], v[u][t2][1], osc);
                    store_frag<NP>(img + (int64_t)(2 * kb + t2) * (NP * 1024) + lane * 16, f[t2]);
                }
                if (!INFER && p.tp[0] && kb < pl_blocks(D0 + 1))
                    emit_planes<NP>(p.tp[0] + ((int64_t)kb * p.tp_steps + 2 * blockIdx.x) * tile_bytes<NP>(), f, idf, lane,
                                    kb == D0 / 32 ? D0 % 32 : -1, rows_lim - row0, inv_tab);
            }
        }
    } else {
        for (int kb = wave; kb < blocks0; kb += PL_WAVES) {
            Frag<NP> f[2];
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const int s = 2 * kb + t2;
                f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
                const int c0 = 16 * s + 4 * h, c1 = c0 + 8;
                if (src && c0 < D0) {
                    v0 = *reinterpret_cast<const f32x4*>(src + c0);
                    if (p.x_copy) *reinterpret_cast<f32x4*>(p.x_copy + (int64_t)gr * D0 + c0) = v0;
                }
                if (src && c1 < D0) {
                    v1 = *reinterpret_cast<const f32x4*>(src + c1);
                    if (p.x_copy) *reinterpret_cast<f32x4*>(p.x_copy + (int64_t)gr * D0 + c1) = v1;
                }
                f[t2] = make_frag<NP>(v0, v1);
                store_frag<NP>(img + (int64_t)s * (NP * 1024) + lane * 16, f[t2]);
            }
            if (!INFER && p.tp[0] && kb < pl_blocks(D0 + 1))
                emit_planes<NP>(p.tp[0] + ((int64_t)kb * p.tp_steps + 2 * blockIdx.x) * tile_bytes<NP>(), f, idf, lane,
                                kb == D0 / 32 ? D0 % 32 : -1, rows_lim - row0);
        }
    }
    if (!INFER && p.tp[0] && pl_blocks(D0 + 1) > blocks0 && wave == PL_WAVES - 1) {     // D0 % 32 == 0 and no padding block to hold the ones
        Frag<NP> z[2] = {};
        emit_planes<NP>(p.tp[0] + ((int64_t)(D0 / 32) * p.tp_steps + 2 * blockIdx.x) * tile_bytes<NP>(), z, idf, lane, 0, rows_lim - row0, inv_tab);
    }
}

template <int NP, int MODE = PL_TRAIN>
__global__ __launch_bounds__(PL_NT) void tower_fwd_planes_kernel(PlanesFwdP p)
{
    constexpr bool INFER = MODE != PL_TRAIN;
    extern __shared__ __attribute__((aligned(16))) char pl_smem[];
    char* const img = pl_smem;
    float* const part = reinterpret_cast<float*>(pl_smem + PL_MAXSTEPS * NP * 1024);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row0 = blockIdx.x * PL_ROWS;
    bf16x8 idf[2];
    make_identity<NP>(idf, lane);
    PSTAMPF(0);
#ifdef PL_EXP_PRIO
    // (experiment: a static issue priority for one half of the workgroup's waves -- 1: waves 4-7, 2: waves 0-3)
    if ((PL_EXP_PRIO == 1) == (wave >= 4)) __builtin_amdgcn_s_setprio(1);
#endif

    float ainv = 1.0f;
    WeightRing<NP> ring;
    ring_open(ring, p.wbase, p.wbytes);
    constexpr bool STAG = PL_STAGGER && NP <= 2 && (MODE == PL_TRAIN || MODE == PL_INFER);
    Stag st;
    if constexpr (STAG) stag_open(st, pl_smem, NP, threadIdx.x);
    planes_input_stage<NP, INFER>(p, img, part, idf, wave, lane, row0, ainv);
    PSTAMPF(1);
    __syncthreads();

    for (int l = 0; l < p.n_layers; ++l) {
        const int nblk = (p.dims[l + 1] + 31) / 32;
        if constexpr (STAG) {
            char* const cur = st.in();
            // (a wide layer whose consumer is wide too leaves its output staggered)
            const bool out_halves = nblk > PL_WAVES && l + 1 < p.n_layers && (p.dims[l + 2] + 31) / 32 > PL_WAVES;
            if (nblk > PL_WAVES) planes_layer<NP, 2, 1, MODE>(p, l, cur, part, idf, wave, lane, row0, ainv, ring, -1, st, true, out_halves);
            else if (nblk > PL_WAVES / 2 || pl_steps(p.dims[l]) % (2 * PL_DEPTH) != 0) planes_layer<NP, 1, 1, MODE>(p, l, cur, part, idf, wave, lane, row0, ainv, ring);
            else planes_layer<NP, 1, 2, MODE>(p, l, cur, part, idf, wave, lane, row0, ainv, ring);
        } else {
            if (nblk > PL_WAVES) planes_layer<NP, 2, 1, MODE>(p, l, img, part, idf, wave, lane, row0, ainv, ring);
            else if (nblk > PL_WAVES / 2 || pl_steps(p.dims[l]) % (2 * PL_DEPTH) != 0) planes_layer<NP, 1, 1, MODE>(p, l, img, part, idf, wave, lane, row0, ainv, ring);
            else planes_layer<NP, 1, 2, MODE>(p, l, img, part, idf, wave, lane, row0, ainv, ring);
        }
    }
    PSTAMPF(2 + 5 * p.n_layers);
}

// ---------------------------------------------------------------------------------------------
// forward of a BatchNorm tower in training: one launch per layer.  The statistics of layer l - 1 span
// all rows of a forward_once call, so layer l starts in a new launch: its workgroups normalise their 32
// rows of z_{l-1} on the way into the operand image (and leave xhat and the activation for the backward),
// run the layer's product exactly as the single-launch forward does, and leave z_l + its per-workgroup
// column statistics (planes_layer, PL_BN_TRAIN).
// ---------------------------------------------------------------------------------------------
struct BnTrainP {
    int l;
    const float* mean;      // [n_calls][dims[l]] of layer l - 1 (l >= 1)
    const float* invstd;
    const float* z_prev;    // [rows][dims[l]]: z_{l-1} (it stays: the backward normalises it again)
    float* a_prev;          // [rows][dims[l]]: act(gamma xhat + beta) out, row-major (null: not wanted)
    // (p.tp[l]: the same activations as the transposed image [a_{l-1} | 1] the weight gradient reads)
    const int* n_valid;     // a padded batch (abn_tower_desc.n_valid): every call's rows past *n_valid do not exist for this launch
};

template <int NP>
__global__ __launch_bounds__(PL_NT) void bn_fwd_layer_kernel(PlanesFwdP p, BnTrainP q)
{
    extern __shared__ __attribute__((aligned(16))) char pl_smem[];
    char* const img = pl_smem;
    float* const part = reinterpret_cast<float*>(pl_smem + PL_MAXSTEPS * NP * 1024);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    // workgroups never straddle two forward_once calls: ceil(rows_call / 32) per call, the last one of a call
    // possibly short (its missing rows are zero rows of the transposed images, whose row axis is padded per call)
    const int wpc = (p.rows_call + PL_ROWS - 1) / PL_ROWS;
    const int call = blockIdx.x / wpc;
    const int row0 = call * p.rows_call + (blockIdx.x - call * wpc) * PL_ROWS;
    // (a padded batch: the call ends behind its real rows -- the rest are handled like the rows a short last workgroup lacks:
    // zero rows of the images, no statistics, nothing stored; a workgroup wholly behind the end leaves zero sums)
    const int row_end = call * p.rows_call + (q.n_valid ? min(max(*q.n_valid, 0), p.rows_call) : p.rows_call);
    const int l = q.l;
    bf16x8 idf[2];
    make_identity<NP>(idf, lane);
    float ainv = 1.0f;
    WeightRing<NP> ring;
    ring_open(ring, p.wbase, p.wbytes);

    if (l == 0) {
        planes_input_stage<NP, false>(p, img, part, idf, wave, lane, row0, ainv, row_end);
    } else {
        // the four per-feature vectors of this workgroup's call, parked in the (idle) K-split buffer
        const int K = p.dims[l];
        float* const mean_s = part, * const is_s = part + PL_MAXW, * const ga_s = part + 2 * PL_MAXW, * const be_s = part + 3 * PL_MAXW;
        for (int c = threadIdx.x; c < K; c += PL_NT) {
            mean_s[c] = q.mean[(int64_t)call * K + c];
            is_s[c] = q.invstd[(int64_t)call * K + c];
            ga_s[c] = p.bn_w[l - 1][c];
            be_s[c] = p.bn_b[l - 1][c];
        }
        __syncthreads();
        const int steps = pl_steps(K), blocks = steps / 2;
        const int gr = row0 + r;
        const bool row_ok = gr < row_end;
        const float* const zrow = q.z_prev + (int64_t)(row_ok ? gr : row0) * K;
        float* const arow = q.a_prev ? q.a_prev + (int64_t)gr * K : nullptr;
        char* const tp = p.tp[l];
        float* const sc = part + PL_PART_BYTES / 4;
        const float* const inv_tab = sc + PL_WAVES * 64 + wave * 32;
        with_act(p.act[l - 1], [&](auto tag) {
            constexpr int ACT = decltype(tag)::value;
            // one 16-feature step of the normalised, activated input for this lane's row
            auto in_step = [&](int s, f32x4* v) {
                v[0] = v[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int c = 16 * s + 4 * h + 8 * u;
                    if (c < K && row_ok) {
                        const f32x4 z4 = *reinterpret_cast<const f32x4*>(zrow + c);
                        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean_s + c), is = *reinterpret_cast<const f32x4*>(is_s + c);
                        const f32x4 ga = *reinterpret_cast<const f32x4*>(ga_s + c), be = *reinterpret_cast<const f32x4*>(be_s + c);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[u][e] = act_apply(((z4[e] - mu[e]) * is[e]) * ga[e] + be[e], ACT);
                        if (arow) *reinterpret_cast<f32x4*>(arow + c) = v[u];
                    }
                }
            };
            if constexpr (NP == 2) {
                // fp16 x 2: a wave's (up to two) blocks wait in registers until the workgroup has agreed on the rows' scales
                f32x4 v[2][2][2];
                float m = 0.0f;
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2) {
                        const int kb = wave + PL_WAVES * u;
                        v[u][t2][0] = v[u][t2][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (kb < blocks) in_step(2 * kb + t2, v[u][t2]);
                        m = fmaxf(m, absmax8(v[u][t2][0], v[u][t2][1]));
                    }
                sc[wave * 64 + lane] = m;
                __syncthreads();
                float osc;
                m = row_scales(sc, wave, lane, osc, ainv);
                if (tp && wave == 0) store_amax_rows(p.amax[l] + (int64_t)blockIdx.x * PL_AMAX, m, 1.0f, lane);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int kb = wave + PL_WAVES * u;
                    if (kb < blocks) {
                        Frag<NP> f[2];
#pragma unroll
                        for (int t2 = 0; t2 < 2; ++t2) {
                            f[t2] = make_frag<NP>(v[u][t2][0], v[u][t2][1], osc);
                            store_frag<NP>(img + (int64_t)(2 * kb + t2) * (NP * 1024) + lane * 16, f[t2]);
                        }
                        if (tp && kb < pl_blocks(K + 1))
                            emit_planes<NP>(tp + ((int64_t)kb * p.tp_steps + 2 * blockIdx.x) * tile_bytes<NP>(), f, idf, lane,
                                            kb == K / 32 ? K % 32 : -1, row_end - row0, inv_tab);
                    }
                }
            } else {
                for (int kb = wave; kb < blocks; kb += PL_WAVES) {
                    Frag<NP> f[2];
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2) {
                        f32x4 v[2];
                        in_step(2 * kb + t2, v);
                        f[t2] = make_frag<NP>(v[0], v[1]);
                        store_frag<NP>(img + (int64_t)(2 * kb + t2) * (NP * 1024) + lane * 16, f[t2]);
                    }
                    if (tp && kb < pl_blocks(K + 1))
                        emit_planes<NP>(tp + ((int64_t)kb * p.tp_steps + 2 * blockIdx.x) * tile_bytes<NP>(), f, idf, lane,
                                        kb == K / 32 ? K % 32 : -1, row_end - row0);
                }
            }
        });
        if (tp && pl_blocks(K + 1) > blocks && wave == PL_WAVES - 1) {      // K % 32 == 0 and no padding block to hold the ones
            Frag<NP> z[2] = {};
            emit_planes<NP>(tp + ((int64_t)(K / 32) * p.tp_steps + 2 * blockIdx.x) * tile_bytes<NP>(), z, idf, lane, 0, row_end - row0, inv_tab);
        }
    }
    __syncthreads();

    const int nblk = (p.dims[l + 1] + 31) / 32;
    if (nblk > PL_WAVES) planes_layer<NP, 2, 1, PL_BN_TRAIN>(p, l, img, part, idf, wave, lane, row0, ainv, ring, row_end);
    else if (nblk > PL_WAVES / 2 || pl_steps(p.dims[l]) % (2 * PL_DEPTH) != 0) planes_layer<NP, 1, 1, PL_BN_TRAIN>(p, l, img, part, idf, wave, lane, row0, ainv, ring, row_end);
    else planes_layer<NP, 1, 2, PL_BN_TRAIN>(p, l, img, part, idf, wave, lane, row0, ainv, ring, row_end);
}

// ---------------------------------------------------------------------------------------------
// backward, data gradient chain: dZ_{l-1} = (dZ_l W_l) act'(A_{l-1}) mask_{l-1}, for l = n_layers-1 .. 1
// ---------------------------------------------------------------------------------------------
struct PlanesBwdP {
    int n_layers;
    int rows;
    int dims[ABN_MAX_LAYERS + 1];
    int act[ABN_MAX_LAYERS];
    const float* d_out;            // [rows, dims[n_layers]]: d loss / d output, or d loss / d (pre-activation) of the last layer
    int d_out_is_dz;
    const float* a_top;                   // [rows, dims[n_layers]] the last layer's output, row-major (needed unless d_out_is_dz)
    const char* tp[ABN_MAX_LAYERS];       // forward's transposed images: tp[l + 1] holds the outputs of layer l
    float* dx;                            // optional [rows, dims[0]]: gradient w.r.t. the inputs (needs wpt[0])
    const float* mask[ABN_MAX_LAYERS];
    const char* wpt[ABN_MAX_LAYERS];      // packed W_l^T images, l >= 1
    const char* wbase;                    // the packed images' common buffer and its size
    int64_t wbytes;
    char* dzp[ABN_MAX_LAYERS];            // out: transposed planes of dZ_l (dims[l+1] features)
    float* amax_dz[ABN_MAX_LAYERS];       // fp16 x 2: dzp[l]'s maxima, PL_AMAX floats per 32-row block
    int64_t tp_steps;
    const unsigned long long* drop_seed;  // the forward's in-kernel dropout, regenerated here (null: off)
    float drop_p;
    // The pair loss inside the chain (abn_tower_backward_loss; loss_kind < 0: d_out is given): rows are
    // [tower 1: pairs 0 .. B-1 | tower 2: pairs 0 .. B-1], the embeddings are a_top, and the first phase
    // computes what abn_pair_loss_dz would have written to d_out -- same arithmetic (loss.hip), fp64 per row.
    int loss_kind;
    int y_dtype;
    const void* y;
    int B;
    double margin, scale;
    double* loss_partial;                 // one per workgroup
    unsigned* loss_counter;               // ticket (zero before, zero after)
    float* loss_out;
    // padded batches (a captured step serving a bucket of batch sizes): only the first *n_valid pairs are real,
    // the others contribute no loss term and get dz = 0 (so nothing of them reaches a gradient); avg divides
    // by *n_valid.  null: all B pairs.
    const int* n_valid;
    double* loss_accum;                   // optional: the call's loss is also added here (an epoch's running sum)
#ifdef ABN_STAMPS
    unsigned long long* stamps;           // diagnostic build (tools/dgrad_stamps.py): [workgroup][64] s_memtime per phase
#endif
};
#ifdef ABN_STAMPS
#define DSTAMP(slot) do { if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 64 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DSTAMP(slot) do {} while (0)
#endif

// dZ_{l-1} from dZ_l (in img, pl_steps(dims[l+1]) steps): output features = the dims[l] inputs of layer l
template <int NP, int BPW, int KS>
__device__ __forceinline__ void planes_dgrad_layer(const PlanesBwdP& p, int l, char* __restrict__ img,
                                                   float* __restrict__ part, const bf16x8* idf, int wave, int lane, int row0, float& ainv,
                                                   WeightRing<NP>& ring)
{
    const int N = p.dims[l + 1], K = p.dims[l];        // sum over N, K output features
    const int nsteps = pl_steps(N), nblk = (K + 31) / 32;
    const int r = lane & 31, h = lane >> 5;
    const WaveShare<BPW, KS> ws(wave, nblk, nsteps);
    const int blk0 = ws.blk0;

    f32x16 acc[BPW];
#pragma unroll
    for (int j = 0; j < BPW; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][q] = 0.0f;
    float cinv[BPW];                                   // fp16 x 2: what turns an accumulator into the product
#pragma unroll
    for (int j = 0; j < BPW; ++j) cinv[j] = NP == 2 && ws.active ? packed_inv(p.wpt[l], nblk, nsteps, blk0 + j) * ainv : 1.0f;
    {
        // the ring is handed on where the next layer down gives every wave the same share (two blocks, all steps)
        bool chain_next = false;
        int nblk2 = 1, nsteps2 = 0, ioff2 = 0;
        if constexpr (PL_HANDOVER && NP <= 2 && BPW == 2 && KS == 1) {
            if (l - 1 >= (p.dx ? 0 : 1)) {
                nblk2 = (p.dims[l - 1] + 31) / 32;
                nsteps2 = pl_steps(K);
                ioff2 = (int)(p.wpt[l - 1] - p.wbase);
                chain_next = nblk2 > PL_WAVES;
            }
        }
        if (!ws.active) ring.filled = false;
        const int ioff = (int)(p.wpt[l] - p.wbase);
        int wv[BPW], dnext[BPW];
#pragma unroll
        for (int j = 0; j < BPW; ++j) {
            const int blk = blk0 + j < nblk ? blk0 + j : nblk - 1;
            const int o = ioff + (blk * nsteps + ws.s_first) * (NP * 1024);
            const int blk2 = blk0 + j < nblk2 ? blk0 + j : nblk2 - 1;
            wv[j] = o + lane * 16;
            dnext[j] = ioff2 + blk2 * nsteps2 * (NP * 1024) - o;
        }
        if (ws.active) ring_kloop<NP, BPW>(acc, ring, wv, dnext, chain_next, img, ws.s_first, ws.my_steps, lane);
    }
    DSTAMP(4 + 4 * (p.n_layers - 1 - l));

    const float* __restrict__ mask = l >= 1 ? p.mask[l - 1] : nullptr;
    const DropGen drop = make_drop(l >= 1 && !mask ? p.drop_seed : nullptr, p.drop_p, l - 1);
    const bool masked = mask || drop.on;
    const int gr = row0 + r;
    const bool row_ok = gr < p.rows;
    const int grc = row_ok ? gr : p.rows - 1;
    // act'(a) comes from the forward's transposed image of this layer's input (lane = batch row here,
    // feature on the lane there: a gather of 4- or 2-byte elements, 8 rows of one feature per 32 bytes)
    f32x4 av[BPW][4], mv[BPW][4];
    if (ws.active && ws.khalf == 0 && l >= 1) {
        const int row_off = tgather_row_offset<NP>(r);
#pragma unroll
        for (int j = 0; j < BPW; ++j) {
            const int blk = blk0 + j < nblk ? blk0 + j : nblk - 1;
            const char* tile = p.tp[l] + ((int64_t)blk * p.tp_steps + 2 * blockIdx.x) * tile_bytes<NP>();
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int e = 0; e < 4; ++e) av[j][g][e] = tgather<NP>(tile, row_off, 8 * g + 4 * h + e);
                const int k = 32 * (blk0 + j) + 4 * h + 8 * g;
                if (mask) mv[j][g] = *reinterpret_cast<const f32x4*>(mask + (int64_t)grc * K + (k < K ? k : K - 4));
                else if (drop.on) mv[j][g] = drop4(drop, gr, k);
            }
        }
    }
    auto finish = [&]() {
        if constexpr (NP == 2) {
#pragma unroll
            for (int j = 0; j < BPW; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[j][q] *= cinv[j];
        }
        if (l == 0) return;                          // d loss / d input: the plain product
        with_act(p.act[l - 1], [&](auto tag) {
            constexpr int ACT = decltype(tag)::value;
#pragma unroll
            for (int j = 0; j < BPW; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bool live = 32 * (blk0 + j) + 4 * h + 8 * g < K;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[j][4 * g + e] * act_grad(av[j][g][e], ACT);
                        if (masked) v *= mv[j][g][e];
                        acc[j][4 * g + e] = live ? v : 0.0f;
                    }
                }
        });
    };
    float* const sc = part + PL_PART_BYTES / 4;
    const bool rescale = NP == 2 && l >= 1;        // dZ_{l-1} is written transposed (and multiplied again)
    auto park_max = [&]() {
        float m = 0.0f;
        if (ws.active && ws.khalf == 0) {
#pragma unroll
            for (int j = 0; j < BPW; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) m = fmaxf(m, fabsf(acc[j][q]));
        }
        sc[wave * 64 + lane] = m;
    };
    if (KS == 1) {
        if (ws.active) finish();
        if (rescale) park_max();
    } else if (ws.active && ws.khalf == 1) {
#pragma unroll
        for (int q = 0; q < 16; ++q) part[((wave & 3) * 16 + q) * 64 + lane] = acc[0][q];
    }
    DSTAMP(5 + 4 * (p.n_layers - 1 - l));
    __syncthreads();                               // every wave is done reading img
    DSTAMP(6 + 4 * (p.n_layers - 1 - l));
    if (KS == 2 && ws.active && ws.khalf == 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[0][q] += part[(wave * 16 + q) * 64 + lane];
        finish();
    }
    float osc = 1.0f;
    if (rescale) {
        if (KS == 2) { park_max(); __syncthreads(); }
        float oinv;
        const float m = row_scales(sc, wave, lane, osc, oinv);
        ainv = oinv;
        if (wave == 0) store_amax_rows(p.amax_dz[l - 1] + (int64_t)blockIdx.x * PL_AMAX, m, 0.0f, lane);
    }
    const float* const inv_tab = sc + PL_WAVES * 64 + wave * 32;
    if (ws.active && ws.khalf == 0) {
#pragma unroll
        for (int j = 0; j < BPW; ++j) {
            const int blk = blk0 + j;
            if (blk < nblk) {
                Frag<NP> f[2];
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    const f32x4 v0 = {acc[j][8 * t2], acc[j][8 * t2 + 1], acc[j][8 * t2 + 2], acc[j][8 * t2 + 3]};
                    const f32x4 v1 = {acc[j][8 * t2 + 4], acc[j][8 * t2 + 5], acc[j][8 * t2 + 6], acc[j][8 * t2 + 7]};
                    if (l == 0) {
                        const int k = 32 * blk + 16 * t2 + 4 * h;
                        if (row_ok && k < K) *reinterpret_cast<f32x4*>(p.dx + (int64_t)gr * K + k) = v0;
                        if (row_ok && k + 8 < K) *reinterpret_cast<f32x4*>(p.dx + (int64_t)gr * K + k + 8) = v1;
                    } else {
                        f[t2] = make_frag<NP>(v0, v1, osc);
                        store_frag<NP>(img + (int64_t)(2 * blk + t2) * (NP * 1024) + lane * 16, f[t2]);
                    }
                }
                if (l >= 1) emit_planes<NP>(p.dzp[l - 1] + ((int64_t)blk * p.tp_steps + 2 * blockIdx.x) * tile_bytes<NP>(), f, idf, lane, -1, 0, inv_tab);
            }
        }
    }
    if (l - 1 >= 1 || (l == 1 && p.dx)) {
        const int next_steps = pl_steps(K);
        const bf16x8 z = {};
        for (int s = 2 * nblk + wave; s < next_steps; s += PL_WAVES)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<bf16x8*>(img + ((int64_t)s * NP + pl) * 1024 + lane * 16) = z;
    }
    __syncthreads();
    DSTAMP(7 + 4 * (p.n_layers - 1 - l));
}

template <int NP>
__global__ __launch_bounds__(PL_NT) void tower_dgrad_planes_kernel(PlanesBwdP p)
{
    extern __shared__ __attribute__((aligned(16))) char pl_smem[];
    char* const img = pl_smem;
    float* const part = reinterpret_cast<float*>(pl_smem + PL_MAXSTEPS * NP * 1024);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * PL_ROWS;
    const int top = p.n_layers - 1;
    const int NT = p.dims[top + 1];
    bf16x8 idf[2];
    make_identity<NP>(idf, lane);
#ifdef PL_EXP_PRIO
    if ((PL_EXP_PRIO == 1) == (wave >= 4)) __builtin_amdgcn_s_setprio(1);
#endif
    DSTAMP(0);

    // the pair loss, when it rides along: per-row coefficients of d loss / d e = partner * inv - self * kself
    double* const coef = reinterpret_cast<double*>(part);          // [32][2]  (the K-split buffer is idle here)
    double* const term_s = coef + 64;                              // [32]
    int* const is_last_s = reinterpret_cast<int*>(term_s + 32);
    const bool with_loss = p.loss_kind >= 0;
    if (with_loss) {
        const int B = p.B;
        const int Bv = p.n_valid ? *p.n_valid : B;
        const double lscale = p.n_valid && p.scale != 1.0 ? 1.0 / (double)(Bv > 0 ? Bv : 1) : p.scale;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int lr = 4 * wave + 2 * it + (lane >> 5), l = lane & 31;
            const int g = row0 + lr;
            double inv = 0.0, kself = 0.0, term = 0.0;
            const int tower = g < p.rows && g >= B ? 1 : 0, pi_ = g < p.rows ? g - tower * B : 0;
            const bool ok = g < p.rows && pi_ < Bv;
            const float* a = p.a_top + (int64_t)pi_ * NT;           // e1[pair], e2[pair]: the order loss.hip sums in
            const float* b = p.a_top + (int64_t)(B + pi_) * NT;
            double dot = 0.0, s11 = 0.0, s22 = 0.0;
            for (int c = l; c < NT / 4; c += 32) {
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
                switch (p.y_dtype) {
                    case ABN_Y_I8: v = ((const int8_t*)p.y)[pi_]; break;
                    case ABN_Y_I32: v = ((const int32_t*)p.y)[pi_]; break;
                    case ABN_Y_I64: v = (double)((const int64_t*)p.y)[pi_]; break;
                    case ABN_Y_F32: v = ((const float*)p.y)[pi_]; break;
                    default: v = ((const double*)p.y)[pi_]; break;
                }
                const int code = v == 1.0 ? 1 : (v == -1.0 ? -1 : 0);
                double dcos;
                if (p.loss_kind == ABN_LOSS_COSCOS2) {
                    if (code == 1) { term = (1.0 - cs) * 0.5; dcos = -0.5; }
                    else if (code == -1) { term = cs * cs; dcos = 2.0 * cs; }
                    else { term = cs; dcos = 1.0; }
                } else {
                    if (code == 1) { term = 1.0 - cs; dcos = -1.0; }
                    else if (code == -1) { const double hh = cs - p.margin; term = hh > 0.0 ? hh : 0.0; dcos = hh >= 0.0 ? 1.0 : 0.0; }
                    else { term = cs; dcos = 1.0; }
                }
                dcos *= lscale;
                inv = dcos / (c1 * c2);
                const double k1 = n1 > 0.0 ? dcos * cs / (c1 * n1) : 0.0;
                const double k2 = n2 > 0.0 ? dcos * cs / (c2 * n2) : 0.0;
                kself = tower ? k2 : k1;
                if (tower) term = 0.0;                               // a pair's term counts once
            }
            if (l == 0) { coef[2 * lr] = inv; coef[2 * lr + 1] = kself; term_s[lr] = term; }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double sum = 0.0;
            for (int i = 0; i < 32; ++i) sum += term_s[i];
            *is_last_s = abn_ticket_publish(&p.loss_partial[blockIdx.x], sum, p.loss_counter, gridDim.x);      // (common.h)
        }
        __syncthreads();
        if (*is_last_s && wave == 0) {                            // the last workgroup to arrive: fixed-order sum of all partials
            double sum = 0.0;
            for (int i = lane; i < (int)gridDim.x; i += 64) sum += abn_ticket_partial(&p.loss_partial[i]);
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 64);
            if (lane == 0) {
                const float lv = (float)(sum * lscale);
                *p.loss_out = lv;
                if (p.loss_accum) *p.loss_accum += (double)lv;        // (one thread of one workgroup per call, calls in stream order)
                __hip_atomic_store(p.loss_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }

    DSTAMP(1);
    // dZ of the last layer -> operand fragments and transposed planes
    const int steps_t = pl_steps(NT), blocks_t = steps_t / 2;
    const int gr = row0 + r;
    const bool row_ok = gr < p.rows;
    const float* __restrict__ mask = p.mask[top];
    const DropGen drop_top = make_drop(mask ? nullptr : p.drop_seed, p.drop_p, top);
    const int my_tower = gr >= p.B ? 1 : 0;
    const float* const self_row = p.a_top + (int64_t)(row_ok ? gr : 0) * NT;
    const float* const partner_row = p.a_top + (int64_t)(row_ok && with_loss ? (my_tower ? gr - p.B : gr + p.B) : 0) * NT;
    const double my_inv = with_loss ? coef[2 * r] : 0.0, my_k = with_loss ? coef[2 * r + 1] : 0.0;
    // one 16-feature step of dZ_top for this lane's row
    auto top_step = [&](int s, f32x4* v) {
        v[0] = v[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int c = 16 * s + 4 * h + 8 * u;
            if (row_ok && c < NT) {
                if (with_loss) {
                    const f32x4 es = *reinterpret_cast<const f32x4*>(self_row + c);
                    const f32x4 ep = *reinterpret_cast<const f32x4*>(partner_row + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float o = (float)(ep[e] * my_inv - es[e] * my_k);
                        if (p.act[top] != ACT_NONE) o *= act_grad(es[e], p.act[top]);
                        v[u][e] = o;
                    }
                    if (mask || drop_top.on) {
                        const f32x4 m = mask ? *reinterpret_cast<const f32x4*>(mask + (int64_t)gr * NT + c) : drop4(drop_top, gr, c);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[u][e] *= m[e];
                    }
                } else {
                    v[u] = *reinterpret_cast<const f32x4*>(p.d_out + (int64_t)gr * NT + c);
                    if (!p.d_out_is_dz) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(p.a_top + (int64_t)gr * NT + c);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[u][e] *= act_grad(a[e], p.act[top]);
                        if (mask || drop_top.on) {
                            const f32x4 m = mask ? *reinterpret_cast<const f32x4*>(mask + (int64_t)gr * NT + c) : drop4(drop_top, gr, c);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[u][e] *= m[e];
                        }
                    }
                }
            }
        }
    };
    float ainv = 1.0f;
    if constexpr (NP == 2) {
        // fp16 x 2: a wave's (up to two) blocks wait in registers until the workgroup has agreed on the rows' scales
        float* const sc = part + PL_PART_BYTES / 4;
        f32x4 v[2][2][2];
        float m = 0.0f;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const int kb = wave + PL_WAVES * u;
                v[u][t2][0] = v[u][t2][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (kb < blocks_t) top_step(2 * kb + t2, v[u][t2]);
                m = fmaxf(m, absmax8(v[u][t2][0], v[u][t2][1]));
            }
        sc[wave * 64 + lane] = m;
        __syncthreads();
        float osc;
        m = row_scales(sc, wave, lane, osc, ainv);
        if (wave == 0) store_amax_rows(p.amax_dz[top] + (int64_t)blockIdx.x * PL_AMAX, m, 0.0f, lane);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int kb = wave + PL_WAVES * u;
            if (kb < blocks_t) {
                Frag<NP> f[2];
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    f[t2] = make_frag<NP>(v[u][t2][0], v[u][t2][1], osc);
                    store_frag<NP>(img + (int64_t)(2 * kb + t2) * (NP * 1024) + lane * 16, f[t2]);
                }
                if (kb < pl_blocks(NT))
                    emit_planes<NP>(p.dzp[top] + ((int64_t)kb * p.tp_steps + 2 * blockIdx.x) * tile_bytes<NP>(), f, idf, lane, -1, 0,
                                    sc + PL_WAVES * 64 + wave * 32);
            }
        }
    } else {
        for (int kb = wave; kb < blocks_t; kb += PL_WAVES) {
            Frag<NP> f[2];
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                f32x4 v[2];
                top_step(2 * kb + t2, v);
                f[t2] = make_frag<NP>(v[0], v[1]);
                store_frag<NP>(img + (int64_t)(2 * kb + t2) * (NP * 1024) + lane * 16, f[t2]);
            }
            if (kb < pl_blocks(NT))
                emit_planes<NP>(p.dzp[top] + ((int64_t)kb * p.tp_steps + 2 * blockIdx.x) * tile_bytes<NP>(), f, idf, lane, -1, 0);
        }
    }
    __syncthreads();
    DSTAMP(2);

    WeightRing<NP> ring;
    ring_open(ring, p.wbase, p.wbytes);
    for (int l = top; l >= (p.dx ? 0 : 1); --l) {
        const int nblk = (p.dims[l] + 31) / 32;
        if (nblk > PL_WAVES) planes_dgrad_layer<NP, 2, 1>(p, l, img, part, idf, wave, lane, row0, ainv, ring);
        else if (nblk > PL_WAVES / 2 || pl_steps(p.dims[l + 1]) % (2 * PL_DEPTH) != 0) planes_dgrad_layer<NP, 1, 1>(p, l, img, part, idf, wave, lane, row0, ainv, ring);
        else planes_dgrad_layer<NP, 1, 2>(p, l, img, part, idf, wave, lane, row0, ainv, ring);
    }
}

// ---------------------------------------------------------------------------------------------
// backward of a BatchNorm tower: one launch per layer, top down.  The launch of layer l turns d loss / d a_l
// into dz_l on the way into the operand image -- dy = da act'(a), dz = gamma invstd / n (n dy - s1 - xhat s2)
// mask, with s1 = sum dy and s2 = sum dy xhat over the rows of the forward_once call (they span every
// workgroup's rows: hence the launch boundary) -- leaves dz_l as the transposed image the weight-gradient
// launch reads, multiplies by W_l, stores d loss / d a_{l-1} and the per-workgroup sums of dy_{l-1} and
// dy_{l-1} xhat_{l-1}; a small kernel (tower.hip) adds them in a fixed order before the next launch.
// ---------------------------------------------------------------------------------------------
struct BnBwdP {
    int l, rows, rows_call;
    float n_stat;                  // rows the statistics of a call span: rows_call
    const float* n_stat_dev;       // ... or (cross-replica statistics) [n_calls] on the device: the replicas' rows, all-reduced by the forward
    int N, K;                      // dims[l + 1], dims[l]
    int act_l, act_prev;           // activations behind BatchNorm l and l - 1
    const float* da;               // [rows][N] d loss / d a_l
    const float* z;                // [rows][N] the forward's pre-normalisation values: xhat = (z - mean) invstd, as there
    const float* mean;             // [n_calls][N]
    const float* invstd;           // [n_calls][N]
    const float* gamma;            // [N]
    const float* beta;
    const float* s1;               // [n_calls][N]
    const float* s2;
    const float* mask;             // layer l's dropout mask [rows][N] (null: none, or the seed's)
    const unsigned long long* drop_seed;   // the forward's in-kernel dropout, regenerated here (null: off; mask wins)
    float drop_p;
    char* dzp;                     // out: transposed planes of dz_l
    float* amax_dz;                // fp16 x 2: dzp's maxima, PL_AMAX floats per 32-row block
    int64_t tp_steps;
    const char* wpt;               // packed W_l^T (null: no product -- layer 0 without an input gradient)
    float* da_prev;                // out [rows][K]: d loss / d a_{l-1}, or d loss / d input for l == 0
    const float* z_prev;           // [rows][K]  (l >= 1)
    const float* mean_prev;        // [n_calls][K]
    const float* invstd_prev;
    const float* gamma_prev;       // [K]
    const float* beta_prev;
    float* part_out;               // out [workgroup][2][PL_MAXW]: sums over its 32 rows of dy_{l-1}, dy_{l-1} xhat_{l-1}
    const int* n_valid;            // a padded batch (abn_tower_desc.n_valid): the statistics span *n_valid rows per call, the rows behind get dz = 0
};

template <int NP, int BPW, int KS>
__device__ __forceinline__ void bn_dgrad_product(const BnBwdP& q, char* __restrict__ img, float* __restrict__ part, int wave,
                                                 int lane, int row0, int row_end, int call, float ainv)
{
    const int N = q.N, K = q.K;                       // sum over N, K output features
    const int nsteps = pl_steps(N), nblk = (K + 31) / 32;
    const int r = lane & 31, h = lane >> 5;
    const WaveShare<BPW, KS> ws(wave, nblk, nsteps);
    const int blk0 = ws.blk0;
    f32x16 acc[BPW];
#pragma unroll
    for (int j = 0; j < BPW; ++j)
#pragma unroll
        for (int x = 0; x < 16; ++x) acc[j][x] = 0.0f;
    if (ws.active) planes_kloop<NP, BPW>(acc, q.wpt, nblk, nsteps, img, blk0, ws.s_first, ws.my_steps, lane);
    if (KS == 2 && ws.active && ws.khalf == 1) {
#pragma unroll
        for (int x = 0; x < 16; ++x) part[((wave & 3) * 16 + x) * 64 + lane] = acc[0][x];
    }
    if (KS == 2) __syncthreads();
    if (!(ws.active && ws.khalf == 0)) return;
    if (KS == 2) {
#pragma unroll
        for (int x = 0; x < 16; ++x) acc[0][x] += part[(wave * 16 + x) * 64 + lane];
    }
    if constexpr (NP == 2) {
#pragma unroll
        for (int j = 0; j < BPW; ++j) {
            const float cinv = packed_inv(q.wpt, nblk, nsteps, blk0 + j) * ainv;
#pragma unroll
            for (int x = 0; x < 16; ++x) acc[j][x] *= cinv;
        }
    }
    const int gr = row0 + r;
    const bool row_ok = gr < row_end;
    const int grc = row_ok ? gr : row0;
    float* const orow = q.da_prev + (int64_t)grc * K;
    float* const pw = q.l >= 1 ? q.part_out + (int64_t)blockIdx.x * (2 * PL_MAXW) : nullptr;
    with_act(q.act_prev, [&](auto tag) {
        constexpr int ACT = decltype(tag)::value;
#pragma unroll
        for (int j = 0; j < BPW; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = 32 * (blk0 + j) + 8 * g + 4 * h;
                const bool live = n < K;              // K % 4 == 0: four features in or out together
                const f32x4 d4 = {acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
                if (live && row_ok) *reinterpret_cast<f32x4*>(orow + n) = d4;
                if (q.l >= 1) {
                    const int nc = live ? n : K - 4;
                    const f32x4 z4 = *reinterpret_cast<const f32x4*>(q.z_prev + (int64_t)grc * K + nc);
                    const f32x4 mu = *reinterpret_cast<const f32x4*>(q.mean_prev + (int64_t)call * K + nc);
                    const f32x4 is = *reinterpret_cast<const f32x4*>(q.invstd_prev + (int64_t)call * K + nc);
                    f32x4 xh;
#pragma unroll
                    for (int e = 0; e < 4; ++e) xh[e] = (z4[e] - mu[e]) * is[e];
                    const f32x4 ga = *reinterpret_cast<const f32x4*>(q.gamma_prev + nc), be = *reinterpret_cast<const f32x4*>(q.beta_prev + nc);
                    f32x4 sd, sq;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a = act_apply(xh[e] * ga[e] + be[e], ACT);
                        const float dy = live && row_ok ? d4[e] * act_grad(a, ACT) : 0.0f;
                        sd[e] = dy;
                        sq[e] = dy * xh[e];
                    }
                    half_wave_sum4(sd);
                    half_wave_sum4(sq);
                    if (r == 16 && live) {
                        *reinterpret_cast<f32x4*>(pw + n) = sd;
                        *reinterpret_cast<f32x4*>(pw + PL_MAXW + n) = sq;
                    }
                }
            }
    });
}

template <int NP>
__global__ __launch_bounds__(PL_NT) void bn_bwd_layer_kernel(BnBwdP q)
{
    extern __shared__ __attribute__((aligned(16))) char pl_smem[];
    char* const img = pl_smem;
    float* const part = reinterpret_cast<float*>(pl_smem + PL_MAXSTEPS * NP * 1024);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wpc = (q.rows_call + PL_ROWS - 1) / PL_ROWS;        // (the forward's workgroup -> rows map)
    const int call = blockIdx.x / wpc;
    const int row0 = call * q.rows_call + (blockIdx.x - call * wpc) * PL_ROWS;
    const int nv = q.n_valid ? min(max(*q.n_valid, 0), q.rows_call) : q.rows_call;
    const int row_end = call * q.rows_call + nv;
    const int N = q.N;
    bf16x8 idf[2];
    make_identity<NP>(idf, lane);

    // the per-feature vectors of this workgroup's call, parked in the (still idle) K-split buffer
    const float nf = q.n_valid ? (float)(nv > 0 ? nv : 1) : (q.n_stat_dev ? q.n_stat_dev[call] : q.n_stat);
    float* const k_s = part, * const ga_s = part + PL_MAXW, * const be_s = part + 2 * PL_MAXW, * const s1_s = part + 3 * PL_MAXW,
               * const s2_s = part + 4 * PL_MAXW, * const mu_s = part + 5 * PL_MAXW, * const is_s = part + 6 * PL_MAXW;
    for (int c = threadIdx.x; c < N; c += PL_NT) {
        const float ga = q.gamma[c];
        k_s[c] = ga * q.invstd[(int64_t)call * N + c] / nf;
        mu_s[c] = q.mean[(int64_t)call * N + c];
        is_s[c] = q.invstd[(int64_t)call * N + c];
        ga_s[c] = ga;
        be_s[c] = q.beta[c];
        s1_s[c] = q.s1[(int64_t)call * N + c];
        s2_s[c] = q.s2[(int64_t)call * N + c];
    }
    __syncthreads();
    const int steps = pl_steps(N), blocks = steps / 2;
    const int gr = row0 + r;
    const bool row_ok = gr < row_end;                 // rows past the call's end: dz = 0 (zero rows of the image)
    const int grc = row_ok ? gr : row0;
    const float* const drow = q.da + (int64_t)grc * N;
    const float* const zrow = q.z + (int64_t)grc * N;
    const float* const mrow = q.mask ? q.mask + (int64_t)grc * N : nullptr;
    const DropGen drop = make_drop(q.mask ? nullptr : q.drop_seed, q.drop_p, q.l);
    float* const sc = part + PL_PART_BYTES / 4;
    float ainv = 1.0f;
    with_act(q.act_l, [&](auto tag) {
        constexpr int ACT = decltype(tag)::value;
        // one 16-feature step of dz_l for this lane's row
        auto dz_step = [&](int s, f32x4* v) {
            v[0] = v[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int c = 16 * s + 4 * h + 8 * u;
                if (c < N && row_ok) {
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(drow + c), z4 = *reinterpret_cast<const f32x4*>(zrow + c);
                    const f32x4 mu = *reinterpret_cast<const f32x4*>(mu_s + c), is = *reinterpret_cast<const f32x4*>(is_s + c);
                    f32x4 xh;
#pragma unroll
                    for (int e = 0; e < 4; ++e) xh[e] = (z4[e] - mu[e]) * is[e];
                    const f32x4 k4 = *reinterpret_cast<const f32x4*>(k_s + c), ga = *reinterpret_cast<const f32x4*>(ga_s + c);
                    const f32x4 be = *reinterpret_cast<const f32x4*>(be_s + c);
                    const f32x4 a1 = *reinterpret_cast<const f32x4*>(s1_s + c), a2 = *reinterpret_cast<const f32x4*>(s2_s + c);
                    f32x4 m4 = {1.f, 1.f, 1.f, 1.f};
                    if (mrow) m4 = *reinterpret_cast<const f32x4*>(mrow + c);
                    else if (drop.on) m4 = drop4(drop, gr, c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a = act_apply(xh[e] * ga[e] + be[e], ACT);
                        const float dy = d4[e] * act_grad(a, ACT);
                        v[u][e] = k4[e] * (nf * dy - a1[e] - xh[e] * a2[e]) * m4[e];
                    }
                }
            }
        };
        if constexpr (NP == 2) {
            // fp16 x 2: a wave's (up to two) blocks wait in registers until the workgroup has agreed on the rows' scales
            f32x4 v[2][2][2];
            float m = 0.0f;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    const int kb = wave + PL_WAVES * u;
                    v[u][t2][0] = v[u][t2][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (kb < blocks) dz_step(2 * kb + t2, v[u][t2]);
                    m = fmaxf(m, absmax8(v[u][t2][0], v[u][t2][1]));
                }
            sc[wave * 64 + lane] = m;
            __syncthreads();
            float osc;
            m = row_scales(sc, wave, lane, osc, ainv);
            if (wave == 0) store_amax_rows(q.amax_dz + (int64_t)blockIdx.x * PL_AMAX, m, 0.0f, lane);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int kb = wave + PL_WAVES * u;
                if (kb < blocks) {
                    Frag<NP> f[2];
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2) {
                        f[t2] = make_frag<NP>(v[u][t2][0], v[u][t2][1], osc);
                        store_frag<NP>(img + (int64_t)(2 * kb + t2) * (NP * 1024) + lane * 16, f[t2]);
                    }
                    if (kb < pl_blocks(N))
                        emit_planes<NP>(q.dzp + ((int64_t)kb * q.tp_steps + 2 * blockIdx.x) * tile_bytes<NP>(), f, idf, lane, -1, 0,
                                        sc + PL_WAVES * 64 + wave * 32);
                }
            }
        } else {
            for (int kb = wave; kb < blocks; kb += PL_WAVES) {
                Frag<NP> f[2];
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    f32x4 v[2];
                    dz_step(2 * kb + t2, v);
                    f[t2] = make_frag<NP>(v[0], v[1]);
                    store_frag<NP>(img + (int64_t)(2 * kb + t2) * (NP * 1024) + lane * 16, f[t2]);
                }
                if (kb < pl_blocks(N))
                    emit_planes<NP>(q.dzp + ((int64_t)kb * q.tp_steps + 2 * blockIdx.x) * tile_bytes<NP>(), f, idf, lane, -1, 0);
            }
        }
    });
    __syncthreads();
    if (!q.wpt) return;

    const int nblk = (q.K + 31) / 32;
    if (nblk > PL_WAVES) bn_dgrad_product<NP, 2, 1>(q, img, part, wave, lane, row0, row_end, call, ainv);
    else if (nblk > PL_WAVES / 2 || pl_steps(N) % (2 * PL_DEPTH) != 0) bn_dgrad_product<NP, 1, 1>(q, img, part, wave, lane, row0, row_end, call, ainv);
    else bn_dgrad_product<NP, 1, 2>(q, img, part, wave, lane, row0, row_end, call, ainv);
}

// ---------------------------------------------------------------------------------------------
// backward, weight gradients: dW_l = dZ_l^T [A_{l-1} | 1], split over the batch rows into slabs
// ---------------------------------------------------------------------------------------------
// One workgroup: a tile of one layer's [N][K + 1] gradient over a range of row steps.  Its eight
// waves form a WN x WK grid, each wave owning TN x TK blocks of 32 x 32.  Both operands arrive by
// LDS-DMA, 1 KB per wave instruction, lane-linear, one row step per stage, three stages in flight.
// The launch is bound by the operand stream (every dZ block is read once per tile column, every
// activation block once per tile row, from the Infinity Cache at ~6 TB/s), so the tiles are as
// large as LDS allows:
//   shape 0  TN 2 TK 2 WN 4 WK 2   256 x 128   the general case
//   shape 1  TN 1 TK 2 WN 8 WK 1   256 x  64   at most two column blocks (the first layer: K + 1 = 41)
//   shape 2  TN 2 TK 2 WN 2 WK 4   128 x 256   at most four row blocks (the output layer: N = 100)
// (256 x 256 tiles halve the operand stream again but need twice the slabs to fill the chip: the slab
// writes and their reduction then cost what the operands saved -- measured, 77 + 23 us against 81 + 16.)
// bf16: LDS-DMA ring of 1 KB tiles, row steps in flight + the one being consumed (HBM latency under this
// load: 3-4 us).  bf16 x 3: the fp32 tiles (2 KB) are fetched into REGISTERS four row steps ahead, split
// once by the wave that fetched them -- not by the four / two waves that multiply them: that was 116 us,
// VALU-bound -- and written as three planes (3 KB per block) into a ring of four LDS stages.
template <int NP> constexpr int wg_stages() { return NP >= 2 ? 4 : 6; }
constexpr int WG_REG_DEPTH = 4;
constexpr int WG_MAX_BLOCKS = 12;                 // operand blocks of one row step (shape 0: 8 + 4)
struct WgradLayer {
    const char* dzp;       // transposed planes of dZ_l        [nblk][steps]
    const char* ap;        // transposed planes of [A_{l-1} | 1] [kblk][steps]
    int N, K;              // gradient is [N][K] (+ bias column K)
    int nblk, kblk;
    int shape;
    int tiles_n, tiles_k;
    int splits;            // slabs this layer's sum over the rows is cut into
    int first_wg;          // workgroups [first_wg, first_wg + tiles_n * tiles_k * splits) belong to this layer
    int64_t slab_off;      // float offset of this layer's packed (dW | db) region inside a slab
    const float* amax_dz;  // fp16 x 2: the two images' maxima, PL_AMAX floats per 32-row block (a slab's scales)
    const float* amax_a;
};
struct WgradP {
    int n_layers;          // in launch order
    WgradLayer L[ABN_MAX_LAYERS];
    float* slabs;
    int64_t slab_stride;   // floats
    int64_t tp_steps;
    int xcd_groups;        // workgroup -> (layer, slab, tile) through wgrad_place (0: in launch order, first_wg)
#ifdef ABN_STAMPS
    unsigned long long* stamps;     // diagnostic build: [workgroup][16] s_memtime at start / loop entry / loop exit / end, N * 1024 + K, steps, XCC_ID, HW_ID, after 16 / 32 / 48 steps
#endif
};

// Placement.  The tiles_n * tiles_k tiles of one (layer, slab) -- a group -- read the same row steps of the
// same two operand images; workgroups b and b + 8 are observed to share an XCD (round-robin dispatch: speed
// only, nothing depends on it), so a group's tiles sit at blockIdx.x = x + 8 (slot ...) of one x and the
// XCD's L2 serves the re-reads.  Groups are dealt round-robin over the eight x in launch order (every XCD
// gets the same mix of heavy and light groups, heavy first).  Grid = 8 x wgrad_slots(p, x) maximised over x.
__host__ __device__ inline int wgrad_group_count(int g0, int splits, int x, int* first)
{
    *first = g0 + ((x - g0) & 7);                  // the layer's first group with index = x (mod 8)
    return *first < g0 + splits ? (g0 + splits - 1 - *first) / 8 + 1 : 0;
}
static inline int wgrad_slots(const WgradP& p, int x)
{
    int slots = 0, g0 = 0, first;
    for (int i = 0; i < p.n_layers; ++i) {
        slots += wgrad_group_count(g0, p.L[i].splits, x, &first) * p.L[i].tiles_n * p.L[i].tiles_k;
        g0 += p.L[i].splits;
    }
    return slots;
}
static inline void wgrad_shape(int nblk, int kblk, int* shape, int* bn, int* bk, bool small_tiles = false)
{
    // shape 3 (fp16 x 2, large batches: tower.hip wgrad_tile128): 128 x 128 tiles for EVERY layer -- eight operand blocks per row
    // step, so the launch needs 64 KB of LDS instead of 96 and TWO workgroups share a CU: the light layers' workgroups run beside
    // the heavy ones instead of behind them, and two workgroups out of step fill each other's barrier waits (C2: -4 us per step,
    // with BatchNorm -14; the 500 x 500 layers also write half the slab bytes).  Small batches keep the wider tiles (485 pairs: +2 us)
    if (small_tiles) { *shape = 3; *bn = 4; *bk = 4; return; }      // (every layer: the launch then needs 64 KB of LDS, two workgroups per CU)
    if (kblk <= 2) { *shape = 1; *bn = 8; *bk = 2; }
    else if (nblk <= 4) { *shape = 2; *bn = 4; *bk = 8; }
    else { *shape = 0; *bn = 8; *bk = 4; }
}

template <int NP>
constexpr size_t wgrad_lds_bytes() { return (size_t)wg_stages<NP>() * WG_MAX_BLOCKS * (NP * 1024) + (NP == 2 ? 128 : 0); }

template <int NP, int TN, int TK, int WN, int WK>
__device__ __forceinline__ void wgrad_tile(const WgradP& p, const WgradLayer& L, char* __restrict__ smem, int nb0, int kb0,
                                           int s_begin, int s_end, int split, int wave, int lane)
{
    static_assert(WN * WK == PL_WAVES, "eight waves");
    constexpr int BN = WN * TN, BK_ = WK * TK, NB = BN + BK_;     // operand blocks per row step: dZ blocks first
    constexpr int PER_WAVE = (NB + PL_WAVES - 1) / PL_WAVES;
    constexpr int FR = tile_bytes<NP>();                 // one (block, row step) tile, in HBM and in LDS
    constexpr int DI = FR / 1024;                        // DMA instructions per tile (1 KB each)
    constexpr int STAGE = NB * FR;
    constexpr int WG_STAGES = wg_stages<NP>();
    static_assert(NB <= WG_MAX_BLOCKS, "LDS stage");
    constexpr int PSTAGE = NB * NP * 1024;               // bf16 x 3, fp16 x 2: one LDS stage = [block][plane][1 KB]
    const int wn = wave % WN, wk = wave / WN;
    // this wave's share of the DMA: operand blocks wave, wave + 8, ... below NB (n_mine of them: the wait
    // for "my part of step c" counts this wave's own instructions)
    const char* src[PER_WAVE];
    int dst[PER_WAVE];
    int n_mine = 0;
#pragma unroll
    for (int u = 0; u < PER_WAVE; ++u) {
        const int b = wave + PL_WAVES * u;
        const bool is_dz = b < BN;
        const int ob = is_dz ? nb0 + b : kb0 + (b - BN);
        const int ob_max = is_dz ? L.nblk : L.kblk;
        src[u] = (is_dz ? L.dzp : L.ap) + (int64_t)(ob < ob_max ? ob : ob_max - 1) * p.tp_steps * FR + lane * 16;
        dst[u] = b * FR;
        if (b < NB) n_mine = u + 1;
    }
    auto dma = [&](int step, int stage) {
#ifdef WEXP_NODMA
        return;
#endif
#pragma unroll
        for (int u = 0; u < PER_WAVE; ++u)
            if (u < n_mine) {
#pragma unroll
                for (int d = 0; d < DI; ++d)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(src[u] + (int64_t)step * FR + d * 1024),
                        (__attribute__((address_space(3))) void*)(smem + stage * STAGE + dst[u] + d * 1024), 16, 0, 0);
            }
    };
    f32x16 acc[TN][TK];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;
#ifdef ABN_STAMPS
#define WSTAMP(slot) do { if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
    WSTAMP(0);
    if (p.stamps && threadIdx.x == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        p.stamps[(size_t)blockIdx.x * 16 + 4] = (unsigned long long)(L.N * 1024 + L.K);
        p.stamps[(size_t)blockIdx.x * 16 + 5] = (unsigned long long)(s_end - s_begin);
        p.stamps[(size_t)blockIdx.x * 16 + 6] = xcc & 15;
        p.stamps[(size_t)blockIdx.x * 16 + 7] = hwid;
    }
#else
#define WSTAMP(slot) do {} while (0)
#endif

    // blocks past the matrix are padding: a wave whose blocks all are skips the MFMAs (not the DMAs or barriers)
    const bool live = nb0 + wn * TN < L.nblk && kb0 + wk * TK < L.kblk;
    const int n_steps = s_end - s_begin;
    // Steps c + 2 .. c + WG_STAGES - 1 are in flight while step c is summed and step c + 1's fragments are
    // read into the second register set (every wave sits behind the same barrier: without that
    // read-ahead they all wait for LDS together, then all issue MFMAs together).  A wave issues the
    // same DMAs for every step (clamped repeats past the end), so "my part of step c + 1 has landed"
    // is vmcnt((WG_STAGES - 3) steps' worth of my DMAs).
    auto step_at = [&](int c) { return s_begin + (c < n_steps ? c : n_steps - 1); };
    static_assert(PER_WAVE == 2 || NP >= 2, "n_mine is 1 or 2 below (the DMA path)");
    auto wait_steps_left = [&](auto steps_tag) {
        constexpr int S = decltype(steps_tag)::value;
        if (n_mine == 2) wait_vmcnt<S * 2 * DI>();
        else if (n_mine == 1) wait_vmcnt<S * DI>();
    };
    struct Frags { bf16x8 a[TN][NP], b[TK][NP]; };
    auto read_frags = [&](Frags& f, int c) {
        if constexpr (NP >= 2) {
            const char* st = smem + (c % WG_STAGES) * PSTAGE + lane * 16;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
                for (int i = 0; i < TN; ++i) f.a[i][pl] = *reinterpret_cast<const bf16x8*>(st + (wn * TN + i) * (NP * 1024) + pl * 1024);
#pragma unroll
                for (int j = 0; j < TK; ++j) f.b[j][pl] = *reinterpret_cast<const bf16x8*>(st + (BN + wk * TK + j) * (NP * 1024) + pl * 1024);
            }
        } else {
            const char* st = smem + (c % WG_STAGES) * STAGE + lane * 16;
#pragma unroll
            for (int i = 0; i < TN; ++i) f.a[i][0] = *reinterpret_cast<const bf16x8*>(st + (wn * TN + i) * FR);
#pragma unroll
            for (int j = 0; j < TK; ++j) f.b[j][0] = *reinterpret_cast<const bf16x8*>(st + (BN + wk * TK + j) * FR);
        }
    };
    auto mfmas = [&](const Frags& f) {
#ifdef WEXP_NOMFMA
        for (int i = 0; i < TN; ++i) for (int j = 0; j < TK; ++j) acc[i][j][0] += (float)f.a[i][0][0] * (float)f.b[j][0][1];
        return;
#endif
#pragma unroll
        for (int u = 0; u < Products<NP>::N; ++u)
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TK; ++j)
                    acc[i][j] = pl_mfma<NP>(f.a[i][Products<NP>::A[u]], f.b[j][Products<NP>::B[u]], acc[i][j]);
    };
    Frags fr[2];
    float out_inv = 1.0f, out_inv_ones = 1.0f;           // fp16 x 2: what turns an accumulator into the sum (the bias column: below)
    if constexpr (NP >= 2) {
        // Register ring: slot k % 4 holds this wave's share of step k's fp32 tiles, fetched four steps before
        // it is split (raw buffer loads: the compiler counts their vmcnt and leaves them where they stand).
        // The share is the same for every wave -- one whole tile (operand block `wave`) and one 1 KB half of
        // a tile of the blocks 8 .. NB-1 -- so that the step's body is ONE branch-free scheduling region:
        // only there can the ~70 VALU instructions of the split be woven between the MFMAs
        // (sched_group_barrier); in a block of their own, in front of the MFMAs, they idle the matrix cores
        // of a SIMD whose two waves sit behind the same barrier (measured: +16 us).
        // Iteration c, behind ONE barrier: split step c + 2 into LDS stage (c + 2) % 4 and refill its slot
        // with step c + 6; read step c + 1's fragments (second register set); step c's MFMAs.  The loop
        // runs to a multiple of four steps: steps past the end are written as zeros.
        static_assert(WG_STAGES == 4, "stage arithmetic below");
        // register ring depth: steps in flight ahead of the split (-DWG_DEPTH2=8: 0.1747 against 0.1735 ms / step with four on the wide
        // tiles; on the 128 x 128 kernel eight spill (52 bytes at its 128 registers: 56 against 45 us), and eight with ONE fragment set
        // instead of two fit and give 47.2 against 47.9 with four -- the loads are not what is exposed)
#ifndef WG_DEPTH2
#define WG_DEPTH2 4
#endif
        constexpr int RD = NP == 2 ? WG_DEPTH2 : WG_REG_DEPTH;
        static_assert(RD == 4 || RD == 8, "slot arithmetic below");
        static_assert(NB >= PL_WAVES && NB <= 12, "one whole tile per wave plus halves of the rest");
        constexpr bool HALVES = NB > PL_WAVES;                     // (NB == 8, the 128 x 128 tile: every wave converts exactly one tile)
        constexpr int NHALF = HALVES ? 2 * (NB - PL_WAVES) : 1;    // 1 KB halves of the tiles 8 .. NB-1 (8: one each; 4: shared, written twice)
        const int hb = HALVES ? PL_WAVES + ((wave % NHALF) >> 1) : 0, hh = wave & 1;          // this wave's half: tile hb, elements 4 hh .. 4 hh + 3
        auto image_of = [&](int b) -> const char* {
            const bool is_dz = b < BN;
            const int ob = is_dz ? nb0 + b : kb0 + (b - BN);
            const int ob_max = is_dz ? L.nblk : L.kblk;
            return (is_dz ? L.dzp : L.ap) + (int64_t)(ob < ob_max ? ob : ob_max - 1) * p.tp_steps * FR;
        };
        const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(image_of(wave)), 0, (int)(p.tp_steps * FR), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(image_of(hb)), 0, (int)(p.tp_steps * FR), 0x00020000);
        v4i rq[RD][3];
        const int n4 = (n_steps + RD - 1) / RD * RD;
        // fp16 x 2: one power of two per operand for this slab's rows, from the maxima the chains left per 32-row block
        // (looked up behind the first tile loads: one round trip, not two, before the first split)
        float sc_whole = 1.0f, sc_half = 1.0f;
        auto slab_scales = [&]() {
        if constexpr (NP == 2) {
            float* const red = reinterpret_cast<float*>(smem + WG_STAGES * PSTAGE);      // [2][8], behind this shape's stages
            const int rb0 = s_begin >> 1, cnt = (((s_end + 1) >> 1) - rb0) * PL_AMAX;
            float md = 0.0f, ma = 0.0f;
            for (int i = threadIdx.x; i < cnt; i += PL_NT) {
                md = fmaxf(md, L.amax_dz[(int64_t)rb0 * PL_AMAX + i]);
                ma = fmaxf(ma, L.amax_a[(int64_t)rb0 * PL_AMAX + i]);
            }
            md = wave_max(md); ma = wave_max(ma);
            if (lane == 0) { red[wave] = md; red[8 + wave] = ma; }
            __syncthreads();
#pragma unroll
            for (int w = 0; w < PL_WAVES; ++w) { md = fmaxf(md, red[w]); ma = fmaxf(ma, red[8 + w]); }
            float sd, sa, id, ia;
            scale_of(md, sd, id);
            scale_of(ma, sa, ia);
            out_inv = id * ia;
            out_inv_ones = id;
            // The column of ones (the bias gradient) keeps a scale of its own, 1: under the activations' scale it would
            // sink into fp16's subnormals once the activations are ~1e5 and more (an unbounded activation on large inputs;
            // the lane holding that column converts with 1 and its sums leave with the dZ inverse alone).
            const bool ones_lane = (lane & 31) == L.K % 32;
            const int ones_blk = L.K / 32 - kb0 + BN;    // its operand block in this tile (if it has it)
            sc_whole = wave < BN ? sd : (ones_lane && wave == ones_blk ? 1.0f : sa);      // this wave's whole tile: operand block `wave`
            sc_half = ones_lane && hb == ones_blk ? 1.0f : sa;                            // (blocks 8 .. NB-1 are activation blocks in every shape)
            static_assert(BN <= PL_WAVES, "the half tiles are activation blocks");
        }
        };
        auto load = [&](int slot, int k) {
            const int off = step_at(k) * FR;
#ifdef WEXP_NOLOAD
            rq[slot][0] = rq[slot][1] = rq[slot][2] = v4i{k, slot, 1, 2};
#else
            rq[slot][0] = __builtin_amdgcn_raw_buffer_load_b128(rs0, lane * 16, off, 0);
            rq[slot][1] = __builtin_amdgcn_raw_buffer_load_b128(rs0, lane * 16, off + 1024, 0);
            if constexpr (HALVES) rq[slot][2] = __builtin_amdgcn_raw_buffer_load_b128(rs1, lane * 16, off + hh * 1024, 0);
#endif
        };
        auto convert = [&](int slot, int k) {
#ifdef WEXP_NOCONVERT
            if (k >= 0) return;
#endif
            const bool real = k < n_steps;
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            // steps past the slab's end (the loop runs in fours) contribute zeros: for fp16 x 2 through a ZERO SCALE
            // -- the loads are clamped re-reads of real, finite values -- instead of twelve selects per lane and step
            const bool by_scale = NP == 2;
            const f32x4 v0 = real || by_scale ? __builtin_bit_cast(f32x4, rq[slot][0]) : z4;
            const f32x4 v1 = real || by_scale ? __builtin_bit_cast(f32x4, rq[slot][1]) : z4;
            const f32x4 v2 = HALVES && (real || by_scale) ? __builtin_bit_cast(f32x4, rq[slot][2]) : z4;
            const float scw = real ? sc_whole : 0.0f, sch = real ? sc_half : 0.0f;
            char* const st = smem + (k % WG_STAGES) * PSTAGE + lane * 16;
            store_frag<NP>(st + wave * (NP * 1024), make_frag<NP>(v0, v1, NP == 2 ? scw : sc_whole));
            // the half tile: four values -> 8 bytes per plane
            char* const sh = st + hb * (NP * 1024) + hh * 8;
            if constexpr (!HALVES) {
            } else if constexpr (NP == 3) {
                typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                bf16x4 ph, pm, pl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const __bf16 hq = (__bf16)v2[e];
                    const float r1 = v2[e] - (float)hq;
                    const __bf16 mq = (__bf16)r1;
                    ph[e] = hq; pm[e] = mq; pl[e] = (__bf16)(r1 - (float)mq);
                }
                *reinterpret_cast<bf16x4*>(sh) = ph;
                *reinterpret_cast<bf16x4*>(sh + 1024) = pm;
                *reinterpret_cast<bf16x4*>(sh + 2048) = pl;
            } else {
                uint2 ph, pl;
                split_f16x2_pair(sch, v2[0], v2[1], ph.x, pl.x);
                split_f16x2_pair(sch, v2[2], v2[3], ph.y, pl.y);
                *reinterpret_cast<uint2*>(sh) = ph;
                *reinterpret_cast<uint2*>(sh + 1024) = pl;
            }
        };
        if (n_steps > 0) {
#pragma unroll
            for (int k = 0; k < RD; ++k) load(k, k);
            slab_scales();
            convert(0, 0); load(0, RD);
            convert(1, 1); load(1, RD + 1);
            __syncthreads();
            read_frags(fr[0], 0);
        }
        WSTAMP(1);
        for (int c0 = 0; c0 < n4; c0 += RD) {
#ifdef ABN_STAMPS
            if (c0 == 16) WSTAMP(8); else if (c0 == 32) WSTAMP(9); else if (c0 == 48) WSTAMP(10);
#endif
#pragma unroll
            for (int i = 0; i < RD; ++i) {
                const int c = c0 + i;
                __syncthreads();        // step c + 1's planes are in LDS for everybody; nobody still reads stage (c + 2) % 4
                read_frags(fr[(i + 1) & 1], c + 1);
                convert((i + 2) & (RD - 1), c + 2);
                mfmas(fr[i & 1]);       // (a wave whose blocks are all padding multiplies clamped copies: nothing of it is stored)
#pragma unroll
                for (int g = 0; g < TN * TK * Products<NP>::N; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, NP == 2 ? 6 : 4, 0);
                }
                load((i + 2) & (RD - 1), c + 2 + RD);
            }
        }
    } else {
        if (n_steps > 0) {
#pragma unroll
            for (int c = 0; c < WG_STAGES - 1; ++c) dma(step_at(c), c);
            wait_steps_left(std::integral_constant<int, WG_STAGES - 2>{});      // step 0 is in
            __syncthreads();
            if (live) read_frags(fr[0], 0);
        }
        // one step: cur holds step c's fragments; nxt receives step c + 1's
        auto one_step = [&](int c, const Frags& cur, Frags& nxt) {
            wait_steps_left(std::integral_constant<int, WG_STAGES - 3>{});      // my part of step c + 1 is in LDS
            __syncthreads();            // ... and everybody's; everybody has read step c (its stage is not reused before the next barrier)
            dma(step_at(c + WG_STAGES - 1), (c + WG_STAGES - 1) % WG_STAGES);   // into the stage of step c - 1
            if (live) {
                if (c + 1 < n_steps) read_frags(nxt, c + 1);
                mfmas(cur);
            }
        };
        for (int c = 0; c < n_steps; c += 2) {
            one_step(c, fr[0], fr[1]);
            if (c + 1 < n_steps) one_step(c + 1, fr[1], fr[0]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the clamped repeats still target this workgroup's LDS
    WSTAMP(2);

    // acc[i][j][q] of lane (c, h): row n = 32 (nb0 + wn TN + i) + (q & 3) + 8 (q >> 2) + 4 h, column k = 32 (kb0 + wk TK + j) + c
    float* const slab = p.slabs + (int64_t)split * p.slab_stride + L.slab_off;
    const int h = lane >> 5;
#pragma unroll
    for (int j = 0; j < TK; ++j) {
        const int k = 32 * (kb0 + wk * TK + j) + (lane & 31);
        if (k <= L.K) {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int n = 32 * (nb0 + wn * TN + i) + (q & 3) + 8 * (q >> 2) + 4 * h;
                    if (n < L.N) slab[k < L.K ? (int64_t)n * L.K + k : (int64_t)L.N * L.K + n] = NP == 2 ? acc[i][j][q] * (k < L.K ? out_inv : out_inv_ones) : acc[i][j][q];
                }
        }
    }
    WSTAMP(3);
}

// blockIdx -> (layer, slab, tile) of the launch's table
__device__ __forceinline__ bool wgrad_place(const WgradP& p, int& li, int& tn, int& tk, int& split)
{
    li = 0;
    if (p.xcd_groups) {
        const int x = blockIdx.x & 7;
        int slot = blockIdx.x >> 3, g0 = 0;
        for (;; ++li) {
            if (li == p.n_layers) return false;    // this x has fewer slots than the grid's eighth
            int first;
            const int G = p.L[li].tiles_n * p.L[li].tiles_k;
            const int cnt = wgrad_group_count(g0, p.L[li].splits, x, &first);
            if (slot < cnt * G) {
                split = first - g0 + 8 * (slot / G);
                slot %= G;
                tk = slot % p.L[li].tiles_k;
                tn = slot / p.L[li].tiles_k;
                return true;
            }
            slot -= cnt * G;
            g0 += p.L[li].splits;
        }
    }
    while (li + 1 < p.n_layers && (int)blockIdx.x >= p.L[li + 1].first_wg) ++li;
    int local = blockIdx.x - p.L[li].first_wg;
    tk = local % p.L[li].tiles_k; local /= p.L[li].tiles_k;
    tn = local % p.L[li].tiles_n; local /= p.L[li].tiles_n;
    split = local;
    return true;
}

// Every layer on 128 x 128 tiles (shape 3, fp16 x 2): a kernel of its own, so that its registers are counted for
// this shape alone -- 128 instead of the general kernel's 173 -- and TWO workgroups (sixteen waves) share a CU.
#ifndef WG128_WAVES_PER_EU
#define WG128_WAVES_PER_EU 4
#endif
__global__ __launch_bounds__(PL_NT) __attribute__((amdgpu_waves_per_eu(WG128_WAVES_PER_EU, WG128_WAVES_PER_EU))) void wgrad_planes128_kernel(WgradP p)
{
    extern __shared__ __attribute__((aligned(16))) char wg_smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int li, tk, tn, split;
    if (!wgrad_place(p, li, tn, tk, split)) return;
    const WgradLayer& L = p.L[li];
    const int s_begin = (int)(p.tp_steps * split / L.splits), s_end = (int)(p.tp_steps * (split + 1) / L.splits);
    wgrad_tile<2, 1, 2, 4, 2>(p, L, wg_smem, 4 * tn, 4 * tk, s_begin, s_end, split, wave, lane);
}

template <int NP>
__global__ __launch_bounds__(PL_NT) void wgrad_planes_kernel(WgradP p)
{
    extern __shared__ __attribute__((aligned(16))) char wg_smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int li, tk, tn, split;
    if (!wgrad_place(p, li, tn, tk, split)) return;
    const WgradLayer& L = p.L[li];
    const int s_begin = (int)(p.tp_steps * split / L.splits), s_end = (int)(p.tp_steps * (split + 1) / L.splits);
    if (L.shape == 0) wgrad_tile<NP, 2, 2, 4, 2>(p, L, wg_smem, 8 * tn, 4 * tk, s_begin, s_end, split, wave, lane);
    else if (L.shape == 1) wgrad_tile<NP, 1, 2, 8, 1>(p, L, wg_smem, 8 * tn, 2 * tk, s_begin, s_end, split, wave, lane);
    else if (L.shape == 2) wgrad_tile<NP, 2, 2, 2, 4>(p, L, wg_smem, 4 * tn, 8 * tk, s_begin, s_end, split, wave, lane);
    else if constexpr (NP == 2) wgrad_tile<NP, 1, 2, 4, 2>(p, L, wg_smem, 4 * tn, 4 * tk, s_begin, s_end, split, wave, lane);     // shape 3 (fp16 x 2 only)
}

}  // namespace abn
