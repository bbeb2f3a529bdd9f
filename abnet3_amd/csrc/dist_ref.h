// dist_ref.h -- the reference's angular distance (abnet3/utils.py:40-60), float32,
// operation for operation as numpy + libm evaluate it on the plain path (see
// oracle/dtw.c, which tests/golden/cosdist_libm.npz pins to the reference bit for bit):
//
//   x2 = np.sqrt(np.sum(x ** 2, axis=1))   squares rounded, numpy's pairwise summation
//   d  = np.dot(x, y.T) / np.outer(x2, y2) one fma chain over k (the callers' MFMA /
//                                          fmaf loops), norms multiplied, ONE division
//   d  = arccos(d) / np.pi                 glibc's acosf (fdlibm e_acosf.c), / float32(pi)
//
// Every translation unit that includes this file is compiled with -ffp-contract=off:
// a*b + c below is a rounded product followed by a rounded sum unless it is written
// fmaf().  Division and square root are the compiler's correctly rounded ones (HIP's
// default -fhip-fp32-correctly-rounded-divide-sqrt).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace abn {

__device__ __forceinline__ float bits_f32(uint32_t u) { return __uint_as_float(u); }

// np.sum(v ** 2) over n <= 128 consecutive floats: 8 interleaved partial sums, combined
// as a tree, then the tail (numpy/core/src/umath/loops_utils.h.src, pairwise sum)
__device__ __forceinline__ float sumsq_block(const float* __restrict__ v, int n)
{
    if (n < 8) {
        float res = 0.0f;
        for (int i = 0; i < n; ++i) res += v[i] * v[i];
        return res;
    }
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = v[j] * v[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += v[i + j] * v[i + j];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += v[i] * v[i];
    return res;
}

// the whole recursion (n > 128 splits at n/2 rounded down to a multiple of 8), written
// with an explicit stack: depth <= 16 covers every D the ABI accepts (< 2^20)
__device__ __noinline__ float sumsq_numpy_split(const float* __restrict__ v, int n)
{
    int off[20], len[20], state[20];
    float left[20];
    int sp = 0;
    off[0] = 0; len[0] = n; state[0] = 0;
    float ret = 0.0f;
    while (sp >= 0) {
        if (len[sp] <= 128) {
            ret = sumsq_block(v + off[sp], len[sp]);
            --sp;
            continue;
        }
        int n2 = len[sp] / 2;
        n2 -= n2 % 8;
        if (state[sp] == 0) {              // descend into the left half
            state[sp] = 1;
            off[sp + 1] = off[sp]; len[sp + 1] = n2; state[sp + 1] = 0;
            ++sp;
        } else if (state[sp] == 1) {       // left done: keep it, descend right
            left[sp] = ret;
            state[sp] = 2;
            off[sp + 1] = off[sp] + n2; len[sp + 1] = len[sp] - n2; state[sp + 1] = 0;
            ++sp;
        } else {
            ret = left[sp] + ret;
            --sp;
        }
    }
    return ret;
}

__device__ __forceinline__ float sumsq_numpy(const float* __restrict__ v, int n)
{
    return n <= 128 ? sumsq_block(v, n) : sumsq_numpy_split(v, n);      // (the stack of the split form lives in scratch)
}

__device__ __forceinline__ float row_norm_numpy(const float* __restrict__ v, int n) { return sqrtf(sumsq_numpy(v, n)); }

// a / b, correctly rounded, for operands whose scaled forms equal themselves: the reciprocal
// refinement the compiler emits for `/` (v_rcp_f32, two Newton steps on the quotient) without
// the v_div_scale / v_div_fixup pair around it, i.e. bit-identical to `/` whenever neither
// operand nor the quotient leaves the normal range.  Callers guarantee that range.
__device__ __forceinline__ float div_normal(float a, float b)
{
    float y = __builtin_amdgcn_rcpf(b);
    const float e = fmaf(-b, y, 1.0f);
    y = fmaf(e, y, y);
    float q = a * y;
    float r = fmaf(-b, q, a);
    q = fmaf(r, y, q);
    r = fmaf(-b, q, a);
    return fmaf(r, y, q);
}

// sqrtf, correctly rounded, for x in [2^-96, 2^96] (no rescaling needed): v_sqrt_f32 and the
// compiler's own one-ulp fix-up
__device__ __forceinline__ float sqrt_normal(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sdn = __uint_as_float(__float_as_uint(s) - 1u), sup = __uint_as_float(__float_as_uint(s) + 1u);
    const float rdn = fmaf(-sdn, s, x), rup = fmaf(-sup, s, x);
    float t = (0.0f >= rdn) ? sdn : s;
    t = (0.0f < rup) ? sup : t;
    return t;
}

// a / float32(pi) for a = 0 or a in [2^-13, pi] (every value acosf returns lies there: the smallest
// positive one is acosf(1 - 2^-24) = 3.45e-4): multiply by the rounded reciprocal, one fma
// for the remainder, one to correct.  Equal to the correctly rounded quotient for ALL of
// those floats (exhaustive check on the CPU, tools/divpi_check.c: 0 of 122 683 875 differ).
__device__ __forceinline__ float div_pi(float a)
{
    const float pi_f = bits_f32(0x40490fdbu), inv_pi = bits_f32(0x3ea2f983u);      // float32(np.pi), float32(1 / pi_f)
    const float q = a * inv_pi;
    const float r = fmaf(-q, pi_f, a);
    return fmaf(r, inv_pi, q);
}

// glibc 2.35 __ieee754_acosf, straight-line for the common |x| < 0.5 case with
// wave-uniform detours for the other two ranges (a wave whose lanes all sit in
// |x| < 0.5 -- most of a distance tile of unrelated frames -- never pays for them)
// and for the rare arguments with a constant result (|x| <= 2^-26, |x| >= 1, NaN).
// The divisions' operands are in the normal range by construction: q in [0.6, 1], |p| < 0.05;
// z in [2^-25, 0.25], s + df in [2^-13, 1].
__device__ __forceinline__ float acosf_ref(float x)
{
    const float one = 1.0f;
    const float pi = bits_f32(0x40490fdau), pio2_hi = bits_f32(0x3fc90fdau), pio2_lo = bits_f32(0x33a22168u);
    const float pS0 = bits_f32(0x3e2aaaabu), pS1 = -bits_f32(0x3ea6b090u), pS2 = bits_f32(0x3e4e0aa8u),
                pS3 = -bits_f32(0x3d241146u), pS4 = bits_f32(0x3a4f7f04u), pS5 = bits_f32(0x3811ef08u);
    const float qS1 = -bits_f32(0x4019d139u), qS2 = bits_f32(0x4001572du), qS3 = -bits_f32(0x3f303361u),
                qS4 = bits_f32(0x3d9dc62eu);
    const uint32_t hx = __float_as_uint(x), ix = hx & 0x7fffffffu;
    const float ax = __uint_as_float(ix);
    const bool small = ix < 0x3f000000u;
    // x < -0.5: z = (1 + x) / 2 = (1 - |x|) / 2, the same operation
    const float z = small ? x * x : (one - ax) * 0.5f;
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = div_normal(p, q);
    float res = pio2_hi - (x - (pio2_lo - x * r));
    if (__any(!small)) {
        const float s = sqrt_normal(z);
        const float wn = r * s - pio2_lo;
        const float neg = pi - 2.0f * (s + wn);
        float big = neg;
        const bool posb = !small && (int32_t)hx >= 0;
        if (__any(posb)) {
            const float df = __uint_as_float(__float_as_uint(s) & 0xfffff000u);
            const float c = div_normal(z - df * df, s + df);
            const float wp = r * s + c;
            big = posb ? 2.0f * (df + wp) : neg;
        }
        res = small ? res : big;
    }
    if (__any(ix <= 0x32800000u || ix >= 0x3f800000u)) {
        res = ix <= 0x32800000u ? pio2_hi + pio2_lo : res;
        res = ix == 0x3f800000u ? ((int32_t)hx > 0 ? 0.0f : pi + 2.0f * pio2_lo) : res;
        res = ix > 0x3f800000u ? __builtin_nanf("") : res;    // |x| > 1 (rounding) or NaN: the pair is dropped
    }
    return res;
}

// one cell of cosine_distance: dot = the fma chain x.y, nx / ny = the row norms
// (utils.py:46-58; a zero-norm row is at distance 1 from everything, 0 from another zero row).
// PLAIN = true: the caller has checked that every norm of the block lies in [2^-40, 2^40] (so
// no zero rows, and dot / (nx ny) stays in the normal range or rounds to a value acosf maps to
// pi/2 anyway); otherwise the compiler's full IEEE division and the zero-row rules.
template <bool PLAIN>
__device__ __forceinline__ float angular_distance_ref(float dot, float nx, float ny)
{
    if (PLAIN) return div_pi(acosf_ref(div_normal(dot, nx * ny)));
    const float pi_f = bits_f32(0x40490fdbu);                  // float32(np.pi)
    float v = acosf_ref(dot / (nx * ny)) / pi_f;
    const bool zx = nx == 0.0f, zy = ny == 0.0f;
    v = (zx || zy) ? 1.0f : v;
    v = (zx && zy) ? 0.0f : v;
    return v;
}

// ---------------------------------------------------------------------------------------
// The same cell function on TWO cells at a time (the plain path only).  The distance epilogue is
// ~95 VALU instructions per cell and the DTW kernel is bound by them; ~70 of those are fp32
// multiplies / adds / fmas, which gfx950 issues two per lane as v_pk_mul_f32 / v_pk_add_f32 /
// v_pk_fma_f32 -- element by element the same IEEE operations in the same order, so every
// result is bit-identical to the scalar functions above (tests/test_gpu_dtw_features.py compare
// both against the reference's own output).
// ---------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef int32_t i32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ u32x2 bits2(f32x2 v) { return __builtin_bit_cast(u32x2, v); }
__device__ __forceinline__ f32x2 float2_of(u32x2 v) { return __builtin_bit_cast(f32x2, v); }

__device__ __forceinline__ f32x2 div_normal2(f32x2 a, f32x2 b)
{
    f32x2 y = f32x2{__builtin_amdgcn_rcpf(b.x), __builtin_amdgcn_rcpf(b.y)};
    const f32x2 e = fma2(-b, y, splat2(1.0f));
    y = fma2(e, y, y);
    f32x2 q = a * y;
    f32x2 r = fma2(-b, q, a);
    q = fma2(r, y, q);
    r = fma2(-b, q, a);
    return fma2(r, y, q);
}

__device__ __forceinline__ f32x2 sqrt_normal2(f32x2 x)
{
    const f32x2 s = f32x2{__builtin_amdgcn_sqrtf(x.x), __builtin_amdgcn_sqrtf(x.y)};
    const f32x2 sdn = float2_of(bits2(s) - 1u), sup = float2_of(bits2(s) + 1u);
    const f32x2 rdn = fma2(-sdn, s, x), rup = fma2(-sup, s, x);
    f32x2 t = (splat2(0.0f) >= rdn) ? sdn : s;
    t = (splat2(0.0f) < rup) ? sup : t;
    return t;
}

__device__ __forceinline__ f32x2 div_pi2(f32x2 a)
{
    const f32x2 pi_f = splat2(bits_f32(0x40490fdbu)), inv_pi = splat2(bits_f32(0x3ea2f983u));
    const f32x2 q = a * inv_pi;
    const f32x2 r = fma2(-q, pi_f, a);
    return fma2(r, inv_pi, q);
}

__device__ __forceinline__ f32x2 acosf_ref2(f32x2 x)
{
    const f32x2 one = splat2(1.0f);
    const f32x2 pi = splat2(bits_f32(0x40490fdau)), pio2_hi = splat2(bits_f32(0x3fc90fdau)), pio2_lo = splat2(bits_f32(0x33a22168u));
    const f32x2 pS0 = splat2(bits_f32(0x3e2aaaabu)), pS1 = splat2(-bits_f32(0x3ea6b090u)), pS2 = splat2(bits_f32(0x3e4e0aa8u)),
                pS3 = splat2(-bits_f32(0x3d241146u)), pS4 = splat2(bits_f32(0x3a4f7f04u)), pS5 = splat2(bits_f32(0x3811ef08u));
    const f32x2 qS1 = splat2(-bits_f32(0x4019d139u)), qS2 = splat2(bits_f32(0x4001572du)), qS3 = splat2(-bits_f32(0x3f303361u)),
                qS4 = splat2(bits_f32(0x3d9dc62eu));
    const u32x2 hx = bits2(x), ix = hx & 0x7fffffffu;
    const f32x2 ax = float2_of(ix);
    const i32x2 small = ix < 0x3f000000u;
    const f32x2 z = small ? x * x : (one - ax) * splat2(0.5f);
    const f32x2 p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const f32x2 q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const f32x2 r = div_normal2(p, q);
    f32x2 res = pio2_hi - (x - (pio2_lo - x * r));
    if (__any(!(small.x && small.y))) {
        const f32x2 s = sqrt_normal2(z);
        const f32x2 wn = r * s - pio2_lo;
        const f32x2 neg = pi - splat2(2.0f) * (s + wn);
        f32x2 big = neg;
        const i32x2 posb = (~small) & (__builtin_bit_cast(i32x2, hx) >= 0);
        if (__any(posb.x || posb.y)) {
            const f32x2 df = float2_of(bits2(s) & 0xfffff000u);
            const f32x2 c = div_normal2(z - df * df, s + df);
            const f32x2 wp = r * s + c;
            big = posb ? splat2(2.0f) * (df + wp) : neg;
        }
        res = small ? res : big;
    }
    const i32x2 tiny = ix <= 0x32800000u, ge1 = ix >= 0x3f800000u;
    if (__any(tiny.x || tiny.y || ge1.x || ge1.y)) {
        res = tiny ? pio2_hi + pio2_lo : res;
        const f32x2 at1 = (__builtin_bit_cast(i32x2, hx) > 0) ? splat2(0.0f) : pi + splat2(2.0f) * pio2_lo;
        res = (ix == 0x3f800000u) ? at1 : res;
        res = (ix > 0x3f800000u) ? splat2(__builtin_nanf("")) : res;
    }
    return res;
}

// two cells of one row of x (norm nx) against two rows of y: the PLAIN path of angular_distance_ref
__device__ __forceinline__ f32x2 angular_distance_plain2(f32x2 dot, float nx, f32x2 ny)
{
    return div_pi2(acosf_ref2(div_normal2(dot, splat2(nx) * ny)));
}

// acosf_ref2 for arguments the CALLER has checked to lie in 2^-26 < |x| < 0.5 in every lane of the
// wavefront: the statements of the first range alone (z = x x; p / q; pi/2 - (x - (lo - x r))), no
// selects, no compares.  The same operations in the same order as acosf_ref takes for such an
// argument, so the same bits.
__device__ __forceinline__ f32x2 acosf_small2(f32x2 x)
{
    const f32x2 one = splat2(1.0f);
    const f32x2 pio2_hi = splat2(bits_f32(0x3fc90fdau)), pio2_lo = splat2(bits_f32(0x33a22168u));
    const f32x2 pS0 = splat2(bits_f32(0x3e2aaaabu)), pS1 = splat2(-bits_f32(0x3ea6b090u)), pS2 = splat2(bits_f32(0x3e4e0aa8u)),
                pS3 = splat2(-bits_f32(0x3d241146u)), pS4 = splat2(bits_f32(0x3a4f7f04u)), pS5 = splat2(bits_f32(0x3811ef08u));
    const f32x2 qS1 = splat2(-bits_f32(0x4019d139u)), qS2 = splat2(bits_f32(0x4001572du)), qS3 = splat2(-bits_f32(0x3f303361u)),
                qS4 = splat2(bits_f32(0x3d9dc62eu));
    const f32x2 z = x * x;
    const f32x2 p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const f32x2 q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const f32x2 r = div_normal2(p, q);
    return pio2_hi - (x - (pio2_lo - x * r));
}

// does every one of the four quotients of EVERY lane lie in acosf's first range, away from its constant-result
// arguments?  (2^-26 < |c| < 0.5; a NaN among them fails one of the two compares unless another value
// hides it in the maximum -- then the straight-line statements carry it through to a NaN distance, which
// is all the callers ask of it: the pair is dropped)
__device__ __forceinline__ bool quotients_small4_all(f32x2 a, f32x2 b)      // wave-uniform; every lane of the wavefront is here
{
    const float mx = fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fabsf(b.x)), fabsf(b.y));
    const float mn = fminf(fminf(fminf(fabsf(a.x), fabsf(a.y)), fabsf(b.x)), fabsf(b.y));
    // (two ballots: each compare lands in a scalar register pair, one scalar AND, no vector select in between)
    return (__builtin_amdgcn_ballot_w64(mx < 0.5f) & __builtin_amdgcn_ballot_w64(mn > bits_f32(0x32800000u))) == ~0ull;
}

// The straight-line statements on FOUR cells at a time: each operation becomes two independent packed
// instructions side by side, so a dependent operation never directly follows the one it waits for.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 splat4(float v) { return f32x4{v, v, v, v}; }
__device__ __forceinline__ f32x4 fma4(f32x4 a, f32x4 b, f32x4 c) { return __builtin_elementwise_fma(a, b, c); }

__device__ __forceinline__ f32x4 div_normal4(f32x4 a, f32x4 b)
{
    f32x4 y = f32x4{__builtin_amdgcn_rcpf(b.x), __builtin_amdgcn_rcpf(b.y), __builtin_amdgcn_rcpf(b.z), __builtin_amdgcn_rcpf(b.w)};
    const f32x4 e = fma4(-b, y, splat4(1.0f));
    y = fma4(e, y, y);
    f32x4 q = a * y;
    f32x4 r = fma4(-b, q, a);
    q = fma4(r, y, q);
    r = fma4(-b, q, a);
    return fma4(r, y, q);
}

__device__ __forceinline__ f32x4 div_pi4(f32x4 a)
{
    const f32x4 pi_f = splat4(bits_f32(0x40490fdbu)), inv_pi = splat4(bits_f32(0x3ea2f983u));
    const f32x4 q = a * inv_pi;
    const f32x4 r = fma4(-q, pi_f, a);
    return fma4(r, inv_pi, q);
}

__device__ __forceinline__ f32x4 acosf_small4(f32x4 x)        // acosf_small2's statements (2^-26 < |x| < 0.5 in every lane)
{
    const f32x4 one = splat4(1.0f);
    const f32x4 pio2_hi = splat4(bits_f32(0x3fc90fdau)), pio2_lo = splat4(bits_f32(0x33a22168u));
    const f32x4 pS0 = splat4(bits_f32(0x3e2aaaabu)), pS1 = splat4(-bits_f32(0x3ea6b090u)), pS2 = splat4(bits_f32(0x3e4e0aa8u)),
                pS3 = splat4(-bits_f32(0x3d241146u)), pS4 = splat4(bits_f32(0x3a4f7f04u)), pS5 = splat4(bits_f32(0x3811ef08u));
    const f32x4 qS1 = splat4(-bits_f32(0x4019d139u)), qS2 = splat4(bits_f32(0x4001572du)), qS3 = splat4(-bits_f32(0x3f303361u)),
                qS4 = splat4(bits_f32(0x3d9dc62eu));
    const f32x4 z = x * x;
    const f32x4 p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const f32x4 q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const f32x4 r = div_normal4(p, q);
    return pio2_hi - (x - (pio2_lo - x * r));
}

// is the norm inside the range angular_distance_ref<true> is valid for?
__device__ __forceinline__ bool norm_is_plain(float v) { return v >= 9.094947e-13f && v <= 1.0995116e12f; }

}  // namespace abn
