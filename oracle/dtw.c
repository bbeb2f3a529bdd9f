/* ORACLE (test infrastructure, never shipped as a product path).
 *
 * Plain-C restatement of the DTW frame-alignment path:
 *   abnet3/utils.py:40-60    cosine_distance (angular distance arccos(cos)/pi)
 *   abnet3/utils.py:147-153  get_dtw_alignment -> dtw.DTW(..., dist_array=D,
 *                            return_alignment=True)
 *
 * cosine_distance restates the reference's numpy statements operation by
 * operation, in float32 as numpy evaluates them for float32 inputs:
 *   x2 = np.sqrt(np.sum(x ** 2, axis=1))     squares rounded to float32, then numpy's
 *                                            pairwise summation (8 interleaved partial
 *                                            sums, blocks of <= 128; numpy
 *                                            core/src/umath/loops_utils.h), then sqrtf
 *   np.dot(x, y.T)                           sgemm: ONE fused-multiply-add chain over k
 *                                            per cell -- what OpenBLAS's register-tiled
 *                                            kernels compute on FMA hardware (see below)
 *   / np.outer(x2, y2)                       float32 product of the norms, then ONE
 *                                            float32 division
 *   scipy.arccos(d)                          = np.arccos on float32 = libm acosf where
 *                                            numpy has no SIMD override; restated here
 *                                            from glibc 2.35's e_acosf.c (fdlibm), see
 *                                            abn_oracle_acosf
 *   / np.pi                                  float32 division by float32(pi)
 *   np.float64(...)                          widened, exact
 * PINNED by tests/golden/cosdist_libm.npz: outputs of the reference's own
 * cosine_distance, generated in the build container with numpy's AVX512 dispatch
 * disabled (NPY_DISABLE_CPU_FEATURES), on inputs large enough for OpenBLAS's regular
 * sgemm kernel -- this file reproduces them BIT FOR BIT, including which pairs trip
 * the reference's `assert np.all(d >= 0)` (NaN from arccos(>1)) and are dropped by the
 * caller (abnet3/dataloader.py:188-191).
 * Two parts of the reference's arithmetic depend on the machine it runs on and cannot
 * be restated portably; the oracle takes the plain-libm / regular-kernel branch:
 *   - with AVX512_SKX numpy evaluates float32 arccos with Intel SVML (<= 4 ulp, differs
 *     from libm in ~1/3 of the arguments by 1 ulp);
 *   - for M*N*K <= 10^6 OpenBLAS's SkylakeX build takes a "small matrix" sgemm kernel
 *     that sums k in vector-lane order.
 * tests/golden/cosdist.npz (default dispatch on an AVX512 host) is therefore compared
 * with a tolerance (2e-7 in d), cosdist_libm.npz exactly.
 *
 * PARITY UNPINNED for the DP itself: dtw.DTW lives in the un-vendored
 * third-party package Rachine/DTW_Cython (requirements.txt:9, no version pin),
 * absent from /root/reference, and no reference test holds a known answer for
 * it.  This file restates the classic unit-step DTW its call site implies
 *  and IS the definition the HIP kernel must match
 * bit-for-bit:
 *   cost[0][0] = D[0][0]; first row / column are running sums;
 *   cost[i][j] = D[i][j] + min(cost[i-1][j-1], cost[i-1][j], cost[i][j-1]);
 *   traceback from (N-1, M-1): first minimum in the order
 *   diagonal, up (i-1), left (j-1) wins a tie; path returned start -> end.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off: no operation below may be fused
 * unless it is written as fmaf)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* acosf as glibc 2.35 evaluates it (sysdeps/ieee754/flt-32/e_acosf.c, the fdlibm
 * float routine; constants and operation order read off the libm.so.6 of this image,
 * whose code uses separate mulss / addss, no fma).  tools/acosf_exhaustive.py compares
 * this function with libm's acosf for EVERY float32 in [-1, 1] and a band outside. */
float abn_oracle_acosf(float x)
{
    const float one = 1.0f;
    const float pi = u2f(0x40490fdau);          /* 3.1415925026e+00 */
    const float pio2_hi = u2f(0x3fc90fdau);     /* 1.5707962513e+00 */
    const float pio2_lo = u2f(0x33a22168u);     /* 7.5497894159e-08 */
    const float pS0 = u2f(0x3e2aaaabu);         /*  1.6666667163e-01 */
    const float pS1 = -u2f(0x3ea6b090u);        /* -3.2556581497e-01 */
    const float pS2 = u2f(0x3e4e0aa8u);         /*  2.0121252537e-01 */
    const float pS3 = -u2f(0x3d241146u);        /* -4.0055535734e-02 */
    const float pS4 = u2f(0x3a4f7f04u);         /*  7.9153501429e-04 */
    const float pS5 = u2f(0x3811ef08u);         /*  3.4793309169e-05 */
    const float qS1 = -u2f(0x4019d139u);        /* -2.4033949375e+00 */
    const float qS2 = u2f(0x4001572du);         /*  2.0209457874e+00 */
    const float qS3 = -u2f(0x3f303361u);        /* -6.8828397989e-01 */
    const float qS4 = u2f(0x3d9dc62eu);         /*  7.7038154006e-02 */
    const uint32_t hx = f2u(x), ix = hx & 0x7fffffffu;
    float z, p, q, r, w, s, c, df;
    if (ix == 0x3f800000u) {                    /* |x| == 1 */
        if ((int32_t)hx > 0) return 0.0f;
        return pi + 2.0f * pio2_lo;
    }
    if (ix > 0x3f800000u) return (x - x) / (x - x);   /* |x| > 1 or NaN: NaN */
    if (ix < 0x3f000000u) {                     /* |x| < 0.5 */
        if (ix <= 0x32800000u) return pio2_hi + pio2_lo;
        z = x * x;
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if ((int32_t)hx < 0) {                      /* x < -0.5 */
        z = (one + x) * 0.5f;
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        s = sqrtf(z);
        r = p / q;
        w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    }
    z = (one - x) * 0.5f;                       /* x > 0.5 */
    s = sqrtf(z);
    df = u2f(f2u(s) & 0xfffff000u);
    c = (z - df * df) / (s + df);
    p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    r = p / q;
    w = r * s + c;
    return 2.0f * (df + w);
}

/* number of float32 bit patterns in [lo_bits, hi_bits] for which the restatement above
 * and this machine's libm acosf differ (NaN == NaN); first mismatch in *first. */
int64_t abn_oracle_acosf_vs_libm(uint32_t lo_bits, uint32_t hi_bits, uint32_t step, uint32_t* first)
{
    int64_t bad = 0;
    for (uint64_t b = lo_bits; b <= hi_bits; b += step) {
        const float x = u2f((uint32_t)b);
        const uint32_t a = f2u(abn_oracle_acosf(x)), l = f2u(acosf(x));
        const int nan_a = (a & 0x7fffffffu) > 0x7f800000u, nan_l = (l & 0x7fffffffu) > 0x7f800000u;
        if ((nan_a || nan_l) ? (nan_a != nan_l) : (a != l)) {
            if (!bad && first) *first = (uint32_t)b;
            ++bad;
        }
    }
    return bad;
}

void abn_oracle_acosf_array(const float* x, int64_t n, int over_pi, float* out)
{
    const float pi_f = u2f(0x40490fdbu);
    for (int64_t i = 0; i < n; ++i) out[i] = over_pi ? abn_oracle_acosf(x[i]) / pi_f : abn_oracle_acosf(x[i]);
}

/* np.sum(v ** 2) of a contiguous float32 row: the squares are rounded to float32
 * (x ** 2 is a separate array), then numpy's pairwise summation. */
static float pairwise_sum_sq(const float* v, int64_t n)
{
    if (n < 8) {
        float res = 0.0f;
        for (int64_t i = 0; i < n; ++i) res += v[i] * v[i];
        return res;
    }
    if (n <= 128) {
        float r[8], res;
        int64_t i;
        for (int j = 0; j < 8; ++j) r[j] = v[j] * v[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += v[i + j] * v[i + j];
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += v[i] * v[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_sum_sq(v, n2) + pairwise_sum_sq(v + n2, n - n2);
}

float abn_oracle_row_norm(const float* v, int64_t D) { return sqrtf(pairwise_sum_sq(v, D)); }

/* utils.py:40-60 for float32 inputs. d is N x M row-major float64.
 * Returns 0, or 1 if any entry is NaN / negative (the reference's
 * `assert np.all(d >= 0)` then raises and the caller drops the pair,
 * dataloader.py:188-191). */
int abn_oracle_cosine_distance_f32(const float* x, int64_t N, const float* y,
                                   int64_t M, int64_t D, double* d)
{
    const float pi_f = u2f(0x40490fdbu);              /* float32(np.pi) */
    float* nx = (float*)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
    float* ny = (float*)malloc(sizeof(float) * (size_t)(M > 0 ? M : 1));
    int bad = 0;
    for (int64_t i = 0; i < N; ++i) nx[i] = abn_oracle_row_norm(x + i * D, D);
    for (int64_t j = 0; j < M; ++j) ny[j] = abn_oracle_row_norm(y + j * D, D);
    for (int64_t i = 0; i < N; ++i) {
        for (int64_t j = 0; j < M; ++j) {
            float v;
            if (nx[i] == 0.0f && ny[j] == 0.0f) v = 0.0f;        /* :57-58 */
            else if (nx[i] == 0.0f || ny[j] == 0.0f) v = 1.0f;    /* :55-56 */
            else {
                float dot = 0.0f;
                for (int64_t k = 0; k < D; ++k)
                    dot = fmaf(x[i * D + k], y[j * D + k], dot);
                const float c = dot / (nx[i] * ny[j]);
                v = abn_oracle_acosf(c) / pi_f;
            }
            if (!(v >= 0.0f)) bad = 1;
            d[i * M + j] = (double)v;
        }
    }
    free(nx);
    free(ny);
    return bad;
}

/* utils.py:40-60 for float64 inputs (the reference computes in the input precision):
 * the same statements in double; dgemm as a sequential fma chain, libm acos.  Pinned by
 * tests/golden/cosdist.npz `f64` to a few ulp (numpy's double arccos is libm's). */
int abn_oracle_cosine_distance_f64(const double* x, int64_t N, const double* y,
                                   int64_t M, int64_t D, double* d)
{
    int bad = 0;
    for (int64_t i = 0; i < N; ++i) {
        double sx = 0.0;
        for (int64_t k = 0; k < D; ++k) sx += x[i * D + k] * x[i * D + k];
        const double nx = sqrt(sx);
        for (int64_t j = 0; j < M; ++j) {
            double sy = 0.0, dot = 0.0, v;
            for (int64_t k = 0; k < D; ++k) {
                sy += y[j * D + k] * y[j * D + k];
                dot = fma(x[i * D + k], y[j * D + k], dot);
            }
            const double ny = sqrt(sy);
            if (nx == 0.0 && ny == 0.0) v = 0.0;
            else if (nx == 0.0 || ny == 0.0) v = 1.0;
            else v = acos(dot / (nx * ny)) / 3.14159265358979323846;
            if (!(v >= 0.0)) bad = 1;
            d[i * M + j] = v;
        }
    }
    return bad;
}

/* DTW over a precomputed N x M float64 distance matrix (row-major).
 * path1/path2 must hold N+M-1 entries. Returns the path length (>= max(N,M)),
 * or 0 for an empty problem.  *total gets cost[N-1][M-1]. */
int64_t abn_oracle_dtw(const double* d, int64_t N, int64_t M, int32_t* path1,
                       int32_t* path2, double* total)
{
    if (N <= 0 || M <= 0) return 0;
    double* cost = (double*)malloc(sizeof(double) * (size_t)(N * M));
    cost[0] = d[0];
    for (int64_t j = 1; j < M; ++j) cost[j] = d[j] + cost[j - 1];
    for (int64_t i = 1; i < N; ++i) cost[i * M] = d[i * M] + cost[(i - 1) * M];
    for (int64_t i = 1; i < N; ++i) {
        for (int64_t j = 1; j < M; ++j) {
            double best = cost[(i - 1) * M + (j - 1)];
            double up = cost[(i - 1) * M + j];
            double left = cost[i * M + (j - 1)];
            if (up < best) best = up;
            if (left < best) best = left;
            cost[i * M + j] = d[i * M + j] + best;
        }
    }
    if (total) *total = cost[N * M - 1];
    int64_t i = N - 1, j = M - 1, len = 0;
    path1[len] = (int32_t)i;
    path2[len] = (int32_t)j;
    ++len;
    while (i > 0 || j > 0) {
        if (i == 0) --j;
        else if (j == 0) --i;
        else {
            double diag = cost[(i - 1) * M + (j - 1)];
            double up = cost[(i - 1) * M + j];
            double left = cost[i * M + (j - 1)];
            int dir = 0;
            double best = diag;
            if (up < best) { best = up; dir = 1; }
            if (left < best) { best = left; dir = 2; }
            if (dir == 0) { --i; --j; }
            else if (dir == 1) --i;
            else --j;
        }
        path1[len] = (int32_t)i;
        path2[len] = (int32_t)j;
        ++len;
    }
    for (int64_t a = 0, b = len - 1; a < b; ++a, --b) {
        int32_t t = path1[a]; path1[a] = path1[b]; path1[b] = t;
        t = path2[a]; path2[a] = path2[b]; path2[b] = t;
    }
    free(cost);
    return len;
}

/* Whole batch: features -> distance -> DTW, pair p uses rows
 * [off1[p], off1[p]+n1[p]) of feats1 and likewise of feats2.  Paths are
 * written at pair-major stride `path_stride`; path_len[p]=0 marks a dropped
 * pair (NaN distance).  Used as the bench's CPU baseline ("port").
 * abn_oracle_dtw_batch_mt: the same over `threads` OpenMP threads -- the pairs are
 * independent (abnet3/dataloader.py:183-191 aligns them one by one), every pair is
 * computed by exactly one thread with the single-thread arithmetic: identical
 * results whatever the thread count (BASELINE.md section 3's "all cores" figure). */
static int64_t dtw_one_pair(const float* feats1, const int64_t* off1, const int32_t* n1, const float* feats2,
                            const int64_t* off2, const int32_t* n2, int64_t p, int64_t D, int32_t* path1,
                            int32_t* path2, int32_t* path_len, int64_t path_stride)
{
    int64_t N = n1[p], M = n2[p];
    double* d = (double*)malloc(sizeof(double) * (size_t)(N * M > 0 ? N * M : 1));
    int bad = abn_oracle_cosine_distance_f32(feats1 + off1[p] * D, N,
                                             feats2 + off2[p] * D, M, D, d);
    if (bad || N * M == 0) path_len[p] = 0;
    else
        path_len[p] = (int32_t)abn_oracle_dtw(
            d, N, M, path1 + p * path_stride, path2 + p * path_stride, 0);
    free(d);
    return N * M;
}

int64_t abn_oracle_dtw_batch(const float* feats1, const int64_t* off1,
                             const int32_t* n1, const float* feats2,
                             const int64_t* off2, const int32_t* n2,
                             int64_t npairs, int64_t D, int32_t* path1,
                             int32_t* path2, int32_t* path_len,
                             int64_t path_stride)
{
    int64_t cells = 0;
    for (int64_t p = 0; p < npairs; ++p)
        cells += dtw_one_pair(feats1, off1, n1, feats2, off2, n2, p, D, path1, path2, path_len, path_stride);
    return cells;
}

int64_t abn_oracle_dtw_batch_mt(const float* feats1, const int64_t* off1,
                                const int32_t* n1, const float* feats2,
                                const int64_t* off2, const int32_t* n2,
                                int64_t npairs, int64_t D, int32_t* path1,
                                int32_t* path2, int32_t* path_len,
                                int64_t path_stride, int threads)
{
    int64_t cells = 0;
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 8) num_threads(threads) reduction(+ : cells)
    for (int64_t p = 0; p < npairs; ++p)
        cells += dtw_one_pair(feats1, off1, n1, feats2, off2, n2, p, D, path1, path2, path_len, path_stride);
    return cells;
}
