/* ORACLE (test infrastructure, never shipped as a product path).
 *
 * Plain-C restatement of the DTW frame-alignment path:
 *   abnet3/utils.py:40-60    cosine_distance (angular distance arccos(cos)/pi)
 *   abnet3/utils.py:147-153  get_dtw_alignment -> dtw.DTW(..., dist_array=D,
 *                            return_alignment=True)
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
 * cosine_distance IS pinned (tests/golden/cosdist.npz, from the reference) to
 * a tolerance; its float32 arithmetic is fixed here operation by operation
 * (sequential fmaf dot products, reciprocal norms, a division-free acos) so
 * that CPU and GPU agree bitwise.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* acos in pure float32 arithmetic, division-free: Abramowitz & Stegun 4.4.46,
 *   acos(x) = sqrt(1 - x) * (a0 + a1 x + ... + a7 x^7),  0 <= x <= 1, |err| <= 2e-8,
 * and acos(-x) = pi - acos(x).  Every operation is one IEEE-754 binary32 fma /
 * add / mul / sqrt in a fixed order (Horner with fmaf), so the GPU kernel
 * reproduces it bit for bit.  (An earlier revision used fdlibm's rational form;
 * its three divisions per cell dominated the GPU distance kernel.) */
float abn_oracle_acosf(float x)
{
    const float pi_f = 3.14159274101257324f;
    const float ax = fabsf(x);
    if (!(ax <= 1.0f)) return NAN;       /* |x| > 1 or NaN: utils.py:59 then trips its assert */
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

static float row_norm(const float* v, int64_t D)
{
    float s = 0.0f;
    for (int64_t k = 0; k < D; ++k) s = fmaf(v[k], v[k], s);
    return sqrtf(s);
}

/* utils.py:40-60 for float32 inputs. d is N x M row-major float64.
 *   cos = (dot * (1/|x|)) * (1/|y|)   (reciprocal norms once per row; the
 *                                      reference divides by the outer product),
 *   d   = acos(cos) * float32(1/pi)   (the reference divides by pi).
 * Returns 0, or 1 if any entry is NaN / negative (the reference's
 * `assert np.all(d >= 0)` then raises and the caller drops the pair,
 * dataloader.py:188-191). */
int abn_oracle_cosine_distance_f32(const float* x, int64_t N, const float* y,
                                   int64_t M, int64_t D, double* d)
{
    const float inv_pi_f = 0.318309873342514038f;     /* float32(1/pi) */
    float* nx = (float*)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
    float* ny = (float*)malloc(sizeof(float) * (size_t)(M > 0 ? M : 1));
    int bad = 0;
    for (int64_t i = 0; i < N; ++i) nx[i] = row_norm(x + i * D, D);
    for (int64_t j = 0; j < M; ++j) ny[j] = row_norm(y + j * D, D);
    for (int64_t i = 0; i < N; ++i) {
        for (int64_t j = 0; j < M; ++j) {
            float v;
            if (nx[i] == 0.0f && ny[j] == 0.0f) v = 0.0f;        /* :57-58 */
            else if (nx[i] == 0.0f || ny[j] == 0.0f) v = 1.0f;    /* :55-56 */
            else {
                float dot = 0.0f;
                for (int64_t k = 0; k < D; ++k)
                    dot = fmaf(x[i * D + k], y[j * D + k], dot);
                const float c = (dot * (1.0f / nx[i])) * (1.0f / ny[j]);
                v = abn_oracle_acosf(c) * inv_pi_f;
            }
            if (!(v >= 0.0f)) bad = 1;
            d[i * M + j] = (double)v;
        }
    }
    free(nx);
    free(ny);
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
 * pair (NaN distance).  Used as the bench's CPU baseline ("port"). */
int64_t abn_oracle_dtw_batch(const float* feats1, const int64_t* off1,
                             const int32_t* n1, const float* feats2,
                             const int64_t* off2, const int32_t* n2,
                             int64_t npairs, int64_t D, int32_t* path1,
                             int32_t* path2, int32_t* path_len,
                             int64_t path_stride)
{
    int64_t cells = 0;
    for (int64_t p = 0; p < npairs; ++p) {
        int64_t N = n1[p], M = n2[p];
        double* d = (double*)malloc(sizeof(double) * (size_t)(N * M > 0 ? N * M : 1));
        int bad = abn_oracle_cosine_distance_f32(feats1 + off1[p] * D, N,
                                                 feats2 + off2[p] * D, M, D, d);
        if (bad || N * M == 0) path_len[p] = 0;
        else
            path_len[p] = (int32_t)abn_oracle_dtw(
                d, N, M, path1 + p * path_stride, path2 + p * path_stride, 0);
        cells += N * M;
        free(d);
    }
    return cells;
}
