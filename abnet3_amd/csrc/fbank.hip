// fbank.hip -- log mel filterbank extraction for gfx950.
//
// Replaces FeaturesGenerator.do_fbank, abnet3/features.py:99-114, i.e.
//   Spectral(nfilt=40, alpha=0.97, do_dct=False, fs, frate=100, wlen=0.025,
//            nfft=1024, ...).transform(sound)  -> float32 [T, 40]
// The arithmetic of the third-party `spectral` package is not in the
// reference; the definition implemented here is oracle/features_np.py (parity
// unpinned, see its header): per frame pre-emphasis -> Hamming window ->
// |rfft(nfft)|^2 -> triangular mel bank -> log(max(., 1e-5)).  Framing follows the
// Sphinx-III mfcc.py lineage of that package, quirks included: the pre-emphasis history of a
// frame's first sample is the LAST sample of the previous frame (0 for frame 0), and a frame
// cut short by the end of the signal is filled by repeating its samples cyclically
// (np.resize), see FrameSrc.
//
// One workgroup per frame: the frame is loaded straight into bit-reversed
// order in LDS, a radix-2 FFT runs in place (8 KB for nfft=1024), and the mel
// projection reads the 82 KB weight table, which stays L2-resident across
// frames.  Algorithmic HBM traffic is 2 B per new sample in and 160 B per frame
// out; the kernel is LDS/VALU-bound by the FFT.
#include "common.h"

namespace abn {

constexpr int FB_MAX_NFFT = 2048;
constexpr int FB_MAX_FILT = 128;
constexpr float FB_FLOOR = 1e-5f;

__device__ __forceinline__ unsigned bitrev(unsigned v, int bits) { return __brev(v) >> (32 - bits); }

// Where a frame's samples come from (oracle/features_np.py, `frame_samples`): frame `fr` of an
// utterance of `len` samples starts at round(fr * fshift) and holds wlen samples; with fewer than
// wlen left it repeats what is there cyclically (np.resize; an empty frame is zeros).  The
// pre-emphasis history of element 0 is element wlen - 1 of the PREVIOUS frame (0 for frame 0).
struct FrameSrc {
    const void* base;          // the utterance's first sample
    int is_i16;
    int64_t start;             // first sample of the frame
    int avail;                 // samples between start and the end of the utterance, capped at wlen
    int wlen;
    __device__ __forceinline__ float raw(int64_t i) const
    {
        return is_i16 ? (float)((const int16_t*)base)[i] : ((const float*)base)[i];
    }
    // element i of the frame, 0 <= i < wlen
    __device__ __forceinline__ float at(int i) const
    {
        if (avail >= wlen) return raw(start + i);
        if (avail <= 0) return 0.0f;
        return raw(start + i % avail);
    }
};
__device__ __forceinline__ FrameSrc frame_src(const void* base, int is_i16, int64_t len, int64_t fr, double fshift, int wlen)
{
    FrameSrc f;
    f.base = base; f.is_i16 = is_i16; f.wlen = wlen;
    f.start = (int64_t)rint((double)fr * fshift);
    const int64_t left = len - f.start;
    f.avail = left >= wlen ? wlen : (left > 0 ? (int)left : 0);
    return f;
}
// utterance of global frame `frame` in a batch (utt_foff: cumulative frame counts, [n_utts + 1])
__device__ __forceinline__ int find_utt(const int64_t* __restrict__ utt_foff, int n_utts, int64_t frame)
{
    int lo = 0, hi = n_utts - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (utt_foff[mid] <= frame) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(256) void fbank_kernel(const void* __restrict__ samples, int is_i16, int64_t nsamples,
                                                    int wlen, double fshift, int nfft, int log2n, int nfilt,
                                                    float alpha, const float* __restrict__ window,
                                                    const float* __restrict__ melbank, float* __restrict__ out,
                                                    const int64_t* __restrict__ utt_soff, const int64_t* __restrict__ utt_foff,
                                                    int n_utts)
{
    __shared__ float re[FB_MAX_NFFT], im[FB_MAX_NFFT];
    __shared__ float twr[FB_MAX_NFFT / 2], twi[FB_MAX_NFFT / 2];
    __shared__ float part[256];
    const int tid = threadIdx.x;
    const int64_t frame = blockIdx.x;
    int64_t fr = frame, len = nsamples;
    const char* base = (const char*)samples;
    if (n_utts > 0) {
        const int u = find_utt(utt_foff, n_utts, frame);
        fr = frame - utt_foff[u];
        len = utt_soff[u + 1] - utt_soff[u];
        base += utt_soff[u] * (is_i16 ? 2 : 4);
    }
    const FrameSrc cur = frame_src(base, is_i16, len, fr, fshift, wlen);
    const FrameSrc prv = frame_src(base, is_i16, len, fr > 0 ? fr - 1 : 0, fshift, wlen);
    const float prior = fr > 0 ? prv.at(wlen - 1) : 0.0f;
    for (int n = tid; n < nfft; n += 256) {
        float v = 0.0f;
        if (n < wlen) v = (cur.at(n) - alpha * (n > 0 ? cur.at(n - 1) : prior)) * window[n];
        const unsigned r = bitrev((unsigned)n, log2n);
        re[r] = v;
        im[r] = 0.0f;
    }
    for (int k = tid; k < nfft / 2; k += 256) {
        float s, c;
        sincospif(-2.0f * (float)k / (float)nfft, &s, &c);
        twr[k] = c;
        twi[k] = s;
    }
    __syncthreads();
    for (int s = 1; s <= log2n; ++s) {
        const int half = 1 << (s - 1), tstep = nfft >> s;
        for (int b = tid; b < nfft / 2; b += 256) {
            const int grp = b >> (s - 1), pos = b & (half - 1);
            const int i0 = (grp << s) + pos, i1 = i0 + half;
            const float wr = twr[pos * tstep], wi = twi[pos * tstep];
            const float xr = re[i1], xi = im[i1];
            const float tr = wr * xr - wi * xi, ti = wr * xi + wi * xr;
            const float ur = re[i0], ui = im[i0];
            re[i0] = ur + tr; im[i0] = ui + ti;
            re[i1] = ur - tr; im[i1] = ui - ti;
        }
        __syncthreads();
    }
    const int nbins = nfft / 2 + 1;
    for (int k = tid; k < nbins; k += 256) re[k] = re[k] * re[k] + im[k] * im[k];     // power, in place
    __syncthreads();
    // mel projection: groups of `nfilt` threads stride over the bins
    const int groups = 256 / nfilt;
    const int f = tid % nfilt, g = tid / nfilt;
    float acc = 0.0f;
    if (g < groups)
        for (int k = g; k < nbins; k += groups) acc = fmaf(re[k], melbank[(int64_t)k * nfilt + f], acc);
    part[tid] = (g < groups) ? acc : 0.0f;
    __syncthreads();
    if (tid < nfilt) {
        float e = 0.0f;
        for (int q = 0; q < groups; ++q) e += part[q * nfilt + tid];
        out[frame * nfilt + tid] = logf(fmaxf(e, FB_FLOOR));
    }
}

// ---------------------------------------------------------------------------------------
// nfft = 1024 (the reference's call, features.py:108): ONE WAVEFRONT per frame, persistent.
//   * the real frame x[0..1023] (wlen non-zero samples, zero padded) is taken as 512 complex
//     points z[m] = x[2m] + i x[2m+1]; a 512-point complex FFT and one split pass give the
//     513 bins of rfft(x, 1024): half the butterflies of the complex FFT of x;
//   * 512 = 8 x 8 x 8: three radix-8 Stockham passes, one 8-point DFT per lane and pass in
//     registers; the passes exchange data through a wave-private 4 KB strip of LDS (a
//     wavefront's LDS operations are ordered: no workgroup barrier anywhere);
//   * every twiddle a lane ever needs depends on the lane only: 7 + 7 for the passes and 4
//     for the split, computed ONCE per wavefront (sincospif) and kept in registers while the
//     wavefront walks its frames;
//   * the mel projection is sparse: lane f owns filter f and sums its own band of bins
//     [band[2f], band[2f+1]] (a triangle spans 4 .. 70 bins; the dense form read 513 x 40).
// HBM traffic stays the algorithmic 2 B per new sample + 160 B per frame; the mel table
// (82 KB) and the window live in L2 / L1.
// ---------------------------------------------------------------------------------------
struct cf { float x, y; };
__device__ __forceinline__ cf cmul(cf a, cf b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cf cadd(cf a, cf b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cf csub(cf a, cf b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cf mul_mi(cf a) { return {a.y, -a.x}; }            // a * (-i)

// forward 8-point DFT in place (natural order in, natural order out)
__device__ __forceinline__ void dft8(cf* v)
{
    const float h = 0.70710678118654752f;
    // even part: v0 v2 v4 v6, odd part: v1 v3 v5 v7 (4-point DFTs)
    cf e0 = cadd(v[0], v[4]), e1 = csub(v[0], v[4]), e2 = cadd(v[2], v[6]), e3 = mul_mi(csub(v[2], v[6]));
    cf o0 = cadd(v[1], v[5]), o1 = csub(v[1], v[5]), o2 = cadd(v[3], v[7]), o3 = mul_mi(csub(v[3], v[7]));
    cf E0 = cadd(e0, e2), E2 = csub(e0, e2), E1 = cadd(e1, e3), E3 = csub(e1, e3);
    cf O0 = cadd(o0, o2), O2 = csub(o0, o2), O1 = cadd(o1, o3), O3 = csub(o1, o3);
    // odd part times w8^k: w8 = (1 - i) / sqrt 2, w8^2 = -i, w8^3 = (-1 - i) / sqrt 2
    O1 = {h * (O1.x + O1.y), h * (O1.y - O1.x)};
    O2 = mul_mi(O2);
    O3 = {h * (O3.y - O3.x), -h * (O3.x + O3.y)};
    v[0] = cadd(E0, O0); v[4] = csub(E0, O0);
    v[1] = cadd(E1, O1); v[5] = csub(E1, O1);
    v[2] = cadd(E2, O2); v[6] = csub(E2, O2);
    v[3] = cadd(E3, O3); v[7] = csub(E3, O3);
}

constexpr int FB_WAVES = 4;            // wavefronts (frames in flight) per workgroup, one per SIMD (6 or 8 per workgroup: 0.87 / 0.74 ms against 0.69)
constexpr int FB_SPARSE_Q = 20;        // 16-byte pieces of a filter's band in LDS (40 mel triangles over 513 bins: the widest spans 18)

// Where element i of the wave's 512-point strip lives in LDS.  The passes read it lane-linearly (z[j + 64 r]: conflict free
// as it stands) but WRITE it at strides of 8 elements (pass 0: z[8 j + r]: sixteen lanes on two bank pairs, 8-way) and in
// 8-element runs 64 apart (pass 1: 2-way) -- profiles/r04_fbank_counters.txt: 40 % of the LDS-active cycles were conflicts.
// An XOR inside each block of 8 (by the block's index / 2) spreads pass 0's lanes over all banks, an XOR of the block's
// parity (by bit 3 of its index) separates pass 1's two runs; every 8-aligned run of 8 stays a permutation of itself, so
// the lane-linear reads remain conflict free.  (i >> 3 = B: 8 (B ^ ((B >> 3) & 1)) + ((i & 7) ^ ((B >> 1) & 7)).)
__device__ __forceinline__ int zsw(int i)
{
    const int B = i >> 3;
    return ((B ^ ((B >> 3) & 1)) << 3) + ((i & 7) ^ ((B >> 1) & 7));
}

// Three waves per SIMD: a frame is a chain of LDS round trips (three passes, split, projection), and at two waves per SIMD the
// vector units idled half of the time (profiles/r05_fbank_counters.txt).  What kept the kernel at 247 registers were the
// lane's constants, held through the frame loop: the 18 twiddles (52 registers) are the same for every wavefront of the
// workgroup and live in LDS tables now (6.5 KB, lane-linear reads: conflict free), read where they are used; the 16 window
// taps stay in registers (163 in all).  Three workgroups per CU need <= 53 KB of LDS each: 51.3 with 20 pieces per filter.
#ifndef FB_OCC
#define FB_OCC 3
#endif
__global__ __launch_bounds__(64 * FB_WAVES) __attribute__((amdgpu_waves_per_eu(FB_OCC, FB_OCC))) void fbank1024_kernel(const void* __restrict__ samples, int is_i16,
                                                                  int64_t nsamples, int wlen, double fshift, int nfilt,
                                                                  float alpha, const float* __restrict__ window,
                                                                  const float* __restrict__ melbank,
                                                                  const int32_t* __restrict__ band, int64_t nframes,
                                                                  float* __restrict__ out, const int64_t* __restrict__ utt_soff,
                                                                  const int64_t* __restrict__ utt_foff, int n_utts)
{
    __shared__ cf zbuf[FB_WAVES][512];
    __shared__ __attribute__((aligned(16))) float pw[FB_WAVES][516];
    // the filters' non-zero weights as 16-byte pieces aligned to the bins' index, piece q of filter f at [q][f]: the lanes of a
    // wave read consecutive pieces (conflict free; filter after filter, 4 bytes at a time, they met on the same banks)
    __shared__ __attribute__((aligned(16))) float sw4[FB_SPARSE_Q][64][4];
    // lane-only constants, one copy per workgroup: the twiddles of pass 1 (sub-transform size 8: angle -2 pi r (j & 7) / 64),
    // of pass 2 (-2 pi r j / 512) and of the split (e^{-2 pi i k / 1024}, k = j + 64 q)
    __shared__ cf tw1_s[8][8], tw2_s[8][64], tws_s[4][64];
    const int wave = threadIdx.x >> 6, j = threadIdx.x & 63;
    cf* const z = zbuf[wave];
    float* const power = pw[wave];

    for (int i = threadIdx.x; i < 8 * 64; i += 64 * FB_WAVES) {
        const int r = i >> 6, jj = i & 63;
        float sn, cs;
        sincospif(-2.0f * (float)(r * jj) / 512.0f, &sn, &cs);
        tw2_s[r][jj] = {cs, sn};
        if (jj < 8) {
            sincospif(-2.0f * (float)(r * jj) / 64.0f, &sn, &cs);
            tw1_s[r][jj] = {cs, sn};
        }
        if (r < 4) {
            sincospif(-2.0f * (float)(jj + 64 * r) / 1024.0f, &sn, &cs);
            tws_s[r][jj] = {cs, sn};
        }
    }
    // this lane's filter: its band of bins; the band's weights go to LDS once per workgroup
    const int blo = j < nfilt ? band[2 * j] : 1, bhi = j < nfilt ? band[2 * j + 1] : 0;
    const int a0 = blo & ~3;                                       // the band's first piece starts at a multiple of four bins
    const int nq = bhi >= blo ? (bhi - a0) / 4 + 1 : 0;            // pieces of this lane's band
    const bool sparse_ok = __all(nq <= FB_SPARSE_Q && bhi <= 512);
    const int nq_max = __builtin_amdgcn_readfirstlane(__reduce_max_sync(~0ull, nq));
    if (wave == 0 && sparse_ok) {
        for (int q = 0; q < nq_max; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = a0 + 4 * q + e;
                sw4[q][j][e] = (q < nq && k >= blo && k <= bhi) ? melbank[(int64_t)k * nfilt + j] : 0.0f;
            }
    }
    if (j < 4) power[512 + j] = 0.0f;                               // (bins 513 .. 515: never written below, read under zero weights)
    __syncthreads();
    // window taps of the samples this lane touches: elements m = j + 64 r, samples 2m, 2m+1
    float w0[8], w1[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int n0 = 2 * (j + 64 * r);
        w0[r] = n0 < wlen ? window[n0] : 0.0f;
        w1[r] = n0 + 1 < wlen ? window[n0 + 1] : 0.0f;
    }
    const int nwaves = gridDim.x * FB_WAVES;
    for (int64_t frame = (int64_t)blockIdx.x * FB_WAVES + wave; frame < nframes; frame += nwaves) {
        // (the tables' index is opaque to the compiler in every iteration: seen as loop invariants their loads are hoisted in
        // front of the loop and the constants are back in registers)
        int jt = j;
        asm volatile("" : "+v"(jt));
        int64_t fr = frame, len = nsamples;
        const char* base = (const char*)samples;
        if (n_utts > 0) {                                          // (wave-uniform: one frame per wavefront)
            const int u = find_utt(utt_foff, n_utts, frame);
            fr = frame - utt_foff[u];
            len = utt_soff[u + 1] - utt_soff[u];
            base += utt_soff[u] * (is_i16 ? 2 : 4);
        }
        const FrameSrc cur = frame_src(base, is_i16, len, fr, fshift, wlen);
        cf v[8];
        // pre-emphasis y[n] = x[n] - alpha x[n-1] inside the frame; element 0's history is the previous frame's
        // last element (FrameSrc).  Whole frames -- all but an utterance's last two -- read their samples directly.
        if (cur.avail >= wlen) {
            // element 0's history: the previous frame (whole too: it starts earlier) ends at sample prev_start + wlen - 1
            const int64_t i_prior = fr > 0 ? (int64_t)rint((double)(fr - 1) * fshift) + wlen - 1 : -1;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int n0 = 2 * (j + 64 * r);
                float a = 0.0f, b = 0.0f, c = 0.0f;                  // x[n0 - 1], x[n0], x[n0 + 1]
                if (n0 < wlen) {
                    const int64_t i = cur.start + n0;
                    const int64_t ia = n0 > 0 ? i - 1 : i_prior;
                    a = ia >= 0 ? cur.raw(ia) : 0.0f;
                    b = cur.raw(i);
                    c = n0 + 1 < wlen ? cur.raw(i + 1) : 0.0f;
                }
                v[r] = {(b - alpha * a) * w0[r], (c - alpha * b) * w1[r]};
            }
        } else {
            const FrameSrc prv = frame_src(base, is_i16, len, fr > 0 ? fr - 1 : 0, fshift, wlen);
            const float prior = fr > 0 ? prv.at(wlen - 1) : 0.0f;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int n0 = 2 * (j + 64 * r);
                float a = 0.0f, b = 0.0f, c = 0.0f;
                if (n0 < wlen) {
                    a = n0 > 0 ? cur.at(n0 - 1) : prior;
                    b = cur.at(n0);
                    c = n0 + 1 < wlen ? cur.at(n0 + 1) : 0.0f;
                }
                v[r] = {(b - alpha * a) * w0[r], (c - alpha * b) * w1[r]};
            }
        }
        // pass 0 (sub-transform size 1): no twiddles; outputs to 8 j + r
        dft8(v);
#pragma unroll
        for (int r = 0; r < 8; ++r) z[zsw(8 * j + r)] = v[r];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // pass 1 (size 8): k = j & 7; outputs to ((j - k) << 3) + k + 8 r
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = z[zsw(j + 64 * r)];
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], tw1_s[r][jt & 7]);
        dft8(v);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        {
            const int k = j & 7, j0 = ((j - k) << 3) + k;
#pragma unroll
            for (int r = 0; r < 8; ++r) z[zsw(j0 + 8 * r)] = v[r];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // pass 2 (size 64): k = j; outputs to j + 64 r: natural order
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = z[zsw(j + 64 * r)];
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], tw2_s[r][jt]);
        dft8(v);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 8; ++r) z[zsw(j + 64 * r)] = v[r];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // split: X[k] = (Z[k] + conj Z[512-k]) / 2 - i e^{-2 pi i k / 1024} (Z[k] - conj Z[512-k]) / 2, and its mirror
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = j + 64 * q;                              // 0 .. 255
            const cf a = z[zsw(k)], b = z[zsw((512 - k) & 511)];
            const cf s = {0.5f * (a.x + b.x), 0.5f * (a.y - b.y)};        // (Z[k] + conj Z[N-k]) / 2
            const cf d = {0.5f * (a.x - b.x), 0.5f * (a.y + b.y)};        // (Z[k] - conj Z[N-k]) / 2
            const cf t = cmul(tws_s[q][jt], d);                                 // w d ; -i w d = (t.y, -t.x)
            const cf xk = {s.x + t.y, s.y - t.x};
            // mirror bin 512 - k: conj(s) - (-i) conj(w) ... = conj(s) + i conj(w d) -> (s.x - t.y, -s.y - t.x)
            const cf xm = {s.x - t.y, -s.y - t.x};
            power[k] = xk.x * xk.x + xk.y * xk.y;
            power[512 - k] = xm.x * xm.x + xm.y * xm.y;            // k = 0: bin 512 = (Re Z0 - Im Z0)^2
        }
        if (j == 0) {                                              // bin 256: its own mirror, w = -i
            const cf a = z[zsw(256)];
            power[256] = a.x * a.x + a.y * a.y;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // mel projection + log
        float e = 0.0f;
        if (sparse_ok) {                               // 16 bytes of the band and of its weights per step, four partial sums
            typedef float fb4 __attribute__((ext_vector_type(4)));
            float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f;
            for (int q = 0; q < nq_max; ++q) {         // (every lane walks the widest band: past its own end the weights are zeros)
                const fb4 pk = *reinterpret_cast<const fb4*>(power + min(a0 + 4 * q, 512));
                const fb4 wk = *reinterpret_cast<const fb4*>(&sw4[q][j][0]);
                e0 = fmaf(pk[0], wk[0], e0); e1 = fmaf(pk[1], wk[1], e1);
                e2 = fmaf(pk[2], wk[2], e2); e3 = fmaf(pk[3], wk[3], e3);
            }
            e = (e0 + e1) + (e2 + e3);
        } else {
            for (int k = blo; k <= bhi; ++k) e = fmaf(power[k], melbank[(int64_t)k * nfilt + j], e);
        }
        if (j < nfilt) out[frame * nfilt + j] = logf(fmaxf(e, FB_FLOOR));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
}

// regression deltas over time (spectral's _deltas: a 9-tap slope filter, sum_n n x[t+n] / 60, the
// edges padded with copies of frame 1 and frame T-2): out[t][c]
__global__ void deltas_kernel(const float* __restrict__ x, int64_t T, int D, float* __restrict__ out)
{
    const int64_t n = T * D;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = i / D;
        const int c = (int)(i - t * D);
        float acc = 0.0f;
#pragma unroll
        for (int k = 1; k <= 4; ++k) {
            int64_t tp = t + k, tm = t - k;
            tp = tp >= T ? (T >= 2 ? T - 2 : 0) : tp;
            tm = tm < 0 ? (T >= 2 ? 1 : 0) : tm;
            acc += (float)k * (x[tp * D + c] - x[tm * D + c]);
        }
        out[i] = acc / 60.0f;
    }
}

}  // namespace abn

using namespace abn;

static int fbank_launch(const void* samples, int sample_is_i16, int64_t nsamples, const int64_t* utt_soff,
                        const int64_t* utt_foff, int64_t n_utts, int32_t wlen, double fshift, int32_t nfft, int32_t nfilt,
                        float alpha, const float* window, const float* melbank, const int32_t* band, int64_t nframes,
                        float* out, void* stream)
{
    ABN_REQUIRE(nframes >= 0 && nsamples >= 0, "fbank: negative sizes");
    if (nframes == 0) return ABN_OK;
    ABN_REQUIRE(samples && window && melbank && out, "fbank: null pointer");
    ABN_REQUIRE(nfft >= 64 && nfft <= FB_MAX_NFFT && (nfft & (nfft - 1)) == 0, "fbank: nfft=%d must be a power of two in [64, %d]", nfft, FB_MAX_NFFT);
    ABN_REQUIRE(wlen >= 1 && wlen <= nfft, "fbank: wlen=%d must be in [1, nfft]", wlen);
    ABN_REQUIRE(nfilt >= 1 && nfilt <= FB_MAX_FILT, "fbank: nfilt=%d out of range", nfilt);
    ABN_REQUIRE(fshift > 0.0, "fbank: frame shift must be positive");
    ABN_REQUIRE(n_utts >= 0 && n_utts < (1LL << 30) && nframes < (1LL << 31), "fbank: too many utterances / frames");
    hipStream_t st = (hipStream_t)stream;
    if (nfft == 1024 && nfilt <= 64 && band) {
        // persistent: enough wavefronts to fill the chip, each walks frames wave, wave + W, ...
        int64_t wgs = (nframes + FB_WAVES - 1) / FB_WAVES;
        if (wgs > 256 * 6) wgs = 256 * 6;
        hipLaunchKernelGGL(fbank1024_kernel, dim3((unsigned)wgs), dim3(64 * FB_WAVES), 0, st, samples, sample_is_i16, nsamples,
                           (int)wlen, fshift, (int)nfilt, alpha, window, melbank, band, nframes, out, utt_soff, utt_foff, (int)n_utts);
        ABN_CHECK_LAUNCH("fbank1024");
        return ABN_OK;
    }
    int log2n = 0;
    while ((1 << log2n) < nfft) ++log2n;
    hipLaunchKernelGGL(fbank_kernel, dim3((unsigned)nframes), dim3(256), 0, st, samples, sample_is_i16,
                       nsamples, (int)wlen, fshift, (int)nfft, log2n, (int)nfilt, alpha, window, melbank, out, utt_soff, utt_foff,
                       (int)n_utts);
    ABN_CHECK_LAUNCH("fbank");
    return ABN_OK;
}

extern "C" int abn_fbank(const void* samples, int sample_is_i16, int64_t nsamples, int32_t wlen, double fshift,
                         int32_t nfft, int32_t nfilt, float alpha, const float* window, const float* melbank,
                         const int32_t* band, int64_t nframes, float* out, void* stream)
{
    return fbank_launch(samples, sample_is_i16, nsamples, nullptr, nullptr, 0, wlen, fshift, nfft, nfilt, alpha, window, melbank,
                        band, nframes, out, stream);
}

extern "C" int abn_fbank_batched(const void* samples, int sample_is_i16, const int64_t* utt_sample_off,
                                 const int64_t* utt_frame_off, int64_t n_utts, int32_t wlen, double fshift, int32_t nfft,
                                 int32_t nfilt, float alpha, const float* window, const float* melbank, const int32_t* band,
                                 int64_t nframes, float* out, void* stream)
{
    ABN_REQUIRE(n_utts >= 1 && utt_sample_off && utt_frame_off, "fbank_batched: utterance tables missing");
    return fbank_launch(samples, sample_is_i16, 0, utt_sample_off, utt_frame_off, n_utts, wlen, fshift, nfft, nfilt, alpha, window,
                        melbank, band, nframes, out, stream);
}

extern "C" int abn_deltas(const float* feats, int64_t T, int64_t D, float* out, void* stream)
{
    ABN_REQUIRE(T >= 0 && D >= 1 && D < (1 << 24), "deltas: bad shape");
    if (T == 0) return ABN_OK;
    ABN_REQUIRE(feats && out && feats != out, "deltas: null or aliased pointer");
    const int64_t n = T * D;
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(deltas_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream, feats,
                       T, (int)D, out);
    ABN_CHECK_LAUNCH("deltas");
    return ABN_OK;
}
