// fbank.hip -- log mel filterbank extraction for gfx950.
//
// Replaces FeaturesGenerator.do_fbank, abnet3/features.py:99-114, i.e.
//   Spectral(nfilt=40, alpha=0.97, do_dct=False, fs, frate=100, wlen=0.025,
//            nfft=1024, ...).transform(sound)  -> float32 [T, 40]
// The arithmetic of the third-party `spectral` package is not in the
// reference; the definition implemented here is oracle/features_np.py (parity
// unpinned, see its header): per frame pre-emphasis -> Hamming window ->
// |rfft(nfft)|^2 -> triangular mel bank -> log(max(., 1e-5)).
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

__global__ __launch_bounds__(256) void fbank_kernel(const void* __restrict__ samples, int is_i16, int64_t nsamples,
                                                    int wlen, double fshift, int nfft, int log2n, int nfilt,
                                                    float alpha, const float* __restrict__ window,
                                                    const float* __restrict__ melbank, float* __restrict__ out)
{
    __shared__ float re[FB_MAX_NFFT], im[FB_MAX_NFFT];
    __shared__ float twr[FB_MAX_NFFT / 2], twi[FB_MAX_NFFT / 2];
    __shared__ float part[256];
    const int tid = threadIdx.x;
    const int64_t frame = blockIdx.x;
    const int64_t start = (int64_t)rint((double)frame * fshift);
    auto sample_at = [&](int64_t i) -> float {
        if (i < 0 || i >= nsamples) return 0.0f;
        return is_i16 ? (float)((const int16_t*)samples)[i] : ((const float*)samples)[i];
    };
    for (int n = tid; n < nfft; n += 256) {
        float v = 0.0f;
        if (n < wlen) {
            const int64_t i = start + n;
            // pre-emphasis runs over the zero-padded frame: its history is the
            // previous SIGNAL sample (0 before the first one), so the first
            // padded sample still sees -alpha * last (oracle/features_np.py)
            v = (sample_at(i) - alpha * sample_at(i - 1)) * window[n];
        }
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

}  // namespace abn

using namespace abn;

extern "C" int abn_fbank(const void* samples, int sample_is_i16, int64_t nsamples, int32_t wlen, double fshift,
                         int32_t nfft, int32_t nfilt, float alpha, const float* window, const float* melbank,
                         int64_t nframes, float* out, void* stream)
{
    ABN_REQUIRE(nframes >= 0 && nsamples >= 0, "fbank: negative sizes");
    if (nframes == 0) return ABN_OK;
    ABN_REQUIRE(samples && window && melbank && out, "fbank: null pointer");
    ABN_REQUIRE(nfft >= 64 && nfft <= FB_MAX_NFFT && (nfft & (nfft - 1)) == 0, "fbank: nfft=%d must be a power of two in [64, %d]", nfft, FB_MAX_NFFT);
    ABN_REQUIRE(wlen >= 1 && wlen <= nfft, "fbank: wlen=%d must be in [1, nfft]", wlen);
    ABN_REQUIRE(nfilt >= 1 && nfilt <= FB_MAX_FILT, "fbank: nfilt=%d out of range", nfilt);
    ABN_REQUIRE(fshift > 0.0, "fbank: frame shift must be positive");
    int log2n = 0;
    while ((1 << log2n) < nfft) ++log2n;
    hipLaunchKernelGGL(fbank_kernel, dim3((unsigned)nframes), dim3(256), 0, (hipStream_t)stream, samples, sample_is_i16,
                       nsamples, (int)wlen, fshift, (int)nfft, log2n, (int)nfilt, alpha, window, melbank, out);
    ABN_CHECK_LAUNCH("fbank");
    return ABN_OK;
}
