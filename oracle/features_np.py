"""ORACLE (test infrastructure, never shipped as a product path).

Numpy restatement of the feature-extraction leg of the hot path:
  abnet3/features.py:135-159  stack_fbanks         (pinned: tests/golden/stack.npz)
  abnet3/features.py:99-114   do_fbank -> spectral.Spectral(...).transform

PARITY UNPINNED for the filterbank arithmetic: spectral.Spectral lives in the
un-vendored third-party package bootphon/spectral (requirements.txt:10, a git
URL with no version pin), absent from /root/reference; no reference test calls
do_fbank.  The only facts the reference fixes are the call-site arguments
(nfilt=40, alpha=0.97, do_dct=False, fs=srate, frate=100, wlen=0.025,
nfft=1024, no deltas) and the float32 cast.  This file restates the published
algorithm of that package's lineage (the CMU Sphinx-III `mfcc.py` front end its
parameter names come from) and IS the definition the HIP kernel is tested
against:
  frames:   nfr = int(len(sig)/fshift + 1), fshift = fs/frate, frame t starts
            at round(t*fshift), wlen = int(0.025*fs) samples.  A frame cut
            short by the end of the signal is np.resize()d to wlen -- its
            samples repeated CYCLICALLY (the lineage's `frame[wlen:] = 0` that
            follows is a no-op); an empty last frame is zeros
  pre-emph: y[0] = x[0] - alpha*prior, y[i] = x[i] - alpha*x[i-1] inside the
            frame, where prior = the LAST sample of the previous (overlapping,
            resized) frame, 0 for the first frame -- the lineage keeps
            `self.prior = frame[-1]` between frames; do_fbank builds a new
            Spectral per file (abnet3/features.py:101-113), so it starts at 0.
            (Rounds 1-2 of this build zero-padded the tail and took x[start-1]
            as the history: one sample per frame / the last two frames differ.)
  window:   numpy.hamming(wlen) (symmetric)
  power:    |rfft(frame, nfft)|^2, nfft=1024 -> 513 bins
  mel bank: 40 triangular filters between lowerf=133.3333 Hz and
            upperf=6855.4976 Hz, mel(f)=2595 log10(1+f/700), edges rounded to
            DFT bins, height 2/(width in Hz)
  output:   log(max(power . filters, 1e-5)) as float32 [nfr, 40]
"""
import numpy as np

LOWERF = 133.3333
UPPERF = 6855.4976
FLOOR = 1e-5


def stack_fbanks(features, nframes=7):
    """features.py:135-159: row t = concat(rows t-n//2 .. t+n//2), zero padded."""
    assert nframes % 2 == 1, 'number of stacked frames must be odd'
    T, dim = features.shape
    h = nframes // 2
    out = np.zeros((T, dim * nframes), dtype=features.dtype)
    for k in range(nframes):
        lo = max(0, h - k)
        hi = min(T, T + h - k)
        if hi > lo:
            out[lo:hi, k * dim:(k + 1) * dim] = features[lo + k - h:hi + k - h]
    return out


def mel(f):
    return 2595.0 * np.log10(1.0 + f / 700.0)


def melinv(m):
    return 700.0 * (np.power(10.0, m / 2595.0) - 1.0)


def mel_filterbank(fs, nfft=1024, nfilt=40, lowerf=LOWERF, upperf=UPPERF):
    """[nfft/2+1, nfilt] float64 triangular bank (see module docstring)."""
    if upperf > fs / 2:
        raise ValueError('Upper frequency %f exceeds Nyquist %f' % (upperf, fs / 2))
    filters = np.zeros((nfft // 2 + 1, nfilt), dtype=np.float64)
    dfreq = float(fs) / nfft
    melmax, melmin = mel(upperf), mel(lowerf)
    dmelbw = (melmax - melmin) / (nfilt + 1)
    edges = melinv(melmin + dmelbw * np.arange(nfilt + 2, dtype=np.float64))
    for w in range(nfilt):
        leftfr = int(round(edges[w] / dfreq))
        centerfr = int(round(edges[w + 1] / dfreq))
        rightfr = int(round(edges[w + 2] / dfreq))
        fwidth = (rightfr - leftfr) * dfreq
        height = 2.0 / fwidth
        leftslope = height / (centerfr - leftfr) if centerfr != leftfr else 0.0
        freq = leftfr + 1
        while freq < centerfr:
            filters[freq, w] = (freq - leftfr) * leftslope
            freq += 1
        if freq == centerfr:
            filters[freq, w] = height
            freq += 1
        if centerfr != rightfr:
            rightslope = height / (centerfr - rightfr)
        while freq < rightfr:
            filters[freq, w] = (freq - rightfr) * rightslope
            freq += 1
    return filters


def frame_count(nsamples, fs, frate=100):
    return int(nsamples / (float(fs) / frate) + 1)


def frame_samples(sig, t, fshift, wl):
    """Frame t of the signal, wl samples (see the module docstring: cyclic np.resize of a short tail)."""
    start = int(round(t * fshift))
    frame = sig[start:min(len(sig), start + wl)]
    if len(frame) < wl:
        frame = np.resize(frame, wl)
    return frame


def fbank(sig, fs, nfilt=40, alpha=0.97, frate=100, wlen=0.025, nfft=1024,
          dtype=np.float64):
    """Log mel filterbank energies, float32 [nfr, nfilt] (do_fbank's result).

    `dtype` is the arithmetic type of the per-frame pipeline: float64 is the
    definition; float32 mirrors what the HIP kernel computes in."""
    sig = np.asarray(sig).astype(np.float64)
    fshift = float(fs) / frate
    wl = int(wlen * fs)
    win = np.hamming(wl)
    filt = mel_filterbank(fs, nfft, nfilt)
    nfr = frame_count(len(sig), fs, frate)
    out = np.zeros((nfr, nfilt), dtype=np.float32)
    prior = 0.0
    for t in range(nfr):
        frame = frame_samples(sig, t, fshift, wl)
        prev = np.concatenate(([prior], frame[:-1]))
        prior = frame[-1]
        pre = ((frame - alpha * prev) * win).astype(dtype)
        spec = np.fft.rfft(pre.astype(np.float64), nfft)
        power = (spec.real * spec.real + spec.imag * spec.imag).astype(dtype)
        e = np.dot(power.astype(np.float64), filt)
        out[t] = np.log(np.clip(e, FLOOR, np.inf)).astype(np.float32)
    return out


def deltas(x):
    """do_deltas of the Spectral call (features.py:110): the package's regression deltas --
    a 9-tap slope filter, delta[t] = sum_{n=1..4} n (x[t+n] - x[t-n]) / 60 (60 = 2 sum n^2),
    the sequence padded with 4 copies of frame 1 in front and 4 copies of frame T-2 behind
    (the package pads with its second and second-to-last frames).  PARITY UNPINNED like the
    rest of the filterbank arithmetic; second-order deltas are deltas(deltas(x))."""
    x = np.asarray(x, dtype=np.float64)
    T = x.shape[0]
    first = x[1] if T >= 2 else x[0]
    last = x[T - 2] if T >= 2 else x[0]
    g = np.vstack([np.tile(first, (4, 1)), x, np.tile(last, (4, 1))])
    out = np.zeros_like(x)
    for n in range(1, 5):
        out += n * (g[4 + n:4 + n + T] - g[4 - n:4 - n + T])
    return (out / 60.0).astype(np.float32)


def fbank_with_deltas(sig, fs, do_deltas=True, do_deltasdeltas=True, **kw):
    """[T, nfilt * (1 + deltas + deltasdeltas)]: static energies, then the slopes."""
    fb = fbank(sig, fs, **kw)
    cols = [fb]
    d1 = deltas(fb)
    if do_deltas:
        cols.append(d1)
    if do_deltasdeltas:
        cols.append(deltas(d1))
    return np.hstack(cols).astype(np.float32)


def mvn(features, per_channel, stats_on=None, params=None):
    """abnet3/features.py:216-244 (and :283-293 per file): mean = np.mean,
    std = np.std over axis 0 (per channel) or None (whole spectrum) of
    `stats_on` (default: the features themselves, e.g. the VAD-filtered frames),
    out = (features - mean) / (std + finfo.eps).  Pinned by the reference's own
    use of numpy for this arithmetic (test/test_features.py:37-281)."""
    axis = 0 if per_channel else None
    ref = features if stats_on is None else stats_on
    if params is not None:
        mean, std = params['mean'], params['variance']
    else:
        mean, std = np.mean(ref, axis=axis), np.std(ref, axis=axis)
    eps = np.finfo(features.dtype).eps
    return (features - mean) / (std + eps), mean, std
