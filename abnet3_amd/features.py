"""Feature extraction on MI355X: the hot-path part of abnet3/features.py.

Mirrors (file:line relative to the reference checkout)
  FeaturesGenerator.do_fbank      abnet3/features.py:99-114   -> abn_fbank
  FeaturesGenerator.stack_fbanks  abnet3/features.py:135-159  -> abn_stack_frames
The reference delegates the filterbank arithmetic to the third-party
spectral.Spectral (absent, unpinned); the definition implemented by the kernel
is written down in oracle/features_np.py.  The h5features file pipeline
(features.py:161-203, :299-404) is I/O orchestration and out of scope.
"""
import numpy as np
import torch

from . import _lib

LOWERF = 133.3333
UPPERF = 6855.4976


def _mel(f):
    return 2595.0 * np.log10(1.0 + f / 700.0)


def _melinv(m):
    return 700.0 * (np.power(10.0, m / 2595.0) - 1.0)


def mel_filterbank(fs, nfft=1024, nfilt=40, lowerf=LOWERF, upperf=UPPERF):
    """Triangular mel bank [nfft/2+1, nfilt] (host, built once per sampling
    rate): edges equally spaced on the mel scale between lowerf and upperf,
    rounded to DFT bins, height 2 / (width in Hz)."""
    if upperf > fs / 2:
        raise ValueError('Upper frequency %f exceeds Nyquist %f' % (upperf, fs / 2))
    bank = np.zeros((nfft // 2 + 1, nfilt), dtype=np.float64)
    dfreq = float(fs) / nfft
    lo, hi = _mel(lowerf), _mel(upperf)
    edges = _melinv(lo + (hi - lo) / (nfilt + 1) * np.arange(nfilt + 2, dtype=np.float64))
    bins = [int(round(e / dfreq)) for e in edges]
    for w in range(nfilt):
        left, center, right = bins[w], bins[w + 1], bins[w + 2]
        height = 2.0 / ((right - left) * dfreq)
        for k in range(left + 1, center):
            bank[k, w] = (k - left) * height / (center - left)
        if center > left:
            bank[center, w] = height
        for k in range(center + 1, right):
            bank[k, w] = (k - right) * height / (center - right)
    return bank


class FeaturesGenerator:
    """Filterbank front end (abnet3/features.py:18-98 constructor surface kept
    for the hot-path arguments)."""

    def __init__(self, files=None, output_path=None, method='fbanks', n_filters=40,
                 save=True, load_mean_variance_path=None, save_mean_variance_path=None,
                 vad_file=None, normalization=True, norm_per_file=False,
                 norm_per_channel=False, stack=True, nframes=7, deltas=False,
                 deltasdeltas=False, run='once'):
        self.files = files
        self.output_path = output_path
        self.method = method
        self.n_filters = n_filters
        self.stack = stack
        self.nframes = nframes
        self.normalization = normalization
        self.norm_per_file = norm_per_file
        self.norm_per_channel = norm_per_channel
        self.vad_file = vad_file
        self.load_mean_variance_path = load_mean_variance_path
        self.save_mean_variance_path = save_mean_variance_path
        self.deltas = deltas
        self.deltasdeltas = deltasdeltas
        self.run = run
        self._tables = {}

    def whoami(self):
        return {'params': self.__dict__, 'class_name': self.__class__.__name__}

    def _table(self, fs, wlen, nfft, device):
        key = (fs, wlen, nfft, self.n_filters, str(device))
        if key not in self._tables:
            win = torch.from_numpy(np.hamming(wlen).astype(np.float32)).to(device)
            bank64 = mel_filterbank(fs, nfft, self.n_filters)
            bank = torch.from_numpy(bank64.astype(np.float32)).to(device)
            # first / last bin with a non-zero weight of every filter (the sparse projection)
            band = np.zeros((self.n_filters, 2), dtype=np.int32)
            for f in range(self.n_filters):
                nz = np.nonzero(bank64[:, f])[0]
                band[f] = (nz[0], nz[-1]) if len(nz) else (1, 0)
            self._tables[key] = (win, bank, torch.from_numpy(band).to(device))
        return self._tables[key]

    def deltas_of(self, feats):
        """Regression deltas of a [T, D] device tensor (spectral's do_deltas): slope over +-4
        frames, edges padded with frame 1 / frame T-2 (oracle/features_np.py)."""
        lib = _lib.load()
        feats = feats.contiguous()
        out = torch.empty_like(feats)
        _lib.check(lib.abn_deltas(_lib.ptr(feats), feats.shape[0], feats.shape[1], _lib.ptr(out),
                                  _lib.stream()), 'abn_deltas')
        return out

    def fbank_from_samples(self, sound, srate, alpha=0.97, frate=100, wlen=0.025,
                           nfft=1024):
        """Log mel energies [T, n_filters] float32 (device tensor) from int16 or
        float mono samples (numpy array or device tensor); with deltas / deltasdeltas
        (features.py:110-111) the slopes are appended as further columns:
        [T, n_filters * (1 + deltas + deltasdeltas)]."""
        lib = _lib.load()
        if isinstance(sound, torch.Tensor):
            s = sound
        else:
            sound = np.ascontiguousarray(sound)
            if sound.dtype != np.int16:
                sound = sound.astype(np.float32)
            s = torch.from_numpy(sound)
        s = s.cuda().contiguous()
        if s.dtype not in (torch.int16, torch.float32):
            s = s.float()
        wl = int(wlen * srate)
        fshift = float(srate) / frate
        nfr = int(s.numel() / fshift + 1)
        win, bank, band = self._table(srate, wl, nfft, s.device)
        out = torch.empty(nfr, self.n_filters, dtype=torch.float32, device=s.device)
        _lib.check(lib.abn_fbank(_lib.ptr(s), int(s.dtype == torch.int16), s.numel(), wl,
                                 fshift, nfft, self.n_filters, alpha, _lib.ptr(win),
                                 _lib.ptr(bank), _lib.ptr(band), nfr, _lib.ptr(out), _lib.stream()),
                   'abn_fbank')
        if self.deltas or self.deltasdeltas:
            d1 = self.deltas_of(out)
            cols = [out] + ([d1] if self.deltas else [])
            if self.deltasdeltas:
                cols.append(self.deltas_of(d1))
            out = torch.cat(cols, dim=1)
        return out

    # -- a whole corpus at once, device resident (features.py:160-203, :345-404 without the files) -------
    def fbank_batch(self, waves, srate, alpha=0.97, frate=100, wlen=0.025, nfft=1024):
        """do_fbank for a list of utterances in ONE launch (abn_fbank_batched): `waves` = int16 (or float)
        numpy arrays, or one device tensor per utterance, or (device tensor of all samples, sample counts).  Returns (table [sum T_u, n_filters * (1 + deltas +
        deltasdeltas)] float32 on the device, frame counts [T_u]); utterance u is framed on its own, exactly as
        fbank_from_samples(waves[u]) frames it."""
        lib = _lib.load()
        if isinstance(waves, tuple):             # (all samples end to end on the device, sample counts)
            s, lens = waves[0], np.asarray(waves[1], dtype=np.int64)
            assert int(lens.sum()) == s.numel()
        else:
            lens = np.array([len(w) for w in waves], dtype=np.int64)
        if isinstance(waves, tuple):
            pass
        elif isinstance(waves[0], torch.Tensor):
            s = torch.cat([w.reshape(-1) for w in waves]).cuda()
        else:
            dt = np.int16 if all(w.dtype == np.int16 for w in waves) else np.float32
            s = torch.from_numpy(np.concatenate([np.asarray(w, dtype=dt).reshape(-1) for w in waves])).cuda()
        if s.dtype not in (torch.int16, torch.float32):
            s = s.float()
        s = s.contiguous()
        wl = int(wlen * srate)
        fshift = float(srate) / frate
        nfr = (lens / fshift + 1).astype(np.int64)           # int(len / fshift + 1) per utterance
        soff = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
        foff = np.concatenate(([0], np.cumsum(nfr))).astype(np.int64)
        total = int(foff[-1])
        tables = torch.from_numpy(np.concatenate((soff, foff))).cuda()
        win, bank, band = self._table(srate, wl, nfft, s.device)
        out = torch.empty(total, self.n_filters, dtype=torch.float32, device=s.device)
        _lib.check(lib.abn_fbank_batched(_lib.ptr(s), int(s.dtype == torch.int16), _lib.ptr(tables[:len(soff)]),
                                         _lib.ptr(tables[len(soff):]), len(lens), wl, fshift, nfft, self.n_filters, alpha,
                                         _lib.ptr(win), _lib.ptr(bank), _lib.ptr(band), total, _lib.ptr(out), _lib.stream()),
                   'abn_fbank_batched')
        if self.deltas or self.deltasdeltas:               # slopes never cross an utterance boundary: per utterance
            cols = [out]
            d1 = torch.cat([self.deltas_of(out[foff[u]:foff[u + 1]]) for u in range(len(lens))])
            if self.deltas:
                cols.append(d1)
            if self.deltasdeltas:
                cols.append(torch.cat([self.deltas_of(d1[foff[u]:foff[u + 1]]) for u in range(len(lens))]))
            out = torch.cat(cols, dim=1)
        return out, nfr

    def normalize_table(self, table, lengths):
        """mean_variance_normalisation / mean_var_norm_per_file (features.py:205-297) on a device table of
        utterances laid end to end (no VAD here: normalize_features takes one).  Returns (table, stats)."""
        if self.norm_per_file:
            out, stats, o = torch.empty_like(table), [], 0
            for n in lengths:
                n = int(n)
                mean, std = self._stats(table[o:o + n])
                out[o:o + n] = self._apply(table[o:o + n], mean, std)
                stats.append((mean, std))
                o += n
            return out, stats
        mean, std = self._stats(table)
        return self._apply(table, mean, std), (mean, std)

    def stack_table(self, table, lengths, nframes=None):
        """stack_fbanks of every utterance of a device table in one launch (abn_stack_frames_batched)."""
        nframes = self.nframes if nframes is None else nframes
        assert nframes % 2 == 1, 'number of stacked frames must be odd'
        lib = _lib.load()
        foff = torch.from_numpy(np.concatenate(([0], np.cumsum(np.asarray(lengths, dtype=np.int64))))).cuda()
        T, D = table.shape
        out = torch.empty(T, D * nframes, dtype=torch.float32, device=table.device)
        _lib.check(lib.abn_stack_frames_batched(_lib.ptr(table.contiguous()), _lib.ptr(foff), len(lengths), T, D, nframes,
                                                _lib.ptr(out), _lib.stream()), 'abn_stack_frames_batched')
        return out

    def features_from_waves(self, waves, srate, names=None):
        """generate() (features.py:365-404) for in-memory audio, everything staying in HBM: filterbanks ->
        [normalisation] -> [stacking], one launch per stage for the whole corpus.  Returns
        (table [frames, dim], names, frame counts, {name: frame times}) -- DeviceCorpus.from_table's arguments;
        times as h5features_compute writes them (features.py:195)."""
        if isinstance(waves, dict):
            names, waves = list(waves.keys()), list(waves.values())
        names = list(names) if names is not None else ['utt%06d' % i for i in range(len(waves))]
        table, nfr = self.fbank_batch(waves, srate)
        if self.normalization:
            table, _ = self.normalize_table(table, nfr)
        if self.stack:
            table = self.stack_table(table, nfr)
        times = {k: np.arange(int(n), dtype=float) * 0.01 + 0.0025 for k, n in zip(names, nfr)}
        return table, names, nfr, times

    def generate(self):
        """The file-level entry point the gridsearch calls (abnet3/features.py:365-404, gridsearch.py:206-215): the wav
        files of `self.files` (a directory or a list) -> filterbanks -> [normalisation] -> [stacking] ->
        `self.output_path`, an h5features file with one item per wav (its basename), frame times
        0.0025 + 0.01 k (features.py:195) and float32 features.  The stages run on the whole corpus in HBM
        (features_from_waves: one launch per stage; the reference's two temporary h5features files do not exist);
        statistics over a VAD's frames only and injected statistics (load_mean_variance_path) go through
        normalize_features; save_mean_variance_path is honoured.  Needs the h5features package for the output
        (absent from the build image: features_from_waves is the in-memory form)."""
        import os
        from scipy.io import wavfile
        if self.method != 'fbanks':
            raise ValueError("Method %s not authorized." % self.method if self.method != 'mfcc' else
                             "abnet3_amd: method 'mfcc' is not on the accelerated path (filterbanks only)")
        files = self.files
        if isinstance(files, str):
            if not os.path.isdir(files):
                raise ValueError("files must be a directory or a list of files")
            files = [os.path.join(files, f) for f in sorted(os.listdir(files)) if f.endswith('.wav')]
        try:
            import h5features
        except ImportError:
            raise ImportError('FeaturesGenerator.generate() writes an h5features file like the reference; the '
                              'h5features package is not installed.  Use features_from_waves() on in-memory audio.')
        names, waves, rate = [], [], None
        for f in files:
            srate, sound = wavfile.read(f)
            if rate is not None and srate != rate:
                raise ValueError('abnet3_amd: %s is sampled at %d Hz, the files before it at %d' % (f, srate, rate))
            rate = srate
            names.append(os.path.basename(os.path.splitext(f)[0]))
            waves.append(sound)
        special = self.normalization and (self.vad_file is not None or self.load_mean_variance_path is not None)
        keep_norm, keep_stack = self.normalization, self.stack
        if special or (self.normalization and self.save_mean_variance_path):
            self.normalization = self.stack = False          # filterbanks only; the other stages follow below
        try:
            table, names, nfr, times = self.features_from_waves(waves, rate, names)
        finally:
            self.normalization, self.stack = keep_norm, keep_stack
        offs = np.concatenate(([0], np.cumsum(nfr))).astype(np.int64)
        if special or (self.normalization and self.save_mean_variance_path):
            from .utils import read_vad_file
            feats = {k: table[offs[i]:offs[i + 1]].cpu().numpy() for i, k in enumerate(names)}
            params = self.load_mean_variance(self.load_mean_variance_path) if self.load_mean_variance_path is not None else None
            vad = read_vad_file(self.vad_file) if self.vad_file is not None else None
            feats, stats = self.normalize_features(feats, times, vad, params)
            if self.save_mean_variance_path and not self.norm_per_file:
                self.save_mean_variance(np.atleast_1d(stats[0]), np.atleast_1d(stats[1]), self.save_mean_variance_path)
            if self.stack:
                feats = {k: self.stack_fbanks(v, nframes=self.nframes) for k, v in feats.items()}
            arrays = [np.ascontiguousarray(feats[k], dtype=np.float32) for k in names]
        else:
            host = table.cpu().numpy()
            arrays = [host[offs[i]:offs[i + 1]] for i in range(len(names))]
        if os.path.dirname(self.output_path):
            os.makedirs(os.path.dirname(self.output_path), exist_ok=True)
        h5features.write(self.output_path, '/features/', names, [times[k] for k in names], arrays)

    def do_fbank(self, fname):
        """Compute standard filterbanks from a wav file (features.py:99-114)."""
        from scipy.io import wavfile
        srate, sound = wavfile.read(fname)
        return self.fbank_from_samples(sound, srate).cpu().numpy()

    def stack_fbanks(self, features, nframes=7):
        """Each frame becomes the concatenation of its nframes//2 previous and
        next frames, zero padded at the edges (features.py:135-159).  numpy in
        -> numpy out, device tensor in -> device tensor out."""
        assert nframes % 2 == 1, 'number of stacked frames must be odd'
        lib = _lib.load()
        is_np = not isinstance(features, torch.Tensor)
        f = torch.from_numpy(np.ascontiguousarray(features)) if is_np else features
        dtype = f.dtype
        f = f.cuda().float().contiguous()
        T, D = f.shape
        out = torch.empty(T, D * nframes, dtype=torch.float32, device=f.device)
        _lib.check(lib.abn_stack_frames(_lib.ptr(f), T, D, nframes, _lib.ptr(out),
                                        _lib.stream()), 'abn_stack_frames')
        out = out.to(dtype)
        return out.cpu().numpy() if is_np else out

    # -- mean / variance normalisation (abnet3/features.py:205-297, :322-363) --------
    def _stats(self, feats_dev):
        """(mean, std) of a [T, D] device tensor: per channel ([D]) or over the
        whole spectrum ([1]), np.mean / np.std semantics."""
        lib = _lib.load()
        T, D = feats_dev.shape
        k = D if self.norm_per_channel else 1
        mean = torch.empty(k, dtype=torch.float32, device=feats_dev.device)
        std = torch.empty(k, dtype=torch.float32, device=feats_dev.device)
        ws = torch.empty(lib.abn_mvn_ws_bytes(T, D), dtype=torch.uint8, device=feats_dev.device)
        _lib.check(lib.abn_mvn_stats(_lib.ptr(feats_dev), T, D, int(bool(self.norm_per_channel)),
                                     _lib.ptr(mean), _lib.ptr(std), _lib.ptr(ws), _lib.stream()),
                   'abn_mvn_stats')
        return mean, std

    def _apply(self, feats_dev, mean, std):
        lib = _lib.load()
        T, D = feats_dev.shape
        out = torch.empty_like(feats_dev)
        eps = float(np.finfo(np.float32).eps)
        _lib.check(lib.abn_mvn_apply(_lib.ptr(feats_dev), T, D, _lib.ptr(mean), _lib.ptr(std),
                                     int(mean.numel() > 1), eps, _lib.ptr(out), _lib.stream()),
                   'abn_mvn_apply')
        return out

    @staticmethod
    def _vad_rows(times, segments):
        from .utils import Features_Accessor
        idx = [Features_Accessor.get_indices_between(times, s, e) for s, e in segments]
        return np.concatenate(idx) if idx else np.zeros(0, dtype=np.int64)

    def normalize_features(self, features, times=None, vad=None, params=None):
        """In-memory form of normalize() (features.py:345-363): `features` is
        {utt: [T, D] float32}; `vad` {utt: [[start, end], ...]} restricts the
        frames the statistics are computed on (needs `times`); `params`
        {'mean', 'variance'} skips the statistics (load_mean_variance).
        Returns ({utt: normalised array}, stats) where stats is (mean, std) for
        the global mode or [(utt, mean, std), ...] per file."""
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
        if self.norm_per_file:
            out, stats = {}, []
            for f, x in features.items():
                xd = dev(x)
                sd = xd
                if vad is not None and str(f) in vad:
                    rows = self._vad_rows(times[f], vad[str(f)])
                    sd = xd[torch.from_numpy(rows).cuda()]
                mean, std = self._stats(sd)
                out[f] = self._apply(xd, mean, std).cpu().numpy()
                stats.append((f, mean.cpu().numpy(), std.cpu().numpy()))
            return out, stats
        names = list(features.keys())
        if params is not None:
            mean = dev(np.atleast_1d(params['mean']))
            std = dev(np.atleast_1d(params['variance']))
        else:
            parts = []
            for f in names:
                x = features[f]
                if vad is not None and str(f) in vad:
                    x = x[self._vad_rows(times[f], vad[str(f)])]
                parts.append(x)
            mean, std = self._stats(dev(np.vstack(parts)))
        out = {f: self._apply(dev(features[f]), mean, std).cpu().numpy() for f in names}
        return out, (mean.cpu().numpy() if mean.numel() > 1 else float(mean.cpu()[0]),
                     std.cpu().numpy() if std.numel() > 1 else float(std.cpu()[0]))

    def save_mean_variance(self, mean, variance, output_file):
        np.savetxt(output_file, np.vstack((mean, variance)))

    def load_mean_variance(self, file_path):
        mean_var = np.loadtxt(file_path)
        return {'mean': mean_var[0], 'variance': mean_var[1]}
