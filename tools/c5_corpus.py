"""Synthetic ZeroSpeech-2017-shaped corpus for BASELINE.json configs[4] (SURVEY.md 8d "C5"): 16 kHz mono
int16 utterances of 2-10 s made of word tokens of 0.3-1.0 s (formant glides plus white noise, so that tokens
of one word type are acoustically close and a DTW alignment between them is not the diagonal), word types
with Zipfian frequencies, and a pairs list in the sampler's format (abnet3/sampler.py:697-742: same / diff
word pairs, times to 1/100 s), split 70 / 30 into train and dev.  Test / bench infrastructure: not part of
the product path.

    corpus = synth_corpus(n_utts=2000, seed=0)          # waves: list of int16 numpy arrays
    pairs = sample_pairs(corpus, n_pairs=50000, seed=0) # [(f1, s1, e1, f2, s2, e2, 'same' | 'diff')]
"""
import numpy as np


class Corpus(object):
    def __init__(self, names, waves, tokens, n_types, fs):
        self.names, self.waves, self.tokens, self.n_types, self.fs = names, waves, tokens, n_types, fs

    def seconds(self):
        return sum(len(w) for w in self.waves) / float(self.fs)


def synth_corpus(n_utts=2000, n_types=600, seed=0, fs=16000, device=None, min_s=2.0, max_s=10.0):
    """Utterances = word tokens separated by short pauses.  The audio is synthesised with torch (on `device`
    when given -- the full-size corpus is 2e8 samples) from per-token tables; returns a Corpus whose `waves`
    are host int16 arrays (what scipy.io.wavfile.read would hand to do_fbank) and whose `tokens` are
    (utterance name, onset s, offset s, word type)."""
    import torch
    rng = np.random.default_rng(seed)
    # word types: a duration and three formant glides (start -> end frequency), Zipfian usage
    dur_w = rng.uniform(0.3, 1.0, n_types)
    f_start = np.sort(rng.uniform(250.0, 3200.0, (n_types, 3)), axis=1)
    f_end = np.clip(f_start * rng.uniform(0.7, 1.4, (n_types, 3)), 150.0, 3800.0)
    amp = rng.uniform(0.4, 1.0, (n_types, 3))
    zipf = 1.0 / np.arange(1, n_types + 1)
    zipf /= zipf.sum()
    # token table over the whole corpus
    tok_utt, tok_type, tok_len, tok_stretch, tok_shift = [], [], [], [], []
    utt_len = np.zeros(n_utts, dtype=np.int64)
    tokens = []
    names = ['utt%05d' % u for u in range(n_utts)]
    for u in range(n_utts):
        target = int(rng.uniform(min_s, max_s) * fs)
        pos = 0
        while True:
            gap = int(rng.uniform(0.03, 0.15) * fs)
            w = int(rng.choice(n_types, p=zipf))
            stretch = rng.uniform(0.85, 1.15)
            n = int(dur_w[w] * stretch * fs)
            if pos + gap + n > target and pos > 0:
                break
            tok_utt += [u, u]; tok_type += [-1, w]; tok_len += [gap, n]; tok_stretch += [1.0, stretch]
            tok_shift += [1.0, rng.normal(1.0, 0.02)]
            on, off = (pos + gap) / fs, (pos + gap + n) / fs
            tokens.append((names[u], round(on + 0.01, 2), round(off - 0.01, 2), w))
            pos += gap + n
        tail = int(rng.uniform(0.03, 0.15) * fs)
        tok_utt.append(u); tok_type.append(-1); tok_len.append(tail); tok_stretch.append(1.0); tok_shift.append(1.0)
        utt_len[u] = pos + tail
    tok_type = np.asarray(tok_type); tok_len = np.asarray(tok_len, dtype=np.int64)
    dev = torch.device(device) if device is not None else torch.device('cpu')
    total = int(tok_len.sum())
    tt = torch.from_numpy(np.where(tok_type < 0, 0, tok_type)).to(dev)
    voiced = torch.from_numpy((tok_type >= 0).astype(np.float32)).to(dev)
    t_len = torch.from_numpy(tok_len).to(dev)
    t_start = torch.cumsum(t_len, 0) - t_len
    t_dur = (t_len.double() / fs).float()
    shift = torch.from_numpy(np.asarray(tok_shift, dtype=np.float32)).to(dev)
    fs_t = torch.from_numpy(f_start.astype(np.float32)).to(dev)[tt] * shift[:, None]
    fe_t = torch.from_numpy(f_end.astype(np.float32)).to(dev)[tt] * shift[:, None]
    am_t = torch.from_numpy(amp.astype(np.float32)).to(dev)[tt] * voiced[:, None]
    gen = torch.Generator(device=dev).manual_seed(seed)
    out = torch.empty(total, dtype=torch.int16, device=dev)
    CH = 1 << 23
    tok_of = torch.repeat_interleave(torch.arange(len(tok_len), device=dev), t_len)
    for c0 in range(0, total, CH):
        c1 = min(total, c0 + CH)
        tk = tok_of[c0:c1]
        t = ((torch.arange(c0, c1, device=dev) - t_start[tk]).double() / fs).float()        # time inside the token
        d = t_dur[tk].clamp_min(1e-3)
        sig = torch.zeros(c1 - c0, device=dev)
        for k in range(3):
            f0, f1 = fs_t[tk, k], fe_t[tk, k]
            phase = 2.0 * np.pi * (f0 * t + 0.5 * (f1 - f0) / d * t * t)
            sig += am_t[tk, k] * torch.sin(phase)
        env = torch.sin(np.pi * (t / d).clamp(0, 1)) ** 0.5                                  # soft on / offset
        sig = 2500.0 * sig * env + 300.0 * torch.randn(c1 - c0, device=dev, generator=gen)
        out[c0:c1] = sig.clamp(-32000, 32000).to(torch.int16)
    host = out.cpu().numpy()
    off = np.concatenate(([0], np.cumsum(utt_len)))
    waves = [host[off[u]:off[u + 1]] for u in range(n_utts)]
    return Corpus(names, waves, tokens, n_types, fs)


def sample_pairs(corpus, n_pairs=50000, seed=0, ratio_same=0.5, max_size_cluster=20, train_ratio=0.7):
    """Word pairs in the sampler's output format: half same-type, half different-type (ratio_same_diff_type of
    test/data/buckeye.yaml:36), clusters capped at max_size_cluster tokens (:33), a 70 / 30 train / dev split
    (:34).  Returns (train_pairs, dev_pairs)."""
    rng = np.random.default_rng(seed + 1)
    by_type = {}
    for tok in corpus.tokens:
        if tok[2] - tok[1] >= 0.1:
            by_type.setdefault(tok[3], []).append(tok)
    clusters = []
    for w, toks in by_type.items():
        if len(toks) > max_size_cluster:
            toks = [toks[i] for i in rng.choice(len(toks), max_size_cluster, replace=False)]
        if len(toks) >= 2:
            clusters.append(toks)
    weights = np.array([len(c) * (len(c) - 1) / 2.0 for c in clusters])
    weights /= weights.sum()
    all_toks = [(ci, t) for ci, c in enumerate(clusters) for t in c]
    n_same = int(n_pairs * ratio_same)
    pairs = []
    cl = rng.choice(len(clusters), n_same, p=weights)
    for ci in cl:
        c = clusters[ci]
        i, j = rng.choice(len(c), 2, replace=False)
        a, b = c[i], c[j]
        pairs.append((a[0], a[1], a[2], b[0], b[1], b[2], 'same'))
    ia = rng.integers(len(all_toks), size=2 * (n_pairs - n_same))
    k = 0
    while len(pairs) < n_pairs:
        if k + 1 >= len(ia):
            ia = rng.integers(len(all_toks), size=2 * n_pairs)
            k = 0
        (ca, a), (cb, b) = all_toks[ia[k]], all_toks[ia[k + 1]]
        k += 2
        if ca == cb:
            continue
        pairs.append((a[0], a[1], a[2], b[0], b[1], b[2], 'diff'))
    order = rng.permutation(len(pairs))
    pairs = [pairs[i] for i in order]
    split = int(train_ratio * len(pairs))
    return pairs[:split], pairs[split:]
