#!/usr/bin/env python3
"""End-to-end run of the accelerated hot path on a synthetic ZeroSpeech-shaped
corpus (BASELINE.json configs[4]): wav -> log-mel filterbanks (HIP) -> mean /
variance normalisation -> 7-frame stacking -> word pairs -> DTW frame alignment
(HIP, batched) -> Siamese training (HIP) -> embedding.  Everything stays in
memory (the reference's h5features files are third-party I/O and out of scope).

    python examples/end_to_end.py [--utts 40] [--epochs 3] [--hidden 500]
    python -m torch.distributed.run --nproc-per-node N examples/end_to_end.py ...
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abnet3_amd import parallel                                   # noqa: E402
from abnet3_amd.dataloader import FramesDataLoader                # noqa: E402
from abnet3_amd.embedder import EmbedderSiamese                   # noqa: E402
from abnet3_amd.features import FeaturesGenerator                 # noqa: E402
from abnet3_amd.loss import coscos2                               # noqa: E402
from abnet3_amd.model import SiameseNetwork                       # noqa: E402
from abnet3_amd.trainer import TrainerSiamese                     # noqa: E402


def synth_corpus(n_utts, n_words, rng, fs=16000):
    """Utterances = concatenated 'words'; a word type is a fixed formant track
    plus noise, so same-type tokens are acoustically close (what term discovery
    clusters give the reference's sampler)."""
    word_len = rng.uniform(0.3, 0.8, n_words)
    formants = rng.uniform(300, 3000, (n_words, 3))
    wavs, tokens = {}, []                                   # tokens: (utt, start, end, word)
    for u in range(n_utts):
        t0, pieces = 0.0, []
        for _ in range(rng.integers(4, 9)):
            w = rng.integers(n_words)
            dur = word_len[w] * rng.uniform(0.85, 1.15)
            n = int(dur * fs)
            t = np.arange(n) / fs
            sig = sum(np.sin(2 * np.pi * f * (1 + 0.1 * np.sin(2 * np.pi * 3 * t)) * t) for f in formants[w])
            pieces.append(2500 * sig + 300 * rng.standard_normal(n))
            tokens.append(('utt%03d' % u, round(t0 + 0.01, 2), round(t0 + dur - 0.01, 2), int(w)))
            t0 += dur
        wavs['utt%03d' % u] = np.concatenate(pieces).astype(np.int16)
    return wavs, tokens


def sample_pairs(tokens, n_pairs, rng):
    by_word = {}
    for tok in tokens:
        by_word.setdefault(tok[3], []).append(tok)
    words = [w for w, v in by_word.items() if len(v) >= 2]
    pairs = []
    while len(pairs) < n_pairs:
        if len(pairs) % 2 == 0:
            w = words[rng.integers(len(words))]
            i, j = rng.choice(len(by_word[w]), 2, replace=False)
            a, b, kind = by_word[w][i], by_word[w][j], 'same'
        else:
            w1, w2 = rng.choice(words, 2, replace=False)
            a = by_word[w1][rng.integers(len(by_word[w1]))]
            b = by_word[w2][rng.integers(len(by_word[w2]))]
            kind = 'diff'
        pairs.append((a[0], a[1], a[2], b[0], b[1], b[2], kind))
    return pairs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--utts', type=int, default=40)
    ap.add_argument('--words', type=int, default=12)
    ap.add_argument('--pairs', type=int, default=400)
    ap.add_argument('--epochs', type=int, default=3)
    ap.add_argument('--hidden', type=int, default=500)
    ap.add_argument('--batch', type=int, default=4096)
    ap.add_argument('--out', default='/tmp/abnet3_e2e')
    args = ap.parse_args()
    rank, world, local = parallel.init_from_env(os.environ.get('ABN_DIST_BACKEND'))
    torch.cuda.set_device(local % torch.cuda.device_count())
    rng = np.random.default_rng(0)
    t = {}

    t0 = time.perf_counter()
    wavs, tokens = synth_corpus(args.utts, args.words, rng)
    fg = FeaturesGenerator(norm_per_channel=True)
    fb = {k: fg.fbank_from_samples(v, 16000).cpu().numpy() for k, v in wavs.items()}
    t['fbank'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    fb, _ = fg.normalize_features(fb)
    feats = {k: fg.stack_fbanks(v, nframes=7) for k, v in fb.items()}
    times = {k: np.arange(len(v)) * 0.01 + 0.0025 for k, v in feats.items()}
    t['normalise+stack'] = time.perf_counter() - t0

    pairs = sample_pairs(tokens, args.pairs, rng)
    split = int(0.7 * len(pairs))
    dl = FramesDataLoader('unused', 'unused', batch_size=args.batch)
    dl.set_data(feats, times, pairs[:split], pairs[split:])
    t0 = time.perf_counter()
    np.random.seed(0)
    dl.load_data()                                   # batched DTW alignment of all 'same' pairs
    torch.cuda.synchronize()
    t['dtw+frame pairs'] = time.perf_counter() - t0
    n_train = len(dl.frame_pairs['train'][2])

    torch.manual_seed(0)
    net = SiameseNetwork(input_dim=280, num_hidden_layers=2, hidden_dim=args.hidden, output_dim=100,
                         p_dropout=0.0, activation_layer='sigmoid', output_path=args.out + '_network')
    trainer = TrainerSiamese(network=net, loss=coscos2(avg=False), num_epochs=args.epochs, patience=30,
                             optimizer_type='adadelta', lr=0.1, dataloader=dl, log_dir=args.out + '_runs')
    t0 = time.perf_counter()
    trainer.train()
    torch.cuda.synchronize()
    t['train'] = time.perf_counter() - t0

    t0 = time.perf_counter()
    emb = EmbedderSiamese(network=net, network_path=args.out + '_network.pth', feature_path=None,
                          output_path=None).embed_features(list(feats.values()))
    t['embed'] = time.perf_counter() - t0
    if rank == 0:
        print('corpus: %d utterances, %d frames of 280-d stacked fbanks, %d word pairs, %d training frame pairs'
              % (len(feats), sum(len(v) for v in feats.values()), len(pairs), n_train))
        print('losses per epoch (train):', ['%.4f' % v for v in trainer.train_losses])
        print('losses per epoch (dev):  ', ['%.4f' % v for v in trainer.dev_losses])
        print('stage seconds:', {k: round(v, 3) for k, v in t.items()})
        print('embeddings: %d utterances, dim %d' % (len(emb), emb[0].shape[1]))
    return trainer, emb


if __name__ == '__main__':
    main()
