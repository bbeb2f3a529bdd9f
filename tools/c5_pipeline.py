"""BASELINE.json configs[4] on one GPU, stage by stage: what the reference's GridSearch.run_single_experiment
does between features.generate() and embedder.embed() (abnet3/gridsearch.py:204-231), on the synthetic
ZeroSpeech-shaped corpus of tools/c5_corpus.py, with everything but the corpus itself on the product path:

    wav samples -> abn_fbank_batched -> mean / variance normalisation -> 7-frame stacking (280-d)
      -> word pairs: DTW alignment of the 'same' pairs (abn_dtw_batched), frame-pair index lists in HBM
      -> TrainerSiamese over OriginalDataLoader(batch_size = 8 word pairs)   [the reference's canonical loader]
         or FramesDataLoader(batch_size = 4096 frame pairs)
      -> EmbedderSiamese over every utterance

run() returns per-stage seconds, the frame pairs per second of the training passes and what the property
checks need (the loader, the trainer, the embeddings).  Measurement / test infrastructure.

    python tools/c5_pipeline.py [--utts 2000] [--pairs 50000] [--epochs 2] [--loader original|frames|both]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

C5_NET = dict(input_dim=280, num_hidden_layers=2, hidden_dim=500, output_dim=100, p_dropout=0.0, batch_norm=False,
              type_init='xavier_uni', activation_layer='sigmoid')        # test/data/buckeye.yaml:42-53


def _sync():
    import torch
    torch.cuda.synchronize()
    return time.perf_counter()


def build_features(corpus, stage):
    """wav samples -> (DeviceCorpus, table) with per-stage seconds added to `stage`."""
    import torch
    from abnet3_amd.dataloader import DeviceCorpus
    from abnet3_amd.features import FeaturesGenerator
    fg = FeaturesGenerator(normalization=True, norm_per_file=False, norm_per_channel=False, stack=True, nframes=7)
    t0 = _sync()
    lens = [len(w) for w in corpus.waves]
    flat = torch.from_numpy(np.concatenate(corpus.waves)).cuda()         # one host concatenation + one copy
    t1 = _sync()
    fb, nfr = fg.fbank_batch((flat, lens), corpus.fs)
    t2 = _sync()
    fb, _ = fg.normalize_table(fb, nfr)
    t3 = _sync()
    table = fg.stack_table(fb, nfr)
    t4 = _sync()
    times = {k: np.arange(int(n), dtype=float) * 0.01 + 0.0025 for k, n in zip(corpus.names, nfr)}
    dc = DeviceCorpus.from_table(table, corpus.names, nfr, times)
    stage.update(upload_s=t1 - t0, fbank_s=t2 - t1, normalise_s=t3 - t2, stack_s=t4 - t3)
    return dc, nfr


def make_loader(kind, dc, train_pairs, dev_pairs, batch_size=None):
    from abnet3_amd.dataloader import FramesDataLoader, OriginalDataLoader
    if kind == 'original':
        dl = OriginalDataLoader('unused', 'unused', num_max_minibatches=10000, seed=0, batch_size=batch_size or 8)
    else:
        dl = FramesDataLoader('unused', 'unused', batch_size=batch_size or 4096)
    dl.features = dc
    dl.pairs['train'], dl.pairs['dev'] = list(train_pairs), list(dev_pairs)
    dl.train_files = list({p[0] for p in train_pairs} | {p[3] for p in train_pairs})
    return dl


def train(kind, dc, train_pairs, dev_pairs, epochs, out_dir, seed=0, planned=True, batch_size=None):
    """One loader + trainer over `epochs` epochs; returns (stats, trainer, loader)."""
    import torch
    from abnet3_amd.loss import coscos2
    from abnet3_amd.model import SiameseNetwork
    from abnet3_amd.trainer import TrainerSiamese
    np.random.seed(seed)
    torch.manual_seed(seed)
    dl = make_loader(kind, dc, train_pairs, dev_pairs, batch_size)
    t0 = _sync()
    dl.load_data()
    if kind == 'original':                     # the alignment the iterator would do lazily, up front (timed as mining)
        dl._plan_store('train')
        dl._plan_store('dev')
    t1 = _sync()
    net = SiameseNetwork(output_path=os.path.join(out_dir, 'network_' + kind), **C5_NET)
    trainer = TrainerSiamese(network=net, loss=coscos2(avg=False), num_epochs=epochs, patience=30, optimizer_type='adadelta',
                             lr=0.1, dataloader=dl, log_dir=os.path.join(out_dir, 'runs'))
    trainer.time_passes = True
    trainer.planned_passes = planned
    t2 = _sync()
    trainer.train()
    t3 = _sync()
    if kind == 'original':
        offs = dl._plans['train'][2][3]
        fp_all = int(offs[-1])
        n_batches = len(offs) - 1
    else:
        fp_all = int(len(dl.frame_pairs['train'][2]) // dl.batch_size * dl.batch_size)
        n_batches = fp_all // dl.batch_size
    passes = trainer.pass_seconds                      # [(train s, dev s)]: the untrained first pass, then the epochs
    train_s = [p[0] for p in passes[1:]]
    visited = min(n_batches, getattr(dl, 'num_max_minibatches', n_batches)) if kind == 'original' else n_batches
    fp_epoch = fp_all if visited == n_batches else None
    stats = {
        'loader': type(dl).__name__, 'batch_size': dl.batch_size, 'planned_passes': bool(planned),
        'mining_s': round(t1 - t0, 4), 'train_total_s': round(t3 - t2, 4), 'epochs': epochs,
        'train_batches_per_epoch': visited, 'train_frame_pairs_per_epoch': fp_epoch,
        'mean_frame_pairs_per_batch': round(fp_all / max(1, n_batches), 1),
        'train_pass_s': [round(v, 4) for v in train_s], 'dev_pass_s': [round(p[1], 4) for p in passes[1:]],
        'first_untrained_pass_s': round(passes[0][0] + passes[0][1], 4),
        'train_losses': [float(v) for v in trainer.train_losses], 'dev_losses': [float(v) for v in trainer.dev_losses],
    }
    # the same mining once more, by a second loader over the same feature container (token lookups cached there, allocator and
    # code objects warm): what a second experiment on the corpus pays
    np.random.seed(seed)
    dl2 = make_loader(kind, dc, train_pairs, dev_pairs, batch_size)
    t4 = _sync()
    dl2.load_data()
    if kind == 'original':
        dl2._plan_store('train')
        dl2._plan_store('dev')
    stats['mining_s_second_loader'] = round(_sync() - t4, 4)
    del dl2
    if fp_epoch and train_s:
        best = min(train_s)
        stats['train_frame_pairs_per_s'] = round(fp_epoch / best, 1)
        stats['us_per_step'] = round(best / visited * 1e6, 2)
    return stats, trainer, dl


def run(n_utts=2000, n_pairs=50000, epochs=2, loaders=('original', 'frames'), seed=0, out_dir='/tmp/abnet3_c5', planned=True,
        corpus=None, pairs=None, keep=False):
    import torch
    from abnet3_amd.embedder import EmbedderSiamese
    from tools.c5_corpus import sample_pairs, synth_corpus
    os.makedirs(out_dir, exist_ok=True)
    stage = {}
    t0 = time.perf_counter()
    corpus = corpus or synth_corpus(n_utts=n_utts, seed=seed, device='cuda')
    train_pairs, dev_pairs = pairs or sample_pairs(corpus, n_pairs=n_pairs, seed=seed)
    stage['corpus_synthesis_s_untimed'] = round(time.perf_counter() - t0, 3)
    dc, nfr = build_features(corpus, stage)
    out = {'corpus': {'utterances': len(corpus.waves), 'audio_s': round(corpus.seconds(), 1), 'frames': int(dc.total),
                      'feature_dim': int(dc.dim), 'word_tokens': len(corpus.tokens), 'word_pairs_train': len(train_pairs),
                      'word_pairs_dev': len(dev_pairs)},
           'stages': {k: round(v, 4) for k, v in stage.items()}, 'training': {}}
    kept = {}
    for kind in loaders:
        stats, trainer, dl = train(kind, dc, train_pairs, dev_pairs, epochs, out_dir, seed, planned)
        t0 = _sync()
        emb = EmbedderSiamese(network=trainer.network, network_path=None, feature_path=None, output_path=None).embed_table(dc.table)
        stats['embed_s'] = round(_sync() - t0, 4)
        stats['embed_frames_per_s'] = round(dc.total / max(stats['embed_s'], 1e-9), 1)
        stats['embeddings_finite'] = bool(torch.isfinite(emb).all())
        out['training'][kind] = stats
        if keep:
            kept[kind] = (trainer, dl, emb)
    out['stages']['frames_per_s_fbank'] = round(dc.total / max(stage['fbank_s'], 1e-9), 1)
    return (out, kept, dc, corpus, (train_pairs, dev_pairs)) if keep else out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--utts', type=int, default=2000)
    ap.add_argument('--pairs', type=int, default=50000)
    ap.add_argument('--epochs', type=int, default=2)
    ap.add_argument('--loader', default='both')
    ap.add_argument('--no-plan', action='store_true', help='the plain batch iterator instead of planned passes (A/B)')
    args = ap.parse_args()
    loaders = ('original', 'frames') if args.loader == 'both' else (args.loader,)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):            # (the trainer prints its losses: the JSON line stands alone on stdout)
        out = run(args.utts, args.pairs, args.epochs, loaders, planned=not args.no_plan)
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
