"""The C5 pipeline's training leg with BatchNorm on (planned passes with abn_tower_desc.n_valid): two epochs over the 35 000 + 15 000
word pairs through TrainerSiamese.train(), planned against the plain iterator -- losses, finiteness, time per step."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tools import c5_pipeline
from tools.c5_corpus import sample_pairs, synth_corpus
c5_pipeline.C5_NET = dict(c5_pipeline.C5_NET, batch_norm=True)
corpus = synth_corpus(n_utts=int(os.environ.get('UTTS', '600')), seed=0, device='cuda')
train_pairs, dev_pairs = sample_pairs(corpus, n_pairs=int(os.environ.get('PAIRS', '12000')), seed=0)
dc, _ = c5_pipeline.build_features(corpus, {})
os.makedirs('/tmp/abn_c5_bn', exist_ok=True)
for planned in (True, False):
    stats, trainer, dl = c5_pipeline.train('original', dc, train_pairs, dev_pairs, 2, '/tmp/abn_c5_bn', 0, planned)
    graphs = sum(1 for v in getattr(trainer, '_buckets', {}).values() if v.get('graph') is not None)
    print(json.dumps({'planned': planned, 'captured_buckets': graphs, 'train_losses': stats['train_losses'], 'dev_losses': stats['dev_losses'],
                      'us_per_step': stats.get('us_per_step'), 'train_pass_s': stats['train_pass_s'], 'finite': bool(np.isfinite(stats['train_losses']).all())}), flush=True)
