"""Wall time of the C5 pair mining (load_data + both batch plans), unprofiled; three fresh loaders each."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tools import c5_pipeline
from tools.c5_corpus import sample_pairs, synth_corpus
corpus = synth_corpus(n_utts=2000, seed=0, device='cuda')
pairs = sample_pairs(corpus, n_pairs=50000, seed=0)
dc, _ = c5_pipeline.build_features(corpus, {})
for kind in ('original', 'frames'):
    for rep in range(3):
        dl = c5_pipeline.make_loader(kind, dc, pairs[0], pairs[1])
        np.random.seed(0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dl.load_data()
        if kind == 'original':
            dl._plan_store('train'); dl._plan_store('dev')
        torch.cuda.synchronize()
        print(kind, 'mining %.3f s' % (time.perf_counter() - t0), flush=True)
