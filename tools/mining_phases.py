"""Cold against warm pair mining (OriginalDataLoader over the C5 corpus), phase by phase, with a synchronize after each phase:
where the first call of a process spends the time later calls do not."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tools import c5_pipeline
from tools.c5_corpus import sample_pairs, synth_corpus
corpus = synth_corpus(n_utts=2000, seed=0, device='cuda')
pairs = sample_pairs(corpus, n_pairs=50000, seed=0)
dc, _ = c5_pipeline.build_features(corpus, {})


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


for rep in range(3):
    dl = c5_pipeline.make_loader('original', dc, pairs[0], pairs[1])
    np.random.seed(0)
    t0 = sync()
    dl.load_data()
    t1 = sync()
    line = 'rep %d: load_data %.3f' % (rep, t1 - t0)
    for mode in ('train', 'dev'):
        same = [p[:6] for p in dl.pairs[mode] if p[6] == 'same']
        t2 = sync()
        dl.align_pairs(same, exchange=True)
        t3 = sync()
        dl._plan_store(mode)
        t4 = sync()
        line += ' | %s: align %.3f  plan %.3f' % (mode, t3 - t2, t4 - t3)
    print(line + ' | total %.3f s;  allocator reserved %.2f GB' % (sync() - t0, torch.cuda.memory_reserved() / 2 ** 30), flush=True)
