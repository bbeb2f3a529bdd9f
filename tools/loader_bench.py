"""Producer side at scale: FramesDataLoader (align once, frame batches) and
OriginalDataLoader (word-pair batches) on a synthetic corpus; stage timings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from abnet3_amd.dataloader import FramesDataLoader, OriginalDataLoader
rng = np.random.default_rng(0)
U, T = 200, 3000
feats = {'u%03d' % i: rng.standard_normal((T, 40)).astype(np.float32) for i in range(U)}
times = {k: np.arange(T) * 0.01 + 0.0025 for k in feats}
NP = int(sys.argv[1]) if len(sys.argv) > 1 else 20000


def pairs(n, kind):
    out = []
    for _ in range(n):
        f1, f2 = rng.integers(0, U, 2)
        s1, s2 = rng.uniform(0, 28, 2)
        d1, d2 = rng.uniform(0.3, 1.0, 2)
        out.append(('u%03d' % f1, round(s1, 2), round(s1 + d1, 2), 'u%03d' % f2, round(s2, 2), round(s2 + d2, 2), kind))
    return out
train = pairs(NP, 'same') + pairs(NP, 'diff')
devp = pairs(200, 'same') + pairs(200, 'diff')
t0 = time.perf_counter()
dl = FramesDataLoader('unused', 'unused', batch_size=4096)
dl.set_data(feats, times, train, devp)
t1 = time.perf_counter()
dl.load_data(); torch.cuda.synchronize()
t2 = time.perf_counter()
n = 0
for b in dl.batch_iterator(train_mode=True):
    n += 1
torch.cuda.synchronize()
t3 = time.perf_counter()
print('FramesDataLoader: corpus to HBM %.2f s, align + frame-pair list for %d word pairs %.2f s, %d batches of 4096 in %.2f s'
      % (t1 - t0, 2 * NP, t2 - t1, n, t3 - t2), flush=True)
ol = OriginalDataLoader('unused', 'unused', batch_size=8, num_max_minibatches=2000)
ol.set_data(feats, times, train[:8000] + train[NP:NP + 8000], devp)
t0 = time.perf_counter()
n = 0
for b in ol.batch_iterator(train_mode=True):
    n += 1
torch.cuda.synchronize()
print('OriginalDataLoader: %d batches of 8 word pairs in %.2f s' % (n, time.perf_counter() - t0), flush=True)
