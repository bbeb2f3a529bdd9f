"""Where the host side of dtw_align_batch spends its time before the GPU has anything to do (C4 inputs)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from abnet3_amd import _lib, utils
f1, o1, n1, f2, o2, n2 = bench.synth_dtw_pairs(10000, seed=1000)
d1, d2 = torch.from_numpy(f1).cuda(), torch.from_numpy(f2).cuda()
lib = _lib.load()
T = {}
class Timed(object):
    def __init__(self, fn, name): self.fn, self.name = fn, name
    def __call__(self, *a):
        t0 = time.perf_counter(); r = self.fn(*a); T.setdefault(self.name, []).append(time.perf_counter() - t0); return r
class LibProxy(object):
    def __getattr__(self, k):
        return Timed(getattr(lib, k), k)
utils._lib.load = lambda: LibProxy()
for _ in range(3):
    utils.dtw_align_batch(d1, o1, n1, d2, o2, n2)
torch.cuda.synchronize()
T.clear()
tot = []
for _ in range(50):
    t0 = time.perf_counter()
    utils.dtw_align_batch(d1, o1, n1, d2, o2, n2)
    tot.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
print('dtw_align_batch returns after %.1f us (median of 50)' % (np.median(tot) * 1e6))
for k, v in T.items():
    print('   %-28s %.1f us' % (k, np.median(v) * 1e6))
