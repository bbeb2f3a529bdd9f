"""DTW throughput at realistic word-token lengths (30-100 frames) next to the C4 benchmark lengths."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from abnet3_amd.utils import dtw_align_batch
for mean, sd, P in ((60, 15, 10000), (60, 15, 100000), (300, 60, 10000)):
    rng = np.random.default_rng(0)
    n1 = np.clip(np.rint(rng.normal(mean, sd, P)), 8, 1000).astype(np.int32)
    n2 = np.clip(np.rint(rng.normal(mean, sd, P)), 8, 1000).astype(np.int32)
    o1 = np.concatenate(([0], np.cumsum(n1)[:-1])).astype(np.int64)
    o2 = np.concatenate(([0], np.cumsum(n2)[:-1])).astype(np.int64)
    f1 = torch.randn(int(n1.sum()), 40, device='cuda')
    f2 = torch.randn(int(n2.sum()), 40, device='cuda')
    cells = int((n1.astype(np.int64) * n2).sum())
    dtw_align_batch(f1, o1, n1, f2, o2, n2); torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter(); dtw_align_batch(f1, o1, n1, f2, o2, n2); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print('tokens ~N(%d,%d)  %6d pairs  %.2f ms  %.2e cells/s  %.1f M pairs/s' % (mean, sd, P, best * 1e3, cells / best, P / best / 1e6), flush=True)
