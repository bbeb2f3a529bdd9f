import sys, os, time
sys.path.insert(0, '/root/repo')
import torch, bench
from abnet3_amd.utils import dtw_align_batch
f1, o1, n1, f2, o2, n2 = bench.synth_dtw_pairs(10000, seed=1000)
d1, d2 = torch.from_numpy(f1).cuda(), torch.from_numpy(f2).cuda()
for keep in (False, True):
    held = []
    ts = []
    for i in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = dtw_align_batch(d1, o1, n1, d2, o2, n2)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append('%.1f/%.1f' % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
        if keep:
            held = [r]
    print('keep=%s  host/total ms:' % keep, ' '.join(ts), flush=True)
print(torch.cuda.memory_reserved() / 1e9, 'GB reserved')
