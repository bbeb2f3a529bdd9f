"""Times the batched DTW call (BASELINE config C4 inputs) per kernel with HIP
events around each of `reps` calls; used with ABNET3_HIP_LIB to compare builds.
  python tools/dtw_time.py [pairs]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from abnet3_amd.utils import dtw_align_batch

P = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
f1, o1, n1, f2, o2, n2 = bench.synth_dtw_pairs(P, seed=1000)
d1, d2 = torch.from_numpy(f1).cuda(), torch.from_numpy(f2).cuda()
dtw_align_batch(d1, o1, n1, d2, o2, n2)
torch.cuda.synchronize()
best = 1e9
for _ in range(4):
    t0 = time.perf_counter()
    dtw_align_batch(d1, o1, n1, d2, o2, n2)
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
print('%s  dtw %d pairs: %.3f ms' % (os.environ.get('ABNET3_HIP_LIB', 'default'), P, best * 1e3), flush=True)
# host-side cost: time until the call returns (launches are asynchronous)
torch.cuda.synchronize()
t0 = time.perf_counter()
dtw_align_batch(d1, o1, n1, d2, o2, n2)
t1 = time.perf_counter()
torch.cuda.synchronize()
print('host return after %.3f ms, done after %.3f ms' % ((t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
