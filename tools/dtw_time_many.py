"""dtw_time.py with more repetitions: min and median of N calls (ABNET3_HIP_LIB selects the build)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from abnet3_amd.utils import dtw_align_batch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
f1, o1, n1, f2, o2, n2 = bench.synth_dtw_pairs(10000, seed=1000)
d1, d2 = torch.from_numpy(f1).cuda(), torch.from_numpy(f2).cuda()
for _ in range(3):
    dtw_align_batch(d1, o1, n1, d2, o2, n2)
torch.cuda.synchronize()
ts = []
for _ in range(N):
    t0 = time.perf_counter()
    dtw_align_batch(d1, o1, n1, d2, o2, n2)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
print('%-50s min %.3f  median %.3f ms' % (os.environ.get('ABNET3_HIP_LIB', 'default')[-50:], min(ts) * 1e3, float(np.median(ts)) * 1e3), flush=True)
# the same call after the GPU has idled (what a measurement of three calls behind a CPU-side set-up sees)
time.sleep(2.0)
cold = []
for _ in range(12):
    t0 = time.perf_counter()
    dtw_align_batch(d1, o1, n1, d2, o2, n2)
    torch.cuda.synchronize()
    cold.append(time.perf_counter() - t0)
print('after 2 s of idle, call by call (ms):', ' '.join('%.3f' % (t * 1e3) for t in cold), flush=True)
