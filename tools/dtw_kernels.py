"""Per-kernel time of one batched DTW call (C4 inputs) from HIP events around ten calls and the torch profiler-free
split: the traceback alone is timed by running the call with and without it is not possible from outside, so this
prints the whole call and relies on rocprofv3 for the split:  rocprofv3 --kernel-trace --stats -- python3 tools/dtw_kernels.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from abnet3_amd.utils import dtw_align_batch
f1, o1, n1, f2, o2, n2 = bench.synth_dtw_pairs(10000, seed=1000)
d1, d2 = torch.from_numpy(f1).cuda(), torch.from_numpy(f2).cuda()
for _ in range(12):
    dtw_align_batch(d1, o1, n1, d2, o2, n2)
torch.cuda.synchronize()
