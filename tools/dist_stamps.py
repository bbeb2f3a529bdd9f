"""Diagnostic (ABN_DIST_STAMPS build): average cycles per phase of dist_kernel blocks."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from abnet3_amd import _lib
from abnet3_amd.utils import dtw_align_batch
f1, o1, n1, f2, o2, n2 = bench.synth_dtw_pairs(10000, seed=1000)
d1, d2 = torch.from_numpy(f1).cuda(), torch.from_numpy(f2).cuda()
dtw_align_batch(d1, o1, n1, d2, o2, n2)
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 8)()
raw.abn_debug_dist_cycles(out)
n = out[0]
names = ['blocks', 'stage operands (+barrier)', 'norms + MFMA', 'norm barrier', 'acos epilogue (+barrier)', 'write-out issue']
print('blocks', n)
for q in range(1, 6):
    print('%-28s %8.0f cycles/block' % (names[q], out[q] / n))
print('sum %.0f cycles' % (sum(out[1:6]) / n))
