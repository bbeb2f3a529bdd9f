"""Phase cycles of the gang DTW kernel (library built with -DABN_DTW_STAMPS:
tools/variants.sh dtw "-DABN_DTW_STAMPS"; run with ABNET3_HIP_LIB=tools/variants/lib_ABN_DTW_STAMPS.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from abnet3_amd import _lib
from abnet3_amd.utils import dtw_align_batch
P = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
f1, o1, n1, f2, o2, n2 = bench.synth_dtw_pairs(P, seed=1000)
d1, d2 = torch.from_numpy(f1).cuda(), torch.from_numpy(f2).cuda()
dtw_align_batch(d1, o1, n1, d2, o2, n2); torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 16)()
raw.abn_debug_dtw_cycles(buf, 1)
dtw_align_batch(d1, o1, n1, d2, o2, n2); torch.cuda.synchronize()
raw.abn_debug_dtw_cycles(buf, 0)
c = np.array(list(buf), dtype=np.float64)
names = {0: 'P top: exit check, publish', 1: 'P produce (wait rows, MFMA, distances)', 2: 'P advance + next rows requested', 3: 'P barrier',
         8: 'C descriptor + boundary window', 9: 'C sweep', 10: 'C barrier'}
for grp, ks in (('producer (lane 0 of both)', [0, 1, 2, 3]), ('consumer', [8, 9, 10])):
    tot = c[ks].sum()
    print(grp, 'total %.3e cycles' % tot)
    for k in ks:
        print('   %-40s %5.1f %%' % (names[k], 100 * c[k] / tot))
