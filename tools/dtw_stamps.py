"""Phase cycles of the producer / consumer DTW kernel (library built with -DABN_DTW_STAMPS:
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
names = {0: 'P loop top (after B)', 1: 'P schedule/desc', 2: 'P produce', 3: 'P early ring writes', 4: 'P wait A', 5: 'P late writes + wait B',
         8: 'C loop top', 9: 'C top load + stage', 10: 'C sweep + flush', 11: 'C wait A', 12: 'C wait B'}
for grp, ks in (('producer', [0, 1, 2, 3, 4, 5]), ('consumer', [8, 9, 10, 11, 12])):
    tot = c[ks].sum()
    print(grp, 'total %.3e cycles' % tot)
    for k in ks:
        print('   %-28s %5.1f %%' % (names[k], 100 * c[k] / tot))
