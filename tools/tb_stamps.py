"""When the wavefronts of the overlapped traceback start, see their flags and end, against the fill kernel beside them
(library built with -DABN_TB_STAMPS: tools/variants.sh dtw "-DABN_TB_STAMPS"; ABNET3_HIP_LIB=tools/variants/lib_ABN_TB_STAMPS.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from abnet3_amd import _lib
from abnet3_amd.utils import dtw_align_batch
P = 10000
f1, o1, n1, f2, o2, n2 = bench.synth_dtw_pairs(P, seed=1000)
d1, d2 = torch.from_numpy(f1).cuda(), torch.from_numpy(f2).cuda()
for _ in range(3):
    dtw_align_batch(d1, o1, n1, d2, o2, n2); torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros((4096, 4), dtype=np.uint64)
raw.abn_debug_tb_stamps(buf.ctypes.data_as(ctypes.c_void_p), 1)
dtw_align_batch(d1, o1, n1, d2, o2, n2); torch.cuda.synchronize()
raw.abn_debug_tb_stamps(buf.ctypes.data_as(ctypes.c_void_p), 0)
nw = (P + 63) // 64
b = buf[:nw].astype(np.float64)
t0 = b[:, 0].min()
us = lambda x: (x - t0) / 100.0
print('wave  start_us  flags_us  end_us  steps(lane?)')
for w in list(range(0, nw, 8)) + [nw - 1]:
    print('%4d %9.1f %9.1f %9.1f %6d' % (w, us(b[w, 0]), us(b[w, 1]), us(b[w, 2]), b[w, 3]))
print('starts: min %.1f max %.1f; flags max %.1f; end max %.1f' % (us(b[:, 0]).min(), us(b[:, 0]).max(), us(b[:, 1]).max(), us(b[:, 2]).max()))
print('walk time (end - flags): median %.1f max %.1f us' % (np.median(us(b[:, 2]) - us(b[:, 1])), (us(b[:, 2]) - us(b[:, 1])).max()))
