"""Phases of the small-batch weight-gradient + step launch (csrc/tower_wgrad_step.h), on the chip-wide 100 MHz clock:
start, scales agreed, sum done, partial sums parked, parameters written (library built with -DABN_WGS_STAMPS:
tools/variants.sh tower "-DABN_WGS_STAMPS"; ABNET3_HIP_LIB=tools/variants/lib_ABN_WGS_STAMPS.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
os.environ.setdefault('PAIRS', '485')
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import small_batch_probe as sb
from abnet3_amd import _lib
net, tr, pool = sb.make(485)
for i in range(20):
    tr.train_step(pool[i % 4], True)
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros((512, 8), dtype=np.uint64)
raw.abn_debug_wgs_stamps(buf.ctypes.data_as(ctypes.c_void_p))
b = buf[buf[:, 0] > 0].astype(np.float64)
t0 = b[:, 0].min()
print('%d workgroups stamped' % len(b))
for k, name in enumerate(['start', 'scales agreed', 'sum done (wave 0)', 'parked + barrier', 'parameters written']):
    col = b[:, k][b[:, k] > 0]
    print('  %-22s min %6.2f  median %6.2f  max %6.2f us' % (name, (col.min() - t0) / 100, (np.median(col) - t0) / 100, (col.max() - t0) / 100))
