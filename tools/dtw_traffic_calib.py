"""Calibration of FETCH_SIZE for the DTW kernel's access pattern (MI355X_MICROARCH.md, HBM: "other access widths are
uncalibrated: calibrate on a known byte count in your own access pattern"): P pairs whose first token has exactly 32
rows -- ONE band, so every row of both tokens is fetched exactly once, by the same per-lane 16-byte row-piece loads as
in C4 -- and a second token of 320 rows.  Known bytes: P * (32 + 320) * 160 (1.13 GB at P = 20 000: past the Infinity
Cache).  Run under   tools/pmc.sh calib FETCH_SIZE tools/dtw_traffic_calib.py   and compare."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from abnet3_amd.utils import dtw_align_batch
P = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = np.random.default_rng(0)
n1 = np.full(P, 32, dtype=np.int32); n2 = np.full(P, 320, dtype=np.int32)
o1 = np.arange(P, dtype=np.int64) * 32; o2 = np.arange(P, dtype=np.int64) * 320
f1 = torch.from_numpy(rng.standard_normal((P * 32, 40), dtype=np.float32)).cuda()
f2 = torch.from_numpy(rng.standard_normal((P * 320, 40), dtype=np.float32)).cuda()
for _ in range(3):
    dtw_align_batch(f1, o1, n1, f2, o2, n2)
torch.cuda.synchronize()
print('known feature bytes per call: %d' % (P * (32 + 320) * 160))
