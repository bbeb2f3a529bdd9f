"""Times abn_fbank on 600 s of synthetic 16 kHz audio (bench.py's filterbank leg)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from abnet3_amd.features import FeaturesGenerator
secs = int(sys.argv[1]) if len(sys.argv) > 1 else 600
fs = 16000
rng = np.random.default_rng(7)
n = secs * fs
t = np.arange(n) / fs
sig = (2000 * np.sin(2 * np.pi * 440 * t) + 500 * rng.standard_normal(n)).astype(np.int16)
fg = FeaturesGenerator()
d = torch.from_numpy(sig).cuda()
out = fg.fbank_from_samples(d, fs); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): out = fg.fbank_from_samples(d, fs)
e1.record(); torch.cuda.synchronize()
dt = e0.elapsed_time(e1) * 1e-3 / 20
print('fbank %d frames: %.3f ms = %.3e frames/s' % (out.shape[0], dt * 1e3, out.shape[0] / dt))
