#!/usr/bin/env python3
"""Diagnostic: per-phase s_memtime shares of the GEMM k-loop.  Needs the
library built with -DABN_STAMPS (tools/build_stamps.sh); never used in timing."""
import os, sys
os.environ.setdefault('ABNET3_HIP_LIB', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'variants', 'lib_stamps.so'))   # tools/build_stamps.sh
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
buf = torch.zeros(4096 * 128, dtype=torch.int64, device='cuda')
os.environ['ABN_STAMP_BUF'] = str(buf.data_ptr())
from abnet3_amd import _lib
lib = _lib.load()
rows, k, n = 8192, 500, 500
x = torch.randn(rows, k, device='cuda'); w = torch.randn(n, k, device='cuda') * .05
b = torch.zeros(n, device='cuda'); y = torch.empty(rows, n, device='cuda')
for _ in range(20):
    _lib.check(lib.abn_linear_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), rows, k, n, 1, _lib.ptr(y), _lib.stream()), 'f')
torch.cuda.synchronize()
s = buf.cpu().numpy().reshape(4096, 128)
nblk = int((s[:, 0] != 0).sum())
s = s[:nblk].astype(np.float64)
nst = int((s[0] != 0).sum())
t00 = s[:, 0].min()
s = s - t00
print('blocks', nblk, 'stamps per block', nst)
d = np.diff(s[:, :nst], axis=1)
names = ['prologue'] + ['issue', 'mfma', 'commit', 'barrier'] * 16 + ['stage+sync', 'store']
for grp, sel in (('blocks 0..255', slice(0, 256)), ('blocks 256..511', slice(256, 512))):
    agg = {}
    for i in range(d.shape[1]):
        nm = names[i] if i < len(names) else 'x%d' % i
        agg.setdefault(nm, []).append(np.median(d[sel, i]))
    tot = np.median(s[sel, nst - 1] - s[sel, 0])
    print(grp, 'start median %.0f end median %.0f total %.0f' % (np.median(s[sel, 0]), np.median(s[sel, nst - 1]), tot))
    for nm, v in agg.items():
        print('   %-12s per occurrence %8.0f cycles  x%2d = %8.0f  (%.1f%%)' % (nm, np.mean(v), len(v), np.sum(v), 100 * np.sum(v) / tot))
for b in (0, 1, 256, 257):
    per_tile = [s[b, 1 + 4 * (t + 1)] - s[b, 1 + 4 * t] for t in range(16)]
    mf = [d[b, 2 + 4 * t] for t in range(16)]
    print('block %3d start %7.0f tile periods %s' % (b, s[b, 0], ' '.join('%5.0f' % v for v in per_tile)))
    print('              mfma phases  %s' % ' '.join('%5.0f' % v for v in mf))
print('kernel span (first start .. last end): %.0f cycles' % s[:, nst - 1].max())
