#!/usr/bin/env python3
"""fwd GEMM at K=N=500 (2000-B rows, misaligned to 128-B lines) vs 512 (aligned)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from abnet3_amd import _lib
lib = _lib.load()
rows = 8192
def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps
for k, n in [(500, 500), (512, 512), (504, 504), (480, 480), (512, 500), (500, 512)]:
    x = torch.randn(rows, k, device='cuda'); w = torch.randn(n, k, device='cuda') * .05
    b = torch.zeros(n, device='cuda'); y = torch.empty(rows, n, device='cuda')
    t = timeit(lambda: _lib.check(lib.abn_linear_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), rows, k, n, 1, _lib.ptr(y), _lib.stream()), 'f'))
    print('fwd K=%3d N=%3d  %7.1f us  %6.1f TF' % (k, n, t * 1e6, 2.0 * rows * k * n / t / 1e12))
