#!/usr/bin/env python3
"""Times the three tower GEMMs at the C2 shapes through the single-layer C-ABI
entries (HIP events on the launch stream).  Tuning aid for gemm_f32.h."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from abnet3_amd import _lib

lib = _lib.load()
rows = 8192


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


tot = 0.0
for (k, n) in [(40, 500), (500, 500), (500, 100)]:
    x = torch.randn(rows, k, device='cuda'); w = torch.randn(n, k, device='cuda') * .05
    b = torch.zeros(n, device='cuda'); y = torch.empty(rows, n, device='cuda')
    dz = torch.randn(rows, n, device='cuda'); dx = torch.empty(rows, k, device='cuda')
    dW = torch.empty(n, k, device='cuda'); db = torch.empty(n, device='cuda')
    sc_n = lib.abn_linear_wgrad_scratch_floats(rows, k, n)
    sc = torch.empty(sc_n, device='cuda')
    fl = 2.0 * rows * k * n
    t = timeit(lambda: _lib.check(lib.abn_linear_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), rows, k, n, 1, _lib.ptr(y), _lib.stream()), 'f'))
    print('fwd   K=%3d N=%3d  %7.1f us  %6.1f TF' % (k, n, t * 1e6, fl / t / 1e12)); tot += t * (2 if (k, n) == (500, 500) else 1)
    t = timeit(lambda: _lib.check(lib.abn_linear_dgrad(_lib.ptr(dz), _lib.ptr(w), rows, k, n, _lib.ptr(x), 1, _lib.ptr(dx), _lib.stream()), 'd'))
    print('dgrad K=%3d N=%3d  %7.1f us  %6.1f TF' % (k, n, t * 1e6, fl / t / 1e12)); tot += t * (2 if (k, n) == (500, 500) else 1) * (0 if k == 40 else 1)
    t = timeit(lambda: _lib.check(lib.abn_linear_wgrad(_lib.ptr(dz), _lib.ptr(x), rows, k, n, _lib.ptr(dW), _lib.ptr(db), _lib.ptr(sc), sc_n, _lib.stream()), 'w'))
    print('wgrad K=%3d N=%3d  %7.1f us  %6.1f TF (incl. slab reduce)' % (k, n, t * 1e6, fl / t / 1e12)); tot += t * (2 if (k, n) == (500, 500) else 1)
print('sum over the 11 GEMMs of one C2 step: %.1f us  -> %.1f TF' % (tot * 1e6, 27.7e9 / tot / 1e12))
