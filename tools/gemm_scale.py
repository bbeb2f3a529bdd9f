#!/usr/bin/env python3
"""Forward GEMM rate across shapes: separates the kernel's steady-state inner
loop (large shapes) from the fixed costs that dominate the C2 layer shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from abnet3_amd import _lib
from gemm_bench import timeit  # noqa

lib = _lib.load()
for rows, k, n in [(8192, 500, 500), (8192, 512, 512), (32768, 512, 512), (131072, 512, 512),
                   (8192, 2048, 512), (8192, 2048, 2048), (16384, 4096, 4096)]:
    x = torch.randn(rows, k, device='cuda'); w = torch.randn(n, k, device='cuda') * .05
    b = torch.zeros(n, device='cuda'); y = torch.empty(rows, n, device='cuda')
    fl = 2.0 * rows * k * n
    t = timeit(lambda: _lib.check(lib.abn_linear_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), rows, k, n, 1, _lib.ptr(y), _lib.stream()), 'f'), reps=10)
    t2 = timeit(lambda: torch.mm(x, w.t(), out=y), reps=10)
    print('fwd rows=%6d K=%4d N=%4d  %8.1f us %6.1f TF   | torch.mm %8.1f us %6.1f TF' % (rows, k, n, t * 1e6, fl / t / 1e12, t2 * 1e6, fl / t2 / 1e12), flush=True)
