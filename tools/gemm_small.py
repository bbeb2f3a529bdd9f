"""Forward / dgrad GEMM latency at small row counts (latency-bound: one workgroup per CU or fewer)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from abnet3_amd import _lib
from gemm_bench import timeit  # noqa  (prints its own table first)
lib = _lib.load()
for rows in (200, 1000, 2000, 4096):
    k = n = 500
    x = torch.randn(rows, k, device='cuda'); w = torch.randn(n, k, device='cuda') * .05
    b = torch.zeros(n, device='cuda'); y = torch.empty(rows, n, device='cuda')
    t = timeit(lambda: _lib.check(lib.abn_linear_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), rows, k, n, 1, _lib.ptr(y), _lib.stream()), 'f'), reps=50)
    t2 = timeit(lambda: torch.mm(x, w.t(), out=y), reps=50)
    print('SMALL fwd rows=%5d 500x500  %6.1f us   (torch.mm %6.1f us)' % (rows, t * 1e6, t2 * 1e6), flush=True)
