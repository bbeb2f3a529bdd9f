"""us per abn_linear_backward (8192 x 500 x 500: the 128x64 pair grid + slab reduce); precision from ABN_LINEAR_PREC."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from abnet3_amd import _lib
lib = _lib.load()
rows, k, n = 8192, 500, 500
dz, a = torch.randn(rows, n, device='cuda'), torch.rand(rows, k, device='cuda')
W = torch.randn(n, k, device='cuda') * 0.05
dW, db, dx = torch.empty(n, k, device='cuda'), torch.empty(n, device='cuda'), torch.empty(rows, k, device='cuda')
sc_n = lib.abn_linear_wgrad_scratch_floats(rows, k, n); sc = torch.empty(sc_n, device='cuda')
def run():
    _lib.check(lib.abn_linear_backward_prec(_lib.ptr(dz), _lib.ptr(W), _lib.ptr(a), rows, k, n, 1, int(os.environ.get('ABN_LINEAR_PREC', '0')), _lib.ptr(dW), _lib.ptr(db), _lib.ptr(dx), _lib.ptr(sc), sc_n, _lib.stream()), 'bwd')
for _ in range(20): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): run()
e1.record(); torch.cuda.synchronize()
print('prec %s stagger %s: %.2f us' % (os.environ.get('ABN_LINEAR_PREC', '0'), os.environ.get('ABN_GEMM_STAGGER', '0'), e0.elapsed_time(e1) * 5))
