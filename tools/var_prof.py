"""Kernel time of the 128x64 backward pair grid under each variant library in tools/variants/ (bf16x3)."""
import glob, os, subprocess, sys, csv, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch
from abnet3_amd import _lib
lib = _lib.load()
rows, k, n = 8192, 500, 500
dz, a = torch.randn(rows, n, device='cuda'), torch.rand(rows, k, device='cuda')
W = torch.randn(n, k, device='cuda') * 0.05
dW, db, dx = torch.empty(n, k, device='cuda'), torch.empty(n, device='cuda'), torch.empty(rows, k, device='cuda')
sc_n = lib.abn_linear_wgrad_scratch_floats(rows, k, n); sc = torch.empty(sc_n, device='cuda')
import ctypes
def run():
    _lib.check(lib.abn_linear_backward(_lib.ptr(dz), _lib.ptr(W), _lib.ptr(a), rows, k, n, 1, _lib.ptr(dW), _lib.ptr(db), _lib.ptr(dx), _lib.ptr(sc), sc_n, _lib.stream()), 'bwd')
for _ in range(20): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): run()
e1.record(); torch.cuda.synchronize()
print('%%.2f' %% (e0.elapsed_time(e1) * 10))
''' % root
for lib in sorted(glob.glob(os.path.join(root, 'tools', 'variants', 'lib_*.so'))):
    env = dict(os.environ, ABNET3_HIP_LIB=lib)
    r = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True)
    print('%-60s %s us per linear_backward (pair grid + slab reduce)' % (os.path.basename(lib), r.stdout.strip() or r.stderr[-300:]), flush=True)
