#!/usr/bin/env python3
"""Launches only the fwd 8192x500x500 GEMM (for PMC collection)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from abnet3_amd import _lib
lib = _lib.load()
rows, k, n = 8192, 500, 500
x = torch.randn(rows, k, device='cuda'); w = torch.randn(n, k, device='cuda') * .05
b = torch.zeros(n, device='cuda'); y = torch.empty(rows, n, device='cuda')
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    _lib.check(lib.abn_linear_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), rows, k, n, 1, _lib.ptr(y), _lib.stream()), 'f')
torch.cuda.synchronize()
