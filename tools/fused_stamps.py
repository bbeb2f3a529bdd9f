#!/usr/bin/env python3
"""Diagnostic (ABN_STAMPS build): per-layer cycle shares of the fused tower forward."""
import os, sys
os.environ.setdefault('ABNET3_HIP_LIB', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'variants', 'lib_stamps.so'))   # tools/build_stamps.sh
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
buf = torch.zeros(1024 * 128, dtype=torch.int64, device='cuda')
os.environ['ABN_STAMP_BUF'] = str(buf.data_ptr())
from abnet3_amd.model import SiameseNetwork
torch.manual_seed(0)
net = SiameseNetwork(input_dim=40, num_hidden_layers=2, hidden_dim=500, output_dim=100, p_dropout=0.0,
                     activation_layer='sigmoid').cuda()
x = torch.randn(8192, 40, device='cuda')
net.train()
with torch.no_grad():
    for _ in range(10):
        net.forward_pair_rows(x)
torch.cuda.synchronize()
s = buf.cpu().numpy().reshape(1024, 128)[:256, :14].astype(np.float64)
s -= s[:, :1].min()
names = ['input'] + sum([['L%d pre' % l, 'L%d kloop' % l, 'L%d epilogue' % l] for l in range(4)], [])
d = np.diff(s, axis=1)
for i in range(13):
    print('%-14s median %8.0f cycles' % (names[i] if i < len(names) else 'store+end', np.median(d[:, i])))
print('total median %.0f cycles; last end - first start %.0f' % (np.median(s[:, 13] - s[:, 0]), s[:, 13].max()))
