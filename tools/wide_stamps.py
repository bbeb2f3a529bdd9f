#!/usr/bin/env python3
"""Diagnostic (ABN_STAMPS build: tools/variants.sh tower "-DABN_STAMPS", ABNET3_HIP_LIB=tools/variants/lib_ABN_STAMPS.so):
phase timeline of the layer-per-launch forward kernels (workgroup medians, s_memtime ticks of 10 ns)."""
import os, sys
os.environ.setdefault('ABNET3_HIP_LIB', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'variants', 'lib_stamps.so'))   # tools/build_stamps.sh
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
buf = torch.zeros(4 * 1024 * 16, dtype=torch.int64, device='cuda')
os.environ['ABN_STAMP_BUF'] = str(buf.data_ptr())
from abnet3_amd.model import SiameseNetwork
B = int(os.environ.get('PAIRS', 485))
torch.manual_seed(0)
net = SiameseNetwork(input_dim=280, num_hidden_layers=2, hidden_dim=500, output_dim=100, p_dropout=0.0, activation_layer='sigmoid').cuda()
x1, x2 = torch.randn(B, 280, device='cuda'), torch.randn(B, 280, device='cuda')
net.train()
for _ in range(10):
    net.direct_forward(x1, x2)
torch.cuda.synchronize()
s = buf.cpu().numpy().reshape(4, 1024, 16).astype(np.float64)
names = ['ring issue + staging', 'barrier', 'k-loop', 'park + barrier', 'epilogue + stores']
for l in range(4):
    n = int((s[l, :, 0] > 0).sum())
    t = s[l, :n, :6]
    t0 = t[:, 0].min()
    d = np.diff(t, axis=1)
    print('layer %d: %d workgroups, start spread median %.0f max %.0f, wg total median %.0f, last end %.0f (x 10 ns)'
          % (l, n, np.median(t[:, 0] - t0), (t[:, 0] - t0).max(), np.median(t[:, 5] - t[:, 0]), (t[:, 5] - t0).max()))
    print('    ' + '   '.join('%s %.0f/%.0f' % (names[i], np.median(d[:, i]), d[:, i].max()) for i in range(5)))
if l < 3:
    pass
