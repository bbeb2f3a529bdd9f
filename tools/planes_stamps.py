#!/usr/bin/env python3
"""Diagnostic (ABN_STAMPS build, tools/variants.sh tower "-DABN_STAMPS"): phase timeline of the planes tower forward
(workgroup medians, s_memtime ticks of 10 ns)."""
import os, sys
os.environ.setdefault('ABNET3_HIP_LIB', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'variants', 'lib_stamps.so'))   # tools/build_stamps.sh
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
buf = torch.zeros(1024 * 128, dtype=torch.int64, device='cuda')
os.environ['ABN_STAMP_BUF'] = str(buf.data_ptr())
import bench
from abnet3_amd.model import SiameseNetwork
torch.manual_seed(0)
net = SiameseNetwork(output_path='/tmp/abn_st', **bench.C2).cuda()
x1, x2 = torch.randn(4096, 40, device='cuda'), torch.randn(4096, 40, device='cuda')
net.train()
for _ in range(10):
    net.direct_forward(x1, x2)
torch.cuda.synchronize()
nl = 4
n = 3 + 5 * nl
s = buf.cpu().numpy().reshape(1024, 128)[:256, :n].astype(np.float64)
t0 = s[:, 0].min()
s -= t0
names = ['start', 'input staged', 'barrier'] 
names = ['input staging', 'barrier']
for l in range(nl):
    names += ['L%d k-loop' % l, 'L%d finish (bias/act)' % l, 'L%d barrier A' % l, 'L%d img+out writes' % l, 'L%d pad+barrier B' % l]
d = np.diff(s, axis=1)
for i in range(n - 1):
    print('%-24s median %8.0f   max %8.0f' % (names[i], np.median(d[:, i]), d[:, i].max()))
print('wg start spread: median %.0f max %.0f; wg total median %.0f; last end %.0f' % (np.median(s[:, 0]), s[:, 0].max(), np.median(s[:, n - 1] - s[:, 0]), s[:, n - 1].max()))
