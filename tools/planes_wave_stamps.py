#!/usr/bin/env python3
"""Diagnostic (ABN_STAMPS build: tools/variants.sh tower "-DABN_STAMPS" [...]): when each of a workgroup's eight waves leaves the
k-loop of every layer of the C2 forward chain, against the moment wave 0 entered it (s_memtime ticks, medians over the
256 workgroups): the spread between the first and the last wave is what the layer's barrier makes everybody wait."""
import os, sys
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
s = buf.cpu().numpy().reshape(1024, 128)[:256].astype(np.float64)
print(os.environ.get('ABNET3_HIP_LIB', 'default library'))
for l in range(4):
    enter = s[:, 2 + 5 * l]                       # wave 0 at the layer's start
    w = s[:, 64 + 8 * l:64 + 8 * l + 8] - enter[:, None]
    act = w[:, (w > 0).all(axis=0)] if (w > 0).any() else w
    srt = np.sort(np.where(w > 0, w, np.nan), axis=1)
    first, last = np.nanmin(srt, axis=1), np.nanmax(srt, axis=1)
    barrier = s[:, 5 + 5 * l] - enter            # everybody is behind barrier A
    print('layer %d: first wave out of the k-loop %6.0f, last %6.0f (median over workgroups), spread %5.0f; behind barrier A at %6.0f; per wave %s'
          % (l, np.median(first), np.median(last), np.median(last - first), np.median(barrier),
             ' '.join('%5.0f' % np.nanmedian(np.where(w[:, i] > 0, w[:, i], np.nan)) for i in range(8))))
