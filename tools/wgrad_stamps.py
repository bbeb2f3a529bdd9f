#!/usr/bin/env python3
"""Diagnostic (tools/variants.sh tower "-DABN_STAMPS", ABNET3_HIP_LIB=tools/variants/lib_ABN_STAMPS.so): where a
workgroup of the weight-gradient launch spends its time -- prologue / row-step loop / slab stores -- per layer, and
which workgroups share a CU (s_memtime ticks of 10 ns)."""
import os, sys
os.environ.setdefault('ABNET3_HIP_LIB', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'variants', 'lib_stamps.so'))   # tools/build_stamps.sh
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
buf = torch.zeros(4096 * 16, dtype=torch.int64, device='cuda')
os.environ['ABN_WSTAMP_BUF'] = str(buf.data_ptr())
import bench
from abnet3_amd.model import SiameseNetwork
from abnet3_amd.loss import coscos2
from abnet3_amd.trainer import TrainerSiamese
torch.manual_seed(0)
net = SiameseNetwork(output_path='/tmp/abn_st', **bench.C2).cuda()
tr = TrainerSiamese(network=net, loss=coscos2(avg=False), optimizer_type='adadelta', lr=0.1, dataloader=None, log_dir='/tmp/abn_runs')
pool = bench.make_pool(seed=0, device=torch.device('cuda'))
net.train()
for i in range(20): tr.train_step(pool[i % 8], True)
torch.cuda.synchronize()
r = bench.planes_roofline(torch, net)
print('weight-gradient launch by events: %.1f us' % r['weight_gradients']['avg_launch_us'])
buf.zero_()
torch.cuda.synchronize()
tr.train_step(pool[0], True)       # the ONE launch the stamps below are of
torch.cuda.synchronize()
s = buf.cpu().numpy().reshape(4096, 16)
s = s[s[:, 0] > 0]
t = s[:, :4].astype(np.float64)
# (every XCD counts its own time: all times relative to the XCD's first start)
for x in np.unique(s[:, 6]):
    m = s[:, 6] == x
    t[m] -= t[m, 0].min()
print('%d workgroups; launch span per XCD (ticks): %s' % (len(s), [int(t[s[:, 6] == x, 3].max()) for x in np.unique(s[:, 6])]))
for l in np.unique(s[:, 4]):
    m = s[:, 4] == l
    d = np.diff(t[m], axis=1)
    print('layer N*1024+K=%d: %3d workgroups of %3d steps: start median %6.0f max %6.0f | prologue %5.0f  loop %6.0f (%.1f / step)  stores %5.0f | end max %6.0f' % (
        l, m.sum(), s[m, 5][0], np.median(t[m, 0]), t[m, 0].max(), np.median(d[:, 0]), np.median(d[:, 1]), np.median(d[:, 1]) / s[m, 5][0], np.median(d[:, 2]), t[m, 3].max()))
# who shares a CU: HW_ID bits: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (gfx9)
hw = s[:, 7]
cu = (s[:, 6] << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
u, cnt = np.unique(cu, return_counts=True)
print('%d distinct CUs; workgroups per CU: %s' % (len(u), dict(zip(*np.unique(cnt, return_counts=True)))))
kinds = {}
for c in u:
    k = tuple(sorted(s[cu == c, 5]))
    kinds[k] = kinds.get(k, 0) + 1
for k, v in sorted(kinds.items(), key=lambda kv: -kv[1])[:12]:
    print('  CUs with workgroups of steps %s: %d' % (list(map(int, k)), v))
# per CU: the light workgroup relative to the heavy one it shares the CU with
rel = []
for c in u:
    m = np.where(cu == c)[0]
    if len(m) == 2:
        h, l = (m[0], m[1]) if s[m[0], 5] > s[m[1], 5] else (m[1], m[0])
        rel.append((s[l, 0] - s[h, 0], s[l, 3] - s[h, 0], s[h, 3] - s[h, 0], s[h, 1] - s[h, 0], s[h, 2] - s[h, 0]))
rel = np.array(rel, dtype=np.float64)
print('per CU, from the heavy workgroup start: light starts %+.0f (median), light ends %.0f, heavy loop begins %.0f, heavy loop ends %.0f, heavy ends %.0f' % (
    np.median(rel[:, 0]), np.median(rel[:, 1]), np.median(rel[:, 3]), np.median(rel[:, 4]), np.median(rel[:, 2])))
hv = s[s[:, 5] == 64]
q = np.stack([hv[:, 1], hv[:, 8], hv[:, 9], hv[:, 10], hv[:, 2]], 1).astype(np.float64)
print('heavy workgroups, ticks per step in steps 0-15, 16-31, 32-47, 48-63 (median):', [round(float(v) / 16, 1) for v in np.median(np.diff(q, axis=1), axis=0)])
