#!/usr/bin/env python3
"""Diagnostic (ABN_STAMPS build: tools/build_stamps.sh): phase timeline of the resident BatchNorm forward
(csrc/tower_bn_persist.h), workgroup medians of s_memtime ticks (10 ns each)."""
import os, sys
os.environ.setdefault('ABNET3_HIP_LIB', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'variants', 'lib_stamps.so'))   # tools/build_stamps.sh
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
buf = torch.zeros(1024 * 128, dtype=torch.int64, device='cuda')
os.environ['ABN_STAMP_BUF'] = str(buf.data_ptr())
import bench
from abnet3_amd.model import SiameseNetwork
torch.manual_seed(0)
net = SiameseNetwork(output_path='/tmp/abn_st', **dict(bench.C2, batch_norm=True)).cuda()
x1, x2 = torch.randn(4096, 40, device='cuda'), torch.randn(4096, 40, device='cuda')
net.train()
for _ in range(10):
    net.direct_forward(x1, x2)
torch.cuda.synchronize()
nl = 4
n = 2 + 8 * nl - 1          # (the last layer returns before its slot 8)
s = buf.cpu().numpy().reshape(1024, 128)[:256, :n].astype(np.float64)
s -= s[:, 0].min()
names = ['input staging']
for l in range(nl):
    names += ['L%d k-loop' % l, 'L%d z, sums, stores' % l, 'L%d grid barrier 1' % l, 'L%d finish' % l, 'L%d grid barrier 2' % l,
              'L%d stats load + normalise + act' % l, 'L%d scales, image, transposed image' % l, 'L%d (gap to next)' % l]
d = np.diff(s, axis=1)
for i in range(n - 1):
    print('%-40s median %8.0f   min %8.0f   max %8.0f' % (names[i], np.median(d[:, i]), d[:, i].min(), d[:, i].max()))
print('wg start spread: median %.0f max %.0f; wg total median %.0f; last end %.0f (ticks of 10 ns)' % (np.median(s[:, 0]), s[:, 0].max(), np.median(s[:, n - 1] - s[:, 0]), s[:, n - 1].max()))
# who arrives late?  slot 9 + 8 l holds s_memrealtime (the chip-wide 100 MHz clock: 10 ns a tick) at the moment a workgroup has
# published its column sums
full = buf.cpu().numpy().reshape(1024, 128)[:256].astype(np.float64)
for l in (0, 1, 2, 3):
    t = full[:, 9 + 8 * l]
    late = (t - np.median(t)) * 10.0                     # ns against the median workgroup
    print('L%d sums published, ns against the median workgroup: p10 %.0f p90 %.0f max %.0f' % (l, np.percentile(late, 10), np.percentile(late, 90), late.max()))
    print('   median by blockIdx %% 8: ' + ' '.join('%6.0f' % np.median(late[x::8]) for x in range(8)))
    print('   median by quarter of the grid (calls 0, 0, 1, 1): ' + ' '.join('%6.0f' % np.median(late[q * 64:(q + 1) * 64]) for q in range(4)))
    worst = np.argsort(-late)[:10]
    print('   the latest: ' + ' '.join('%d(%+.0f)' % (w, late[w]) for w in worst))
# layer 1, the first 16 finishing workgroups: when each of their 512 terms was seen (chip-wide clock) against when its producer
# published it -- the hand-over's latency, pair by pair (term i = (feature j, call g, producer k): i = (j * 2 + g) * 128 + k)
allb = buf.cpu().numpy().reshape(1024, 128).astype(np.float64)
pub = allb[:256, 9 + 8 * 1]
lat = []
for b in range(16):
    seen = allb[256 + 4 * b:256 + 4 * b + 4].reshape(512)
    for i in range(512):
        k, g = i % 128, (i // 128) % 2
        w = g * 128 + k
        if seen[i] > 0:
            lat.append((seen[i] - pub[w]) * 10.0)
lat = np.array(lat)
own = allb[:16, 9 + 8 * 1]
print('hand-over latency (ns, seen - published), %d pairs: p10 %.0f median %.0f p90 %.0f max %.0f' % (lat.size, np.percentile(lat, 10), np.median(lat), np.percentile(lat, 90), lat.max()))
for b in range(3):
    seen = allb[256 + 4 * b:256 + 4 * b + 4].reshape(512)
    print('   finishing workgroup %d: its own publish at 0, terms seen at p10 %.0f median %.0f max %.0f ns; the latest producer published at %+.0f' % (
        b, (np.percentile(seen, 10) - own[b]) * 10, (np.median(seen) - own[b]) * 10, (seen.max() - own[b]) * 10, (pub.max() - own[b]) * 10))
