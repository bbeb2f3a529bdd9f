"""us per abn_tower_forward at C2 (8192 rows, 40-500-500-100 sigmoid, no BN); ABNET3_PRECISION picks the arithmetic,
ABNET3_HIP_LIB a variant library (tools/variants.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from abnet3_amd.model import SiameseNetwork
torch.manual_seed(0)
net = SiameseNetwork(output_path='/tmp/abn_fwd', **bench.C2).cuda()
net.train()
rows = int(os.environ.get('ROWS', 4096))
x1, x2 = torch.randn(rows, 40, device='cuda'), torch.randn(rows, 40, device='cuda')
for _ in range(30): net.direct_forward(x1, x2)
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300): net.direct_forward(x1, x2)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 300 * 1e3)
print('%-28s %-7s %.2f us' % (os.path.basename(os.environ.get('ABNET3_HIP_LIB', 'default')), net.precision, best), flush=True)
