"""C2 train step time against the batch size, planes kernels (forced) vs per-layer GEMM path: where the library's
row threshold (ABN_FUSED_MIN_ROWS, default in csrc/tower.hip planes_path) should sit."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch, bench
from abnet3_amd.model import SiameseNetwork
from abnet3_amd.loss import coscos2
from abnet3_amd.trainer import TrainerSiamese
B = int(sys.argv[1])
torch.manual_seed(0)
net = SiameseNetwork(output_path='/tmp/abn_sw', **bench.C2)
if os.environ.get('ABN_PRECISION'): net.precision = os.environ['ABN_PRECISION']
tr = TrainerSiamese(network=net, loss=coscos2(avg=False), optimizer_type='adadelta', lr=0.1, dataloader=None, log_dir='/tmp/abn_runs')
pool = [(torch.randn(B, 40, device='cuda'), torch.randn(B, 40, device='cuda'), (torch.rand(B, device='cuda') > 0.5).float() * 2 - 1) for _ in range(4)]
net.train()
for i in range(100): tr.train_step(pool[i %% 4], True)
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(200): tr.train_step(pool[i %% 4], True)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 200)
print('%%.4f' %% (best * 1e3))
''' % ROOT
for B in (256, 512, 1024, 1536, 2048, 3072, 4096, 8192):
    row = []
    for mr in ('0', '1000000000'):
        env = dict(os.environ, ABN_FUSED_MIN_ROWS=mr)
        out = subprocess.run([sys.executable, '-c', CHILD, str(B)], env=env, capture_output=True, text=True)
        row.append(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
    print('pairs %5d (rows %5d): planes %s ms   per-layer %s ms' % (B, 2 * B, row[0], row[1]), flush=True)
