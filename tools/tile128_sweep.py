"""Step time of the C2 tower at several batch sizes with the weight gradients on 128 x 128 tiles (ABN_WGRAD_TILE128=1)
and on the wide tiles (=0): where the default's threshold (4096 tower rows) sits.  One child process per setting."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch, bench
from abnet3_amd.model import SiameseNetwork
from abnet3_amd.loss import coscos2
from abnet3_amd.trainer import TrainerSiamese
B = int(os.environ['PAIRS'])
torch.manual_seed(0)
net = SiameseNetwork(output_path='/tmp/abn_ab', **bench.C2)
tr = TrainerSiamese(network=net, loss=coscos2(avg=False), optimizer_type='adadelta', lr=0.1, dataloader=None, log_dir='/tmp/abn_runs')
g = torch.Generator(device='cuda'); g.manual_seed(0)
pool = [(torch.randn(B, 40, device='cuda', generator=g), torch.randn(B, 40, device='cuda', generator=g),
         (torch.randint(0, 2, (B,), device='cuda', generator=g) * 2 - 1).double()) for _ in range(4)]
net.train()
for i in range(200): tr.train_step(pool[i %% 4], True)
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(300): tr.train_step(pool[i %% 4], True)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 300)
print('%%.4f' %% (best * 1e3))
''' % ROOT
for pairs in (512, 1024, 1536, 2048, 3072):
    row = []
    for v in ('0', '1'):
        env = dict(os.environ, PAIRS=str(pairs), ABN_WGRAD_TILE128=v)
        out = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True)
        row.append(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-200:])
    print('pairs %5d (rows %5d): wide tiles %s ms, 128 x 128 %s ms' % (pairs, 2 * pairs, row[0], row[1]), flush=True)
