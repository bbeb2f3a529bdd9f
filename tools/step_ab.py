"""A/B of the C2 train step under different environment settings, all on the SAME box
(one child process per setting, interleaved twice).   python tools/step_ab.py "A=1" "B=0 C=2" ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch, bench
from abnet3_amd.model import SiameseNetwork
from abnet3_amd.loss import coscos2
from abnet3_amd.trainer import TrainerSiamese
torch.manual_seed(0)
net = SiameseNetwork(output_path='/tmp/abn_ab', **bench.C2)
if os.environ.get('ABN_PRECISION'): net.precision = os.environ['ABN_PRECISION']
tr = TrainerSiamese(network=net, loss=coscos2(avg=False), optimizer_type='adadelta', lr=0.1, dataloader=None, log_dir='/tmp/abn_runs')
pool = bench.make_pool(seed=0, device=torch.device('cuda'))
net.train()
for i in range(300): tr.train_step(pool[i %% 8], True)
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(400): tr.train_step(pool[i %% 8], True)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 400)
print('%%.4f' %% (best * 1e3))
''' % ROOT
settings = sys.argv[1:] or ['']
res = {s: [] for s in settings}
for rep in range(2):
    for s in settings:
        env = dict(os.environ)
        for kv in s.split():
            k, v = kv.split('=', 1)
            env[k] = v
        out = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True)
        res[s].append(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-200:])
for s in settings:
    print('%-40s ms/step: %s' % (s or '(default)', ' '.join(res[s])), flush=True)
