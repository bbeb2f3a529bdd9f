"""C2 step and its three launches for the library in ABNET3_HIP_LIB (tools/variants.sh): ms/step and the live
per-launch times bench.py reports."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from abnet3_amd.model import SiameseNetwork
from abnet3_amd.loss import coscos2
from abnet3_amd.trainer import TrainerSiamese
import time
torch.manual_seed(0)
net = SiameseNetwork(output_path='/tmp/abn_ab', **bench.C2).cuda()
tr = TrainerSiamese(network=net, loss=coscos2(avg=False), optimizer_type='adadelta', lr=0.1, dataloader=None, log_dir='/tmp/abn_runs')
pool = bench.make_pool(seed=0, device=torch.device('cuda'))
net.train()
for i in range(150): tr.train_step(pool[i % 8], True)
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(300): tr.train_step(pool[i % 8], True)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 300)
r = bench.planes_roofline(torch, net)
parts = {r['dominant']: r['avg_launch_us']}
for k in ('weight_gradients', 'dgrad_chain', 'forward'):
    if k in r: parts[k] = r[k]['avg_launch_us']
print('%-22s %.4f ms/step  %s' % (os.path.basename(os.environ.get('ABNET3_HIP_LIB', 'default')), best * 1e3, json.dumps(parts)), flush=True)
