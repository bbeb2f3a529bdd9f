"""Stability soak: many train steps (eager and graph), DTW calls of changing sizes, embed
calls; checks finite losses, bit-stable DTW results and flat memory."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
import abnet3_amd.loss as L
from abnet3_amd.model import SiameseNetwork
from abnet3_amd.trainer import TrainerSiamese
from abnet3_amd.utils import dtw_align_batch
torch.manual_seed(0)
net = SiameseNetwork(output_path='/tmp/abn_soak', **bench.C2)
tr = TrainerSiamese(network=net, loss=L.coscos2(avg=False), optimizer_type='adadelta', lr=0.1, dataloader=None, log_dir='/tmp/abn_runs')
pool = bench.make_pool(seed=0, device=torch.device('cuda'))
net.train()
t0 = time.perf_counter()
for i in range(30000):
    loss = tr.train_step(pool[i % len(pool)], True)
    if i % 5000 == 0:
        print('step %6d loss %.4f reserved %.2f GB' % (i, float(loss), torch.cuda.memory_reserved() / 1e9), flush=True)
assert np.isfinite(float(loss))
step = tr.make_graphed_step(pool[0])
for i in range(20000):
    loss = step(pool[i % len(pool)])
print('graph steps done, loss %.4f, %.1f s' % (float(loss), time.perf_counter() - t0), flush=True)
rng = np.random.default_rng(0)
ref = None
for it in range(150):
    P = int(rng.integers(1, 3000))
    f1, o1, n1, f2, o2, n2 = bench.synth_dtw_pairs(P, seed=it % 3)
    res = dtw_align_batch(torch.from_numpy(f1).cuda(), o1, n1, torch.from_numpy(f2).cuda(), o2, n2)
    c = float(res.total_cost.sum())
    if it % 50 == 0:
        print('dtw call %3d pairs %4d cost %.6f reserved %.2f GB' % (it, P, c, torch.cuda.memory_reserved() / 1e9), flush=True)
f1, o1, n1, f2, o2, n2 = bench.synth_dtw_pairs(2000, seed=7)
d1, d2 = torch.from_numpy(f1).cuda(), torch.from_numpy(f2).cuda()
a = dtw_align_batch(d1, o1, n1, d2, o2, n2)
pa, ca = a.path1.clone(), a.total_cost.clone()
for _ in range(50):
    b = dtw_align_batch(d1, o1, n1, d2, o2, n2)
    assert torch.equal(b.total_cost, ca)
mask = a.mask()
assert torch.equal(b.path1[mask], pa[mask])
print('soak ok, %.1f s' % (time.perf_counter() - t0))
