"""ms per C2 train step with BatchNorm on (B = 4096 pairs, Adadelta), 300 steps after 100 of warm-up;
ABN_BN_PLANES=0 keeps the per-layer kernels, P_DROPOUT sets p_dropout (ABN_DROPOUT_IN_KERNEL=0: mask tensors)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from abnet3_amd.model import SiameseNetwork
from abnet3_amd.loss import coscos2
from abnet3_amd.trainer import TrainerSiamese
torch.manual_seed(0)
net = SiameseNetwork(output_path='/tmp/abn_prof', **dict(bench.C2, batch_norm=True, p_dropout=float(os.environ.get('P_DROPOUT', 0.0))))
if os.environ.get('ABN_PRECISION'): net.precision = os.environ['ABN_PRECISION']
tr = TrainerSiamese(network=net, loss=coscos2(avg=False), optimizer_type='adadelta', lr=0.1, dataloader=None, log_dir='/tmp/abn_runs')
pool = bench.make_pool(seed=0, device=torch.device('cuda'))
net.train()
for i in range(100): tr.train_step(pool[i % 8], True)
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(300): loss = tr.train_step(pool[i % 8], True)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 300)
print('ABN_BN_PLANES=%s p_dropout=%s in-kernel=%s %s: %.4f ms/step  loss %.6f' % (os.environ.get('ABN_BN_PLANES', '1'), os.environ.get('P_DROPOUT', '0'), os.environ.get('ABN_DROPOUT_IN_KERNEL', '1'), net.precision, best * 1e3, float(loss)), flush=True)
