"""C2 train step with the reference's default p_dropout = 0.1 against p_dropout = 0 (mask drawing + mask traffic)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from abnet3_amd.model import SiameseNetwork
from abnet3_amd.loss import coscos2
from abnet3_amd.trainer import TrainerSiamese
for p in (0.0, 0.1):
    torch.manual_seed(0)
    net = SiameseNetwork(output_path='/tmp/abn_do', **dict(bench.C2, p_dropout=p))
    tr = TrainerSiamese(network=net, loss=coscos2(avg=False), optimizer_type='adadelta', lr=0.1, dataloader=None, log_dir='/tmp/abn_runs')
    pool = bench.make_pool(seed=0, device=torch.device('cuda'))
    net.train()
    for i in range(200): tr.train_step(pool[i % 8], True)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(300): tr.train_step(pool[i % 8], True)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 300)
    print('p_dropout %.1f: %.4f ms/step' % (p, best * 1e3), flush=True)
