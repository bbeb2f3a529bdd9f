"""Runs 100 C2 train steps with BatchNorm on (precision from ABN_PRECISION) for a rocprofv3 kernel trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from abnet3_amd.model import SiameseNetwork
from abnet3_amd.loss import coscos2
from abnet3_amd.trainer import TrainerSiamese
torch.manual_seed(0)
net = SiameseNetwork(output_path='/tmp/abn_prof', **dict(bench.C2, batch_norm=True))
if os.environ.get('ABN_PRECISION'): net.precision = os.environ['ABN_PRECISION']
tr = TrainerSiamese(network=net, loss=coscos2(avg=False), optimizer_type='adadelta', lr=0.1, dataloader=None, log_dir='/tmp/abn_runs')
pool = bench.make_pool(seed=0, device=torch.device('cuda'))
net.train()
for i in range(100): tr.train_step(pool[i % 8], True)
torch.cuda.synchronize()
