"""One eager train step at a small batch (B rows per tower, 40->500x2->100) for a kernel trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import abnet3_amd.loss as L
from abnet3_amd.model import SiameseNetwork
from abnet3_amd.trainer import TrainerSiamese
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(0)
dev = lambda a: torch.from_numpy(a).cuda()
b = (dev(rng.standard_normal((B, 40)).astype(np.float32)), dev(rng.standard_normal((B, 40)).astype(np.float32)), dev(rng.choice([1.0, -1.0], B)))
net = SiameseNetwork(input_dim=40, num_hidden_layers=2, hidden_dim=500, output_dim=100, p_dropout=0.0,
                     activation_layer='sigmoid', output_path='/tmp/abn_small').cuda()
tr = TrainerSiamese(network=net, loss=L.coscos2(avg=False), optimizer_type='adadelta', lr=0.1, dataloader=None, log_dir='/tmp/abn_runs')
net.train()
for _ in range(20):
    tr.train_step(b, True)
torch.cuda.synchronize()
