import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import abnet3_amd.loss as L
from abnet3_amd.model import SiameseNetwork
from abnet3_amd.trainer import TrainerSiamese
net = SiameseNetwork(input_dim=40, num_hidden_layers=2, hidden_dim=500, output_dim=100, p_dropout=0.0,
                     activation_layer='sigmoid', output_path='/tmp/abn_ho').cuda()
tr = TrainerSiamese(network=net, loss=L.coscos2(avg=False), optimizer_type='adadelta', lr=0.1, dataloader=None, log_dir='/tmp/abn_runs')
x1, x2, y = torch.randn(64, 40, device='cuda'), torch.randn(64, 40, device='cuda'), torch.ones(64, device='cuda', dtype=torch.float64)
net.train()
for _ in range(20):
    tr.train_step((x1, x2, y), True)
torch.cuda.synchronize()
with torch.autograd.profiler.profile() as prof:
    for _ in range(50):
        tr.train_step((x1, x2, y), True)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='self_cpu_time_total', row_limit=14, max_name_column_width=45))
