#!/usr/bin/env python3
"""Times the C2 tower forward (both towers, 8192 rows) through abn_tower_forward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from abnet3_amd.model import SiameseNetwork
torch.manual_seed(0)
net = SiameseNetwork(input_dim=40, num_hidden_layers=2, hidden_dim=500, output_dim=100, p_dropout=0.0,
                     activation_layer='sigmoid').cuda()
x = torch.randn(8192, 40, device='cuda')
net.train()
with torch.no_grad():
    for _ in range(10):
        net.forward_pair_rows(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        net.forward_pair_rows(x)
    e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 100 * 1e-3
print('tower forward (8192 rows): %.1f us  %.1f TFLOP/s (ABN_FUSED=%s)' % (t * 1e6, 2 * 8192 * 570000 / t / 1e12, os.environ.get('ABN_FUSED', '1')))
