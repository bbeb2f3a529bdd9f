"""Inference forward (eval + no_grad, 8192 rows of C2) in the two split arithmetics, with and without BatchNorm."""
import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from abnet3_amd.model import SiameseNetwork
for bn in (False, True):
  for prec in ('bf16x3','f16x2','bf16x3'):
    torch.manual_seed(0)
    net = SiameseNetwork(output_path='/tmp/abn_inf', **dict(bench.C2, batch_norm=bn)).cuda(); net.precision=prec; net.eval()
    x = torch.randn(8192, 40, device='cuda')
    with torch.no_grad():
        for _ in range(50): net.forward_once(x)
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(500): net.forward_once(x)
        torch.cuda.synchronize()
    print('bn', bn, prec, '%.1f us' % ((time.perf_counter()-t0)/500*1e6))
