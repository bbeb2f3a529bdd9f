"""Embedding error of the bf16 throughput mode against the exact-fp32 mode, planes kernels vs per-layer GEMMs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from abnet3_amd.model import SiameseNetwork
torch.manual_seed(0)
net = SiameseNetwork(output_path='/tmp/abn_e', **bench.C2).cuda()
pool = bench.make_pool(seed=0, device=torch.device('cuda'))
x12 = torch.cat([pool[0][0], pool[0][1]])
net.eval()
with torch.no_grad():
    net.precision = 'fp32'
    ref = torch.cat(net.forward_pair_rows(x12)).double()
    for prec in ('bf16', 'bf16x3'):
        for planes in ('1', '0'):
            os.environ['ABN_PLANES'] = planes
            net.precision = prec
            got = torch.cat(net.forward_pair_rows(x12)).double()
            d = (got - ref).abs()
            print('%-7s planes=%s  max abs %.3e  max rel-to-max %.3e  rms %.3e  ref max %.3f mean %.3f' % (
                prec, planes, float(d.max()), float(d.max() / ref.abs().max()), float((d ** 2).mean().sqrt()), float(ref.abs().max()), float(ref.mean())))
