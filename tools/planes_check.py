"""Gradients of one C2-shaped step on the planes kernels against the per-layer path (same process, ABN_PLANES flipped)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from abnet3_amd.model import SiameseNetwork
from abnet3_amd.loss import coscos2
rows = int(os.environ.get('ROWS', 4096))
cfg = dict(bench.C2)
if os.environ.get('DIMS'):
    d = [int(v) for v in os.environ['DIMS'].split(',')]
    cfg.update(input_dim=d[0], hidden_dim=d[1], output_dim=d[2])
torch.manual_seed(0)
net = SiameseNetwork(output_path='/tmp/abn_pc', **cfg).cuda()
net.precision = os.environ.get('ABNET3_PRECISION', 'bf16x3')
x1, x2 = torch.randn(rows, cfg['input_dim'], device='cuda'), torch.randn(rows, cfg['input_dim'], device='cuda')
y = (torch.rand(rows, device='cuda') > 0.5).float() * 2 - 1
loss = coscos2(avg=False)
net.train()
res = {}
for planes in ('0', '1'):
    os.environ['ABN_PLANES'] = planes
    for p in net.parameters(): p.grad = None
    e1, e2 = net(x1, x2)
    lv = loss(e1, e2, y)
    lv.backward()
    torch.cuda.synchronize()
    res[planes] = ([p.grad.clone() for p in net.parameters()], e1.detach().clone())
print('emb diff', float((res['0'][1] - res['1'][1]).abs().max()))
for (k, _), a, b in zip(net.named_parameters(), res['0'][0], res['1'][0]):
    d = (a - b).abs().max() / a.abs().max()
    print('%-28s max|ref| %.3e  rel diff %.3e  zeros in new %d / %d' % (k, float(a.abs().max()), float(d), int((b == 0).sum()), b.numel()))
