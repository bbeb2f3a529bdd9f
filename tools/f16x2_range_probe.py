"""Diagnostic: fp16 x 2 against the exact-fp32 mode under extreme value ranges (rows of the input spanning ten orders of
magnitude, zero rows, scaled weight blocks): per-tensor gradient errors.   ROWS=300|4096 ACT=relu|sigmoid XSPAN=10 WBLK=1"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from abnet3_amd.model import SiameseNetwork
rows = int(os.environ.get('ROWS', 300)); act = os.environ.get('ACT', 'relu')
xspan = float(os.environ.get('XSPAN', 10)); wblk = int(os.environ.get('WBLK', 1)); zeros = int(os.environ.get('ZEROS', 1))
os.environ['ABN_FUSED_MIN_ROWS'] = '0'
kw = dict(input_dim=40, num_hidden_layers=2, hidden_dim=288, output_dim=64, activation_layer=act, p_dropout=0.0, batch_norm=False)
if act == 'relu': kw['last_non_linearity'] = None
rng = np.random.default_rng(rows)
x = rng.standard_normal((rows, 40)).astype(np.float32)
x *= (10.0 ** rng.uniform(-xspan + 4, 4, size=(rows, 1))).astype(np.float32)
if zeros: x[::17] = 0.0
outs = {}
for prec in ('fp32', 'f16x2', 'bf16x3'):
    torch.manual_seed(3)
    net = SiameseNetwork(**kw).cuda(); net.precision = prec
    with torch.no_grad():
        for k, q in net.named_parameters():
            if k.endswith('weight') and wblk and act == 'relu':
                s = torch.ones(q.shape[0], 1, device=q.device); s[32:64] = 1e-3; s[64:96] = 1e3
                q.mul_(s)
        net.weights_changed_behind_torch()
    net.train()
    e = net.forward_once(torch.from_numpy(x).cuda())
    if 'g' not in outs: outs['g'] = torch.from_numpy((rng.standard_normal(tuple(e.shape)) * 1e-9).astype(np.float32)).cuda()
    e.backward(outs['g'])
    outs[prec] = (e.detach().cpu().numpy().astype(np.float64), {k: q.grad.cpu().numpy().astype(np.float64) for k, q in net.named_parameters()})
e32, g32 = outs['fp32']
for prec in ('f16x2', 'bf16x3'):
    e, g = outs[prec]
    rmax = np.abs(e32).max(axis=1, keepdims=True); ok = rmax[:, 0] > 0
    print(prec, 'emb per-row err %.2e' % (np.abs(e - e32)[ok] / rmax[ok]).max(), ' '.join('%s %.1e' % (k.split('.')[0][:3] + k[-6:], np.abs(g[k] - g32[k]).max() / np.abs(g32[k]).max()) for k in g32))
k = 'hidden_layers.0.bias'
if k in g32:
    d = np.abs(outs['f16x2'][1][k] - g32[k]); i = int(d.argmax())
    print('worst', k, i, outs['f16x2'][1][k][i], g32[k][i], 'max|g|', np.abs(g32[k]).max(), 'n bad', int((d > 1e-4 * np.abs(g32[k]).max()).sum()), np.nonzero(d > 1e-4 * np.abs(g32[k]).max())[0][:20])
