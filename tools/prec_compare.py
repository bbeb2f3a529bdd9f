"""fp32 (exact MFMA) vs bf16 vs bf16x3 tower arithmetic on C2: step time, and error of embeddings /
gradients against a float64 evaluation of the same network (torch CPU)."""
import os, sys, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from abnet3_amd.model import SiameseNetwork
from abnet3_amd.loss import coscos2
from abnet3_amd.trainer import TrainerSiamese

torch.manual_seed(0)
base = SiameseNetwork(output_path='/tmp/abn_prec', **bench.C2)
pool = bench.make_pool(seed=0, device=torch.device('cuda'))
x1, x2, y = pool[0]

# float64 reference of forward + gradients on the CPU
import torch.nn as nn
ref = nn.Sequential(nn.Linear(40, 500), nn.Sigmoid(), nn.Linear(500, 500), nn.Sigmoid(), nn.Linear(500, 500), nn.Sigmoid(),
                    nn.Linear(500, 100), nn.Sigmoid()).double()
sd = base.state_dict()
keys = ['input_emb.0', 'hidden_layers.0', 'hidden_layers.3', 'output_layer.0']
for i, k in zip((0, 2, 4, 6), keys):
    ref[i].weight.data = sd[k + '.weight'].double().cpu(); ref[i].bias.data = sd[k + '.bias'].double().cpu()
a, b = ref(x1.double().cpu()), ref(x2.double().cpu())
cs = nn.functional.cosine_similarity(a, b, dim=1, eps=1e-6)
yy = y.cpu()
loss = torch.where(yy == 1, (1 - cs) / 2, cs * cs).sum()
loss.backward()
ref_e = a.detach().numpy()
ref_g = {k + s: getattr(ref[i], s[1:]).grad.numpy() for i, k in zip((0, 2, 4, 6), keys) for s in ('.weight', '.bias')}

def rel(u, v): return float(np.abs(u - v).max() / np.abs(v).max())
for prec in ('fp32', 'bf16x3', 'f16x2', 'bf16'):
    net = copy.deepcopy(base).cuda(); net.precision = prec
    tr = TrainerSiamese(network=net, loss=coscos2(avg=False), optimizer_type='adadelta', lr=0.1, dataloader=None, log_dir='/tmp/abn_runs')
    net.eval()
    with torch.no_grad():
        e = net.forward_once(x1).cpu().numpy()
    net.train()
    tr.optimizer.lr = 0.0                                   # gradients only
    l = float(tr.train_step((x1, x2, y), True))
    g = {k: p.grad.cpu().numpy() for k, p in net.named_parameters()}
    gerr = max(rel(g[k], ref_g[k]) for k in g)
    tr.optimizer.lr = 0.1
    for i in range(50): tr.train_step(pool[i % 8], True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(300): tr.train_step(pool[i % 8], True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300
    print('%-7s step %.4f ms = %5.2f M pairs/s | embeddings vs f64: %.2e | loss rel err %.2e | worst gradient tensor vs f64: %.2e'
          % (prec, dt * 1e3, 4096 / dt / 1e6, rel(e, ref_e), abs(l - float(loss)) / abs(float(loss)), gerr), flush=True)
