"""Host-side microseconds per call of the eager path at a tiny batch (GPU time negligible)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import abnet3_amd.loss as L
from abnet3_amd.model import SiameseNetwork
from abnet3_amd.trainer import TrainerSiamese
net = SiameseNetwork(input_dim=40, num_hidden_layers=2, hidden_dim=500, output_dim=100, p_dropout=0.0,
                     activation_layer='sigmoid', output_path='/tmp/abn_ho').cuda()
tr = TrainerSiamese(network=net, loss=L.coscos2(avg=False), optimizer_type='adadelta', lr=0.1, dataloader=None, log_dir='/tmp/abn_runs')
x1, x2, y = torch.randn(64, 40, device='cuda'), torch.randn(64, 40, device='cuda'), torch.ones(64, device='cuda', dtype=torch.float64)
loss = tr.loss
net.train()


def rate(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()          # host time only: launches are asynchronous
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6, (time.perf_counter() - t0) / n * 1e6


def fwd_nograd():
    with torch.no_grad():
        net(x1, x2)


def fwd_loss():
    e1, e2 = net(x1, x2)
    return loss(e1, e2, y)


def fwd_loss_bwd():
    lv = fwd_loss()
    tr.optimizer.zero_grad()
    tr._backward(lv)


for name, fn in (('forward (no_grad)', fwd_nograd), ('forward + loss', fwd_loss), ('forward + loss + backward', fwd_loss_bwd),
                 ('whole train_step', lambda: tr.train_step((x1, x2, y), True))):
    h, t = rate(fn)
    print('%-28s host %6.1f us/call   wall %6.1f us/call' % (name, h, t), flush=True)

# where the backward's host time goes: time the C calls themselves
from abnet3_amd import _lib
lib = _lib.load()
acc = {}


def wrap(name):
    fn = getattr(lib, name)

    def timed(*a):
        t0 = time.perf_counter()
        r = fn(*a)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    setattr(lib, name, timed)


for nm in ('abn_tower_forward', 'abn_tower_backward', 'abn_pair_loss', 'abn_optimizer_step'):
    wrap(nm)
n = 300
for _ in range(n):
    tr.train_step((x1, x2, y), True)
torch.cuda.synchronize()
for k, v in acc.items():
    print('C call %-22s %6.1f us/step' % (k, v / n * 1e6))
