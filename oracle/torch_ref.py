"""ORACLE (test infrastructure, never shipped as a product path).

torch-CPU restatement of the reference's Siamese train step: the same
torch.nn modules in the same construction order as abnet3/model.py:110-177 (so
that one torch.manual_seed gives the reference's initial weights), the loss
ops of abnet3/loss.py:57-67 / :95-105 and the statements of
abnet3/trainer.py:236-242.  It exists (a) to regenerate seeded weights for the
C2-sized golden fixtures, which only hold checksums, and (b) as the CPU
baseline of bench.py ("port": it executes what the reference executes, ATen
CPU kernels, on the GPU box's host cores).  Pinned by tests/golden/train_c2_*.
"""
import numpy as np
import torch
import torch.nn as nn

ACT = {'relu': nn.ReLU, 'sigmoid': nn.Sigmoid, 'tanh': nn.Tanh, 'softmax': nn.Softmax}
INIT = {'xavier_uni': nn.init.xavier_uniform_, 'xavier_normal': nn.init.xavier_normal_,
        'orthogonal': nn.init.orthogonal_}


class RefSiamese(nn.Module):
    def __init__(self, input_dim, num_hidden_layers, hidden_dim, output_dim,
                 p_dropout=0.1, batch_norm=False, type_init='xavier_uni',
                 activation_layer=None, last_non_linearity='default'):
        super().__init__()
        act = ACT[activation_layer]

        def block(i, o, a):
            mods = [nn.Linear(i, o), nn.Dropout(p=p_dropout)]
            if batch_norm:
                mods.append(nn.BatchNorm1d(o))
            if a is not None:
                mods.append(a())
            return mods
        self.input_emb = nn.Sequential(*block(input_dim, hidden_dim, act))
        hidden = []
        for _ in range(num_hidden_layers):
            hidden += block(hidden_dim, hidden_dim, act)
        self.hidden_layers = nn.Sequential(*hidden)
        if last_non_linearity == 'default':
            last = act
        elif last_non_linearity is None:
            last = None
        else:
            last = ACT[last_non_linearity]
        self.output_layer = nn.Sequential(*block(hidden_dim, output_dim, last))
        gain = nn.init.calculate_gain(activation_layer)

        def init(layer):
            if isinstance(layer, nn.Linear):
                INIT[type_init](layer.weight.data, gain=gain)
                layer.bias.data.fill_(0.0)
        self.apply(init)

    def forward_once(self, x):
        return self.output_layer(self.hidden_layers(self.input_emb(x)))

    def forward(self, x1, x2):
        return self.forward_once(x1), self.forward_once(x2)


def build(seed, **kw):
    torch.manual_seed(seed)
    return RefSiamese(**kw)


def make_inputs(B, D, seed):
    torch.manual_seed(seed)
    x1 = torch.randn(B, D)
    x2 = torch.randn(B, D)
    np.random.seed(seed)
    y = np.random.choice([1, -1], B)
    return x1, x2, y


def pair_loss(e1, e2, y, kind='coscos2', margin=0.5, avg=True):
    cos = nn.functional.cosine_similarity(e1, e2, dim=1, eps=1e-6)
    same = torch.eq(y, 1)
    diff = torch.eq(y, -1)
    if kind == 'coscos2':
        t = torch.where(same, (1 - cos) / 2, torch.where(diff, cos * cos, cos))
    else:
        t = torch.where(same, 1 - cos,
                        torch.where(diff, torch.clamp(cos - margin, min=0), cos))
    out = t.sum()
    return out / e1.size(0) if avg else out


def train_step(net, opt, x1, x2, y, kind='coscos2', margin=0.5, avg=False):
    e1, e2 = net(x1, x2)
    loss = pair_loss(e1, e2, y, kind, margin, avg)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss


def run_steps(kw, seed, batches, n_steps, dtype=torch.float32, optimizer='adadelta', lr=0.1, kind='coscos2', margin=0.5,
              avg=False):
    """`n_steps` train steps of a seeded network on `batches` (cycled), in `dtype`: torch.float32 is what the
    reference executes; torch.float64 is the same sequence of operations in double -- the yardstick the fp32
    results of the reference AND of the HIP kernels are measured against (tests/test_gpu_timed_path.py: both are
    fp32-grade approximations of it; at the Siamese initialisation, where cos sits in [0.99998, 0.999997], they
    differ from each other by more than either differs from the truth).  The float64 network starts from the
    float32 initial weights (the seed's draw), exactly.
    Returns (losses, first-step gradients {key: array}, parameters after the steps {key: array})."""
    net = build(seed=seed, **kw)
    if dtype == torch.float64:
        net = net.double()
    opt = {'adadelta': lambda p: torch.optim.Adadelta(p, lr=lr), 'sgd': lambda p: torch.optim.SGD(p, lr=lr, momentum=0.9)}[optimizer](net.parameters())
    net.train()
    losses, grads0 = [], None
    for s in range(n_steps):
        x1, x2, y = batches[s % len(batches)]
        y = torch.as_tensor(y)
        loss = train_step(net, opt, x1.to(dtype), x2.to(dtype), y, kind, margin, avg)
        losses.append(float(loss.detach()))
        if s == 0:
            grads0 = {k: p.grad.detach().numpy().copy() for k, p in net.named_parameters()}
    return losses, grads0, {k: v.detach().numpy().copy() for k, v in net.state_dict().items()}
