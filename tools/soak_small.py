"""Race detector for the small-batch step (round 6: tower_wgrad_step.h + abn_step_source): the same seeded planned passes over
ragged batches twice -- captured buckets, the batch read from the plan inside the graph, weight gradients + optimizer in one
launch, dropout drawn in the kernels -- must end with bit-identical parameters and loss sums.
    python tools/soak_small.py [passes]        (each pass: 400 batches of 40 .. 1 000 frame pairs)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                      # noqa: E402
import torch                                            # noqa: E402
from abnet3_amd.dataloader import BatchPlan             # noqa: E402
from abnet3_amd.loss import coscos2                     # noqa: E402
from abnet3_amd.model import SiameseNetwork             # noqa: E402
from abnet3_amd.trainer import TrainerSiamese           # noqa: E402

PASSES = int(sys.argv[1]) if len(sys.argv) > 1 else 4
outs = []
for run in range(2):
    torch.manual_seed(0)
    torch.cuda.manual_seed(0)
    rng = np.random.default_rng(0)
    net = SiameseNetwork(output_path='/tmp/abn_soak_small', input_dim=280, num_hidden_layers=2, hidden_dim=500, output_dim=100,
                         p_dropout=0.1, batch_norm=False, type_init='xavier_uni', activation_layer='sigmoid')
    tr = TrainerSiamese(network=net, loss=coscos2(avg=False), optimizer_type='adadelta', lr=0.1, dataloader=None, log_dir='/tmp/abn_runs')
    sizes = rng.integers(40, 1000, 400)
    offs = np.concatenate(([0], np.cumsum(sizes)))
    n = int(offs[-1])
    table = torch.randn(30000, 280, device='cuda')
    idx1 = torch.randint(0, 30000, (n,), device='cuda')
    idx2 = torch.randint(0, 30000, (n,), device='cuda')
    lab = ((torch.rand(n, device='cuda') > 0.5).double() * 2 - 1)
    net.train()
    sums = []
    for ps in range(PASSES):
        plan = BatchPlan(table, idx1, idx2, lab, offs, list(rng.permutation(400)))
        acc = torch.zeros((), dtype=torch.float64, device='cuda')
        tr._run_planned(plan, True, acc)
        sums.append(acc.clone())
    torch.cuda.synchronize()
    S = torch.stack(sums)
    assert torch.isfinite(S).all()
    outs.append(([p.detach().clone() for p in net.parameters()], S))
same = all(torch.equal(a, b) for a, b in zip(outs[0][0], outs[1][0])) and torch.equal(outs[0][1], outs[1][1])
print('passes', PASSES, 'steps', PASSES * 400, 'bit-identical runs:', same, 'loss sums', [round(float(v), 3) for v in outs[0][1]])
sys.exit(0 if same else 1)
