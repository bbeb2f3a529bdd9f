"""Race detector: the same seeded C2 training run twice (dropout drawn in the kernels, eval forwards interleaved);
every parameter must come out bit-identical, every loss finite.  BATCH_NORM=1: the BatchNorm launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from abnet3_amd.model import SiameseNetwork
from abnet3_amd.loss import coscos2
from abnet3_amd.trainer import TrainerSiamese
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
outs = []
for run in range(2):
    torch.manual_seed(0)
    torch.cuda.manual_seed(0)
    net = SiameseNetwork(output_path='/tmp/abn_soak', **dict(bench.C2, p_dropout=0.1, batch_norm=bool(int(os.environ.get('BATCH_NORM', '0')))))
    if os.environ.get('ABN_PRECISION'): net.precision = os.environ['ABN_PRECISION']
    tr = TrainerSiamese(network=net, loss=coscos2(avg=False), optimizer_type='adadelta', lr=0.1, dataloader=None, log_dir='/tmp/abn_runs')
    pool = bench.make_pool(seed=0, device=torch.device('cuda'))
    net.train()
    losses = []
    for i in range(STEPS):
        losses.append(tr.train_step(pool[i % 8], True))
        if i % 250 == 249:
            net.eval()
            with torch.no_grad():
                e = net.forward_once(pool[0][0])
            assert torch.isfinite(e).all()
            net.train()
    torch.cuda.synchronize()
    L = torch.stack(losses)
    assert torch.isfinite(L).all()
    outs.append(([p.detach().clone() for p in net.parameters()], L))
same = all(torch.equal(a, b) for a, b in zip(outs[0][0], outs[1][0])) and torch.equal(outs[0][1], outs[1][1])
print('steps', STEPS, 'precision', net.precision, 'bit-identical runs:', same, 'last loss', float(outs[0][1][-1]))
sys.exit(0 if same else 1)
