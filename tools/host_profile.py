"""Host-side cost of an eager train step (variable-size batches cannot be graph-captured)."""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import abnet3_amd.loss as L
from abnet3_amd.model import SiameseNetwork
from abnet3_amd.trainer import TrainerSiamese
rng = np.random.default_rng(0)
dev = lambda a: torch.from_numpy(a).cuda()
net = SiameseNetwork(input_dim=40, num_hidden_layers=2, hidden_dim=500, output_dim=100, p_dropout=0.0,
                     activation_layer='sigmoid', output_path='/tmp/abn_hp').cuda()
tr = TrainerSiamese(network=net, loss=L.coscos2(avg=False), optimizer_type='adadelta', lr=0.1, dataloader=None, log_dir='/tmp/abn_runs')
batches = [(dev(rng.standard_normal((n, 40)).astype(np.float32)), dev(rng.standard_normal((n, 40)).astype(np.float32)),
            dev(rng.choice([1.0, -1.0], n))) for n in rng.integers(300, 700, 64)]
net.train()
for b in batches[:10]:
    tr.train_step(b, True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for b in batches:
    tr.train_step(b, True)
torch.cuda.synchronize()
print('eager step, variable batch 300-700 pairs: %.1f us/step' % ((time.perf_counter() - t0) / len(batches) * 1e6))
pr = cProfile.Profile(); pr.enable()
for b in batches:
    tr.train_step(b, True)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
