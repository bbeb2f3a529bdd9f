"""Epoch loop on small fixed-size batches (FramesDataLoader's default 100 frame pairs):
all-eager steps vs train_step_auto (captured hipGraph for the recurring shape)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import abnet3_amd.loss as L
from abnet3_amd.model import SiameseNetwork
from abnet3_amd.trainer import TrainerSiamese


class Loader(object):
    def __init__(self, batches): self.b = batches
    def batch_iterator(self, train_mode=True): return iter(self.b if train_mode else self.b[:2])
    def whoami(self): return {}

rng = np.random.default_rng(0)
dev = lambda a: torch.from_numpy(a).cuda()
for B, hidden in ((100, 100), (100, 500), (1000, 500)):
    batches = [(dev(rng.standard_normal((B, 40)).astype(np.float32)), dev(rng.standard_normal((B, 40)).astype(np.float32)),
                dev(rng.choice([1.0, -1.0], B))) for _ in range(400)]
    for graph_steps in (False, True):
        torch.manual_seed(0)
        net = SiameseNetwork(input_dim=40, num_hidden_layers=2, hidden_dim=hidden, output_dim=100, p_dropout=0.1,
                             activation_layer='sigmoid', output_path='/tmp/abn_epoch').cuda()
        tr = TrainerSiamese(network=net, loss=L.coscos2(avg=False), optimizer_type='adadelta', lr=0.1,
                            dataloader=Loader(batches), log_dir='/tmp/abn_runs')
        tr.graph_steps = graph_steps
        tr.train_losses, tr.dev_losses = [], []
        tr.pretty_print_losses = lambda *a: None
        tr.optimize_model(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.optimize_model(True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print('B=%4d hidden=%3d  %-10s %.1f us/step  %.2f M frame-pairs/s' % (
            B, hidden, 'auto-graph' if graph_steps else 'eager', dt / 400 * 1e6, 400 * B / dt / 1e6), flush=True)
