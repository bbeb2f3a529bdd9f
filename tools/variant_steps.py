"""C2-sized train steps of configurations the headline does not cover (graph replay):
dropout 0.1 (the class default), cosmargin, Adam, the multitask network."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
import abnet3_amd.loss as L
from abnet3_amd.model import SiameseNetwork, SiameseMultitaskNetwork
from abnet3_amd.trainer import TrainerSiamese, TrainerSiameseMultitask
pool = bench.make_pool(seed=0, device=torch.device('cuda'))


def run(name, tr, batches):
    tr.network.train()
    step = tr.make_graphed_step(batches[0])
    for i in range(10):
        step(batches[i % len(batches)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(100):
        step(batches[i % len(batches)])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 100
    print('%-34s %.3f ms/step  %.2f M frame-pairs/s' % (name, dt * 1e3, bench.BATCH / dt / 1e6), flush=True)


for name, kw, loss, opt, lr in (
        ('dropout 0.1', dict(bench.C2, p_dropout=0.1), L.coscos2(avg=False), 'adadelta', 0.1),
        ('cosmargin avg', dict(bench.C2), L.cosmargin(avg=True), 'adadelta', 0.1),
        ('adam', dict(bench.C2), L.coscos2(avg=False), 'adam', 0.001),
        ('dropout 0.1 + batch norm', dict(bench.C2, p_dropout=0.1, batch_norm=True), L.coscos2(avg=False), 'adadelta', 0.1)):
    torch.manual_seed(0)
    net = SiameseNetwork(output_path='/tmp/abn_var', **kw)
    run(name, TrainerSiamese(network=net, loss=loss, optimizer_type=opt, lr=lr, dataloader=None, log_dir='/tmp/abn_runs'), pool)
torch.manual_seed(0)
net = SiameseMultitaskNetwork(input_dim=40, num_hidden_layers_shared=2, num_hidden_layers_spk=0, num_hidden_layers_phn=0,
                              hidden_dim=500, output_dim=100, p_dropout=0.0, activation_layer='sigmoid', output_path='/tmp/abn_var')
mt = [(a, b, y, y.clone()) for a, b, y in pool]
run('multitask (2 heads)', TrainerSiameseMultitask(network=net, loss=L.weighted_loss_multi(loss_spk=L.coscos2(avg=False), loss_phn=L.coscos2(avg=False), weight=0.5),
                                                  optimizer_type='adadelta', lr=0.1, dataloader=None, log_dir='/tmp/abn_runs'), mt)
