"""Where a small-batch train step (the reference's canonical regime: OriginalDataLoader, 8 word pairs = a few
hundred 280-d frame pairs per batch) spends its time: eager step, replayed hipGraph, and -- under rocprofv3
--kernel-trace --stats -- the kernels of the replayed step.

    python tools/small_batch_probe.py [pairs ...]           (default 96 160 320 640 1024)
    MODE=graph PAIRS=320 rocprofv3 --kernel-trace --stats -d out -- python3 tools/small_batch_probe.py
    BATCH_NORM=1: the same tower with BatchNorm (planned passes take it since round 4)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                            # noqa: E402
from abnet3_amd.loss import coscos2                     # noqa: E402
from abnet3_amd.model import SiameseNetwork             # noqa: E402
from abnet3_amd.trainer import TrainerSiamese           # noqa: E402

C5 = dict(input_dim=280, num_hidden_layers=2, hidden_dim=500, output_dim=100, p_dropout=0.0, batch_norm=False,
          type_init='xavier_uni', activation_layer='sigmoid')


def make(B):
    torch.manual_seed(0)
    net = SiameseNetwork(output_path='/tmp/abn_sb', **dict(C5, batch_norm=os.environ.get('BATCH_NORM') == '1'))
    if os.environ.get('ABN_PRECISION'):
        net.precision = os.environ['ABN_PRECISION']
    tr = TrainerSiamese(network=net, loss=coscos2(avg=False), optimizer_type='adadelta', lr=0.1, dataloader=None,
                        log_dir='/tmp/abn_runs')
    pool = [(torch.randn(B, 280, device='cuda'), torch.randn(B, 280, device='cuda'),
             ((torch.rand(B, device='cuda') > 0.5).double() * 2 - 1)) for _ in range(4)]
    net.train()
    return net, tr, pool


def timed(fn, n=300, reps=3):
    for i in range(50):
        fn(i)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    return best


def main():
    sizes = [int(a) for a in sys.argv[1:]] or ([int(os.environ['PAIRS'])] if os.environ.get('PAIRS') else [96, 160, 320, 640, 1024])
    mode = os.environ.get('MODE', 'both')
    for B in sizes:
        net, tr, pool = make(B)
        row = 'pairs %5d:' % B
        if mode in ('both', 'eager'):
            t = timed(lambda i: tr.train_step(pool[i % 4], True))
            row += '  eager %.1f us (%.2f M pairs/s)' % (t * 1e6, B / t / 1e6)
        if mode in ('both', 'graph'):
            step = tr.make_graphed_step(pool[0])
            packed = [tr.pack_batch(b) for b in pool]
            t = timed(lambda i: step(packed[i % 4]))
            row += '  graph %.1f us (%.2f M pairs/s)' % (t * 1e6, B / t / 1e6)
            g = step.graph
            t = timed(lambda i: g.replay())
            row += '  bare replay %.1f us' % (t * 1e6)
        if mode in ('both', 'plan'):
            # the trainer's planned passes: one gather launch + one replay of the bucket's captured step
            import numpy as np
            from abnet3_amd.dataloader import BatchPlan
            table = torch.randn(20000, 280, device='cuda')
            nb = 64
            idx1 = torch.randint(0, 20000, (nb * B,), device='cuda')
            idx2 = torch.randint(0, 20000, (nb * B,), device='cuda')
            lab = ((torch.rand(nb * B, device='cuda') > 0.5).double() * 2 - 1)
            plan = BatchPlan(table, idx1, idx2, lab, np.arange(nb + 1) * B, list(range(nb)) * 8)
            acc = torch.zeros((), dtype=torch.float64, device='cuda')
            tr._run_planned(plan, True, acc)             # (whole passes, as optimize_model drives them: 512 steps each)
            torch.cuda.synchronize()
            t = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                tr._run_planned(plan, True, acc)
                torch.cuda.synchronize()
                t = min(t, (time.perf_counter() - t0) / len(plan.order))
            row += '  planned step %.1f us (%.2f M pairs/s)' % (t * 1e6, B / t / 1e6)
        print(row, flush=True)


if __name__ == '__main__':
    main()
