"""How fast two trainings that differ only by fp32 rounding drift apart on the C5 workload: planned passes
(padded, captured steps) against the plain iterator (eager steps), and -- as the yardstick -- the iterator in
the exact-fp32 arithmetic against the iterator in bf16x3.  Prints the largest relative parameter difference
every few hundred steps: rounding-level differences grow smoothly (the training dynamics amplify them), a
defect shows as a jump."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                     # noqa: E402
from tools import c5_pipeline                                    # noqa: E402
from tools.c5_corpus import sample_pairs, synth_corpus           # noqa: E402
from abnet3_amd.loss import coscos2                              # noqa: E402
from abnet3_amd.model import SiameseNetwork                      # noqa: E402
from abnet3_amd.trainer import TrainerSiamese                    # noqa: E402


def make(dc, pairs, planned, precision):
    np.random.seed(0)
    torch.manual_seed(0)
    dl = c5_pipeline.make_loader('original', dc, pairs[0], pairs[1])
    net = SiameseNetwork(output_path='/tmp/abn_div', **c5_pipeline.C5_NET)
    net.precision = precision
    tr = TrainerSiamese(network=net, loss=coscos2(avg=False), optimizer_type='adadelta', lr=0.1, dataloader=dl, log_dir='/tmp/abn_div_runs')
    tr.planned_passes = planned
    net.train()
    return tr, dl


def main():
    corpus = synth_corpus(n_utts=300, seed=0, device='cuda')
    pairs = sample_pairs(corpus, n_pairs=8000, seed=0)
    dc, _ = c5_pipeline.build_features(corpus, {})
    runs = {'planned bf16x3': make(dc, pairs, True, 'bf16x3'), 'iterator bf16x3': make(dc, pairs, False, 'bf16x3'),
            'iterator fp32': make(dc, pairs, False, 'fp32')}
    plans = {k: dl.plan(True) for k, (tr, dl) in runs.items()}
    order = plans['planned bf16x3'].order
    every = 100
    for step, bid in enumerate(order):
        for k, (tr, dl) in runs.items():
            if k.startswith('planned'):
                assert tr._planned_step(plans[k], bid)
            else:
                tr.train_step(plans[k].materialise(bid), True)
        if step % every == every - 1 or step < 3:
            flat = {k: tr.network.flat_parameters().double() for k, (tr, dl) in runs.items()}
            ref = flat['iterator bf16x3']
            d_plan = float((flat['planned bf16x3'] - ref).abs().max() / ref.abs().max())
            d_f32 = float((flat['iterator fp32'] - ref).abs().max() / ref.abs().max())
            print('step %5d  planned vs iterator %.3e   fp32 vs bf16x3 (both iterator) %.3e' % (step + 1, d_plan, d_f32), flush=True)


if __name__ == '__main__':
    main()
