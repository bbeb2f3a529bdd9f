#!/usr/bin/env python3
"""Diagnostic (ABN_STAMPS build: tools/build_stamps.sh): phase timeline of the data-gradient chain
(tower_dgrad_planes_kernel, csrc/tower_planes.h) inside a C2 train step: wave 0 of every workgroup, medians over the
workgroups, cycles of s_memtime (comparable only within a workgroup: every XCD counts from a base of its own)."""
import os, sys
os.environ.setdefault('ABNET3_HIP_LIB', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'variants', 'lib_stamps.so'))   # tools/build_stamps.sh
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
buf = torch.zeros(1024 * 64, dtype=torch.int64, device='cuda')
os.environ['ABN_DSTAMP_BUF'] = str(buf.data_ptr())
import bench
from abnet3_amd.model import SiameseNetwork
from abnet3_amd.loss import coscos2
from abnet3_amd.trainer import TrainerSiamese
torch.manual_seed(0)
net = SiameseNetwork(output_path='/tmp/abn_st', **bench.C2)
tr = TrainerSiamese(network=net, loss=coscos2(avg=False), optimizer_type='adadelta', lr=0.1, dataloader=None, log_dir='/tmp/abn_runs')
pool = bench.make_pool(seed=0, device=torch.device('cuda'))
net.train()
for i in range(20):
    tr.train_step(pool[i % 8], True)
torch.cuda.synchronize()
nl = 4
# slots: 0 start, 1 loss phase done, 2 dZ_top in the image (barrier passed), then per layer (top down, i = 0 ..):
# 4 + 4 i k-loop done, 5 + 4 i act' gather / publish / finish / row maxima done, 6 + 4 i barrier passed, 7 + 4 i layer done
s = buf.cpu().numpy().reshape(1024, 64)[:256].astype(np.float64)
def phase(name, a, b):
    d = s[:, b] - s[:, a]
    print('%-58s median %8.0f   min %8.0f   max %8.0f' % (name, np.median(d), d.min(), d.max()))
phase('loss phase', 0, 1)
phase('dZ_top -> image, transposed image, barrier', 1, 2)
prev = 2
for i, l in enumerate(range(nl - 1, 0, -1)):
    phase('L%d ring fill + k-loop' % l, prev, 4 + 4 * i)
    phase("L%d act' gather (+ loss ticket), finish, row maxima" % l, 4 + 4 * i, 5 + 4 * i)
    phase('L%d barrier (wave 0 waits for the others)' % l, 5 + 4 * i, 6 + 4 * i)
    phase('L%d scales, image, transposed image, barrier' % l, 6 + 4 * i, 7 + 4 * i)
    prev = 7 + 4 * i
phase('workgroup total', 0, prev)
