#!/usr/bin/env python3
"""Headline benchmark: frame-pairs/s of one Siamese train step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[1] / SURVEY.md section 8 "C2"): SiameseNetwork
40 -> 500 x2 -> 100 (input_dim=40, num_hidden_layers=2, hidden_dim=500,
output_dim=100, sigmoid, no BN, dropout 0), coscos2(avg=False), Adadelta(lr=0.1),
4096 synthetic frame pairs per GPU per step, fp32 (exact-fp32 MFMA, the parity
mode).  One step = what TrainerSiamese.train_step runs: forward of both towers,
pair loss, backward, [RCCL all-reduce of the flat gradient bucket], optimizer.
Inputs are resident in HBM before the timed region.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     fp32-MFMA roofline of the dominant kernel, timed live with
               HIP events on the launch stream
  cpu_baseline the oracle's torch-CPU restatement of the same step timed on
               this host's cores (N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

C2 = dict(input_dim=40, num_hidden_layers=2, hidden_dim=500, output_dim=100,
          p_dropout=0.0, batch_norm=False, type_init='xavier_uni',
          activation_layer='sigmoid')
BATCH = 4096
POOL = 8
FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
# MACs per tower row: fwd 570 000, wgrad 570 000, dgrad 550 000
FLOP_PER_PAIR = 2 * 2 * (570000 + 570000 + 550000)


def make_pool(seed, device):
    import numpy as np
    import torch
    g = torch.Generator().manual_seed(seed)
    pool = []
    for i in range(POOL):
        x1 = torch.randn(BATCH, 40, generator=g)
        x2 = torch.randn(BATCH, 40, generator=g)
        y = torch.from_numpy(np.random.default_rng(seed * 100 + i).choice([1, -1], BATCH))
        pool.append((x1.to(device), x2.to(device), y.to(device)))
    return pool


def gemm_roofline(torch, reps=50):
    """Times the dominant kernel -- gemm_f32_kernel<128,128,K-contig,K-contig,FWD>,
    launched three times per step (layers 40->500, 500->500, 500->500 over the
    2x4096 tower rows) -- in isolation through the single-layer C-ABI entry,
    with HIP events on the launch stream; reports per-launch averages."""
    from abnet3_amd import _lib
    lib = _lib.load()
    dev = 'cuda'
    rows = 2 * BATCH
    shapes = [(40, 500), (500, 500), (500, 500)]
    bufs = []
    for k, n in shapes:
        bufs.append((torch.randn(rows, k, device=dev), torch.randn(n, k, device=dev) * 0.05,
                     torch.zeros(n, device=dev), torch.empty(rows, n, device=dev)))

    def run_all():
        for (k, n), (x, w, b, y) in zip(shapes, bufs):
            _lib.check(lib.abn_linear_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), rows, k, n,
                                              _lib.ACT['sigmoid'], _lib.ptr(y), _lib.stream()),
                       'abn_linear_forward')
    for _ in range(10):
        run_all()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run_all()
    e1.record()
    torch.cuda.synchronize()
    launches = reps * len(shapes)
    avg_s = e0.elapsed_time(e1) * 1e-3 / launches
    flop_per_launch = sum(2.0 * rows * k * n for k, n in shapes) / len(shapes)
    achieved = flop_per_launch / avg_s / 1e12
    return {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': FP32_MFMA_PEAK_TFLOPS,
            'unit': 'TFLOP/s', 'frac': round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
            'traffic': None,
            'kernel': 'gemm_f32_kernel<128,128,Kcontig,Kcontig,FWD>',
            'avg_launch_us': round(avg_s * 1e6, 2),
            'flop_per_launch': flop_per_launch}


def cpu_baseline(torch, budget_s=12.0):
    """The reference's train step as the oracle restates it with torch.nn on
    the CPU (oracle/torch_ref.py, pinned by tests/golden/train_c2_*), timed on
    this host's cores on a bounded sample of the same workload."""
    from oracle import torch_ref
    # the box's CPU share for one GPU is 16 cores (more threads than that only
    # oversubscribe the cgroup); never more than the affinity mask allows
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(16, avail)))
    net = torch_ref.build(seed=0, **C2)
    opt = torch.optim.Adadelta(net.parameters(), lr=0.1)
    pool = [torch_ref.make_inputs(BATCH, 40, s) for s in range(2)]
    pool = [(a, b, torch.from_numpy(y)) for a, b, y in pool]
    net.train()
    for i in range(3):
        torch_ref.train_step(net, opt, *pool[i % 2])
    t0 = time.perf_counter()
    n = 0
    while True:
        torch_ref.train_step(net, opt, *pool[n % 2])
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 2000:
            break
    dt = time.perf_counter() - t0
    return {'value': round(n * BATCH / dt, 1), 'unit': 'frame-pairs/s',
            'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '%d steps of the same C2 step (B=%d, fp32, torch-CPU restatement '
                      'of the reference, %.1f s)' % (n, BATCH, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=30)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph', action='store_true',
                    help='launch every step eagerly instead of replaying a hipGraph')
    args = ap.parse_args()

    import torch
    from abnet3_amd import parallel
    rank, world, local = parallel.init_from_env('nccl')
    if world != args.gpus:
        if args.gpus > 1:
            raise SystemExit('bench.py --gpus %d must be launched with torch.distributed.run '
                             '--nproc-per-node %d' % (args.gpus, args.gpus))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    from abnet3_amd.model import SiameseNetwork
    from abnet3_amd.loss import coscos2
    from abnet3_amd.trainer import TrainerSiamese

    torch.manual_seed(0)                       # identical replicas
    net = SiameseNetwork(output_path='/tmp/abnet3_bench_r%d' % rank, **C2)
    trainer = TrainerSiamese(network=net, loss=coscos2(avg=False), optimizer_type='adadelta',
                             lr=0.1, dataloader=None, log_dir='/tmp/abnet3_bench_runs')
    pool = make_pool(seed=rank, device=dev)    # rank r sees its own pairs
    net.train()
    stepper = trainer.make_graphed_step(pool[0]) if not args.no_graph else None

    def step(i):
        if stepper is not None:
            return stepper(pool[i % POOL])
        return trainer.train_step(pool[i % POOL], True)

    for i in range(args.warmup):
        step(i)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    last_loss = float(loss)

    if rank == 0:
        value = args.steps * BATCH * world / elapsed
        out = {
            'metric': 'frame-pairs/sec (Siamese train step: fwd both towers + coscos2 + bwd + Adadelta)',
            'value': round(value, 1), 'unit': 'frame-pairs/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'C2: SiameseNetwork 40->500x2->100 sigmoid, coscos2(avg=False), '
                                   'Adadelta(0.1), 4096 frame pairs per GPU per step, 40-d N(0,1) frames',
                       'pairs_per_gpu': BATCH, 'global_pairs': BATCH * world,
                       'parallelism': 'dp%d' % world, 'graph_replay': stepper is not None},
            'tflops_whole_step': round(value * FLOP_PER_PAIR / 1e12, 2),
            'last_loss': last_loss,
        }
        out['roofline'] = gemm_roofline(torch)
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(torch)
            out['cpu_baseline'] = cb
            out['gpu_over_cpu'] = round(value / cb['value'], 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
