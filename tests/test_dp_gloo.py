"""Data-parallel plumbing on CPU with gloo, world_size 2: the flat-bucket
all-reduce, the grad scale for avg=True, batch sharding and the parameter
broadcast (the GPU path uses the same code with backend nccl = RCCL)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from abnet3_amd import parallel
    r, w, _ = parallel.init_from_env('gloo')
    assert (r, w) == (rank, world) and parallel.world() == (rank, world)
    # per-rank "gradients": rank r holds r+1 everywhere, padded like a flat bucket
    g = torch.full((128,), float(rank + 1))
    scale_sum = parallel.all_reduce_gradients(g, loss_is_mean=False)
    ok = bool((g == sum(range(1, world + 1))).all()) and scale_sum == 1.0
    g2 = torch.full((128,), float(rank + 1))
    scale_mean = parallel.all_reduce_gradients(g2, loss_is_mean=True)
    ok &= abs(scale_mean - 1.0 / world) < 1e-12
    # round-robin sharding: same number of batches on every rank, disjoint
    mine = list(parallel.shard_batches(iter(range(7)), rank, world))
    ok &= mine == [rank, rank + world, rank + 2 * world]
    # evaluation keeps the incomplete last group: rank 0 also gets batch 6
    tail = list(parallel.shard_batches(iter(range(7)), rank, world, drop_tail=False))
    ok &= tail == [rank, rank + world, rank + 2 * world] + ([6] if rank == 0 else [])
    ids = np.arange(7)
    ok &= list(parallel.shard_ids(ids, rank, world, equal=True)) == [rank, rank + 2, rank + 4]
    ok &= list(parallel.shard_ids(ids, rank, world, equal=False)) == list(range(rank, 7, 2))
    # rank 0's draw wins, whatever the local RNG state
    np.random.seed(100 + rank)
    perm = parallel.broadcast_array(np.random.permutation(11))
    np.random.seed(100)
    ok &= (perm == np.random.permutation(11)).all()
    # variable-length exchange (the split DTW alignment's index lists)
    parts = parallel.all_gather_varlen(torch.arange(3 + 4 * rank) + 100 * rank)
    ok &= len(parts) == world and all(
        torch.equal(parts[r], torch.arange(3 + 4 * r) + 100 * r) for r in range(world))
    empty = parallel.all_gather_varlen(torch.zeros(0, dtype=torch.int64))
    ok &= all(p.numel() == 0 for p in empty)
    flat = torch.full((64,), float(rank))
    parallel.broadcast_parameters(flat, src=0)
    ok &= bool((flat == 0).all())
    # sum of per-shard gradients == full-batch gradient (summed loss, avg=False):
    # a linear model y = w.x with loss sum((w.x - t)^2)
    rng = np.random.default_rng(0)
    X, t, wv = rng.standard_normal((8, 4)), rng.standard_normal(8), rng.standard_normal(4)
    full = 2 * X.T @ (X @ wv - t)
    sl = slice(rank * 4, rank * 4 + 4)
    part = torch.from_numpy(2 * X[sl].T @ (X[sl] @ wv - t[sl]))
    parallel.all_reduce_gradients(part, loss_is_mean=False)
    ok &= np.allclose(part.numpy(), full)
    out[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_single_process_defaults():
    from abnet3_amd import parallel
    assert parallel.world() == (0, 1)
    g = torch.ones(4)
    assert parallel.all_reduce_gradients(g, True) == 1.0 and bool((g == 1).all())
    assert list(parallel.shard_batches(iter(range(3)), 0, 1)) == [0, 1, 2]
    assert list(parallel.shard_ids(np.arange(5), 0, 1, equal=True)) == [0, 1, 2, 3, 4]
    assert (parallel.broadcast_array([3, 1, 2]) == [3, 1, 2]).all()
    assert len(parallel.all_gather_varlen(torch.arange(4))) == 1
