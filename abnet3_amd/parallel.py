"""Data-parallel plumbing: one process per GPU, RCCL over xGMI.

The reference has no multi-process path at all.  Pair
minibatches are independent, so data-parallel training needs exactly one
exchange per step: a SUM all-reduce of the flat fp32 gradient bucket between
backward and optimizer step (abnet3/trainer.py:239 -> :240).  The bucket is ONE
contiguous buffer (SiameseNetwork.flat_grad()), i.e. one small-message
collective (2.29 MB for C2) instead of eight.

torch.distributed's "nccl" backend IS RCCL on ROCm; "gloo" is used by the CPU
tests of this file's logic.
"""
import os

import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_from_env(backend=None):
    """Initialises torch.distributed from RANK / WORLD_SIZE / MASTER_* if the
    process was launched by torch.distributed.run; returns (rank, world, local)."""
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if ws > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
        elif torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend=backend, rank=rank, world_size=ws)
    return rank, ws, local


def all_reduce_gradients(flat_grad, loss_is_mean):
    """SUM-reduces the flat gradient bucket in place over all ranks.

    Returns the factor the optimizer must apply to the reduced gradient so that
    R ranks x B pairs equal one process with R*B pairs: 1 for a summed loss
    (avg=False, the canonical configuration), 1/R for a mean loss (avg=True)."""
    _, ws = world()
    if ws == 1:
        return 1.0
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return 1.0 / ws if loss_is_mean else 1.0


def shard_batches(iterator, rank, ws):
    """Round-robin shard of a batch iterator: rank r takes batches r, r+ws, ...
    Only complete groups of `ws` batches are used, so every rank performs the
    same number of steps (the all-reduce is a collective)."""
    if ws == 1:
        for b in iterator:
            yield b
        return
    group = []
    for b in iterator:
        group.append(b)
        if len(group) == ws:
            yield group[rank]
            group = []


def broadcast_parameters(flat_params, src=0):
    _, ws = world()
    if ws > 1:
        dist.broadcast(flat_params, src=src)
