"""Decodes the operand-fragment images of tower_planes.h from a forward workspace / backward scratch and compares
them with the tensors they were made from (used by tests/test_gpu_planes.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch


def bf16_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def perm16():
    """k index inside a 16-step held by element j of lane half h: 8 (j >> 2) + 4 h + (j & 3)  -> [h][j]"""
    return np.array([[8 * (j >> 2) + 4 * h + (j & 3) for j in range(8)] for h in range(2)])


def decode(buf_u16, nblk, nsteps, np_):
    """image [nblk][nsteps][np][64 lanes][8] bf16 -> float64 matrix [32 nblk operand rows][16 nsteps sum index].
    np_ = 2 (fp16 x 2): the planes are fp16 values of the scaled tensor, one inverse scale (fp32) per 32-row
    block behind the fragments."""
    if np_ == 2:
        n = nblk * nsteps * 2 * 512
        img = buf_u16[:n].view(np.float16).reshape(nblk, nsteps, 2, 2, 32, 8).astype(np.float64).sum(axis=2)
        inv = buf_u16[n:n + 2 * nblk].view(np.float32).astype(np.float64)
        img = img * inv[:, None, None, None, None]
    else:
        img = bf16_to_f32(buf_u16[:nblk * nsteps * np_ * 512].reshape(nblk, nsteps, np_, 2, 32, 8)).astype(np.float64).sum(axis=2)
    out = np.zeros((nblk * 32, nsteps * 16))
    P = perm16()
    for h in range(2):
        for j in range(8):
            # img[blk, s, h, r, j] -> row 32 blk + r, column 16 s + P[h][j]
            out[:, P[h][j]::16] = img[:, :, h, :, j].transpose(0, 2, 1).reshape(nblk * 32, nsteps)
    return out


def decode_t(buf_u16, nblk, nsteps, np_):
    """Transposed image (tower_planes.h emit_planes) -> float64 matrix [32 nblk features][16 nsteps batch rows].
    np_ = 1: [nblk][nsteps][64 lanes][8 bf16], the layout decode() reads with one plane.
    np_ = 3, 2: [nblk][nsteps][elements 0..3 | 4..7][64 lanes][4 fp32]."""
    if np_ == 1:
        return decode(buf_u16, nblk, nsteps, 1)
    f = buf_u16[:nblk * nsteps * 1024].view(np.float32).reshape(nblk, nsteps, 2, 2, 32, 4).astype(np.float64)
    out = np.zeros((nblk * 32, nsteps * 16))
    P = perm16()
    for h in range(2):
        for j in range(8):
            # f[blk, s, j >> 2, h, c, j & 3] -> row 32 blk + c, column 16 s + P[h][j]
            out[:, P[h][j]::16] = f[:, :, j >> 2, h, :, j & 3].transpose(0, 2, 1).reshape(nblk * 32, nsteps)
    return out
