"""ORACLE (test infrastructure): ctypes front-end of oracle/dtw.c.

Restates abnet3/utils.py:40-60 (cosine_distance) and :147-153
(get_dtw_alignment).  The DP's parity is UNPINNED (third-party dtw.DTW absent
from /root/reference; see dtw.c header); cosine_distance is pinned bit for bit by
tests/golden/cosdist_libm.npz (reference outputs, numpy on its libm path) and to
2e-7 by tests/golden/cosdist.npz (numpy on its SVML path).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'liboracle.so')
_lib = None


def build(force=False):
    if force or not os.path.exists(_SO) or (
            os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, 'dtw.c'))):
        subprocess.check_call(['make', '-C', _HERE, '-s'] + (['-B'] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        c = ctypes
        L.abn_oracle_acosf.restype = c.c_float
        L.abn_oracle_acosf.argtypes = [c.c_float]
        L.abn_oracle_cosine_distance_f32.restype = c.c_int
        L.abn_oracle_cosine_distance_f32.argtypes = [
            c.c_void_p, c.c_int64, c.c_void_p, c.c_int64, c.c_int64, c.c_void_p]
        L.abn_oracle_cosine_distance_f64.restype = c.c_int
        L.abn_oracle_cosine_distance_f64.argtypes = [
            c.c_void_p, c.c_int64, c.c_void_p, c.c_int64, c.c_int64, c.c_void_p]
        L.abn_oracle_row_norm.restype = c.c_float
        L.abn_oracle_row_norm.argtypes = [c.c_void_p, c.c_int64]
        L.abn_oracle_acosf_vs_libm.restype = c.c_int64
        L.abn_oracle_acosf_vs_libm.argtypes = [c.c_uint32, c.c_uint32, c.c_uint32, c.c_void_p]
        L.abn_oracle_dtw.restype = c.c_int64
        L.abn_oracle_dtw.argtypes = [c.c_void_p, c.c_int64, c.c_int64,
                                     c.c_void_p, c.c_void_p, c.c_void_p]
        L.abn_oracle_dtw_batch.restype = c.c_int64
        L.abn_oracle_dtw_batch.argtypes = [c.c_void_p] * 6 + [
            c.c_int64, c.c_int64, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int64]
        L.abn_oracle_dtw_batch_mt.restype = c.c_int64
        L.abn_oracle_dtw_batch_mt.argtypes = L.abn_oracle_dtw_batch.argtypes + [c.c_int]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def cosine_distance(x, y, check=True):
    """[N,D], [M,D] (both float32 or both float64, utils.py:41-42) -> float64 [N,M],
    computed in the input precision; raises AssertionError like utils.py:59 when an
    entry is NaN/negative (check=False returns the matrix and the flag instead)."""
    x, y = np.asarray(x), np.asarray(y)
    assert (x.dtype == np.float64 and y.dtype == np.float64) or (
        x.dtype == np.float32 and y.dtype == np.float32)
    x, y = np.ascontiguousarray(x), np.ascontiguousarray(y)
    d = np.empty((x.shape[0], y.shape[0]), dtype=np.float64)
    fn = (lib().abn_oracle_cosine_distance_f32 if x.dtype == np.float32
          else lib().abn_oracle_cosine_distance_f64)
    bad = fn(_p(x), x.shape[0], _p(y), y.shape[0], x.shape[1], _p(d))
    if not check:
        return d, bool(bad)
    assert not bad, 'cosine_distance produced NaN / negative entries'
    return d


def row_norms(x):
    """np.sqrt(np.sum(x ** 2, axis=1)) for float32 rows, numpy's summation order."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    return np.array([lib().abn_oracle_row_norm(_p(x[i:i + 1]), x.shape[1])
                     for i in range(x.shape[0])], dtype=np.float32)


def acosf_vs_libm(lo_bits, hi_bits, step=1):
    """(# of float32 bit patterns in [lo, hi] (stride `step`) where the restated acosf
    differs from this machine's libm, first such pattern)."""
    first = ctypes.c_uint32(0)
    n = lib().abn_oracle_acosf_vs_libm(lo_bits, hi_bits, step, ctypes.byref(first))
    return int(n), int(first.value)


def dtw_path(d):
    """float64 [N,M] -> (path1, path2) int32 arrays, start -> end."""
    d = np.ascontiguousarray(d, dtype=np.float64)
    N, M = d.shape
    p1 = np.empty(N + M, dtype=np.int32)
    p2 = np.empty(N + M, dtype=np.int32)
    tot = ctypes.c_double()
    n = lib().abn_oracle_dtw(_p(d), N, M, _p(p1), _p(p2), ctypes.byref(tot))
    return p1[:n].copy(), p2[:n].copy()


def dtw_cost(d):
    d = np.ascontiguousarray(d, dtype=np.float64)
    N, M = d.shape
    p1 = np.empty(N + M, dtype=np.int32)
    p2 = np.empty(N + M, dtype=np.int32)
    tot = ctypes.c_double()
    lib().abn_oracle_dtw(_p(d), N, M, _p(p1), _p(p2), ctypes.byref(tot))
    return tot.value


def get_dtw_alignment(feat1, feat2):
    """utils.py:147-153."""
    d = cosine_distance(feat1, feat2)
    p1, p2 = dtw_path(d)
    assert len(p1) == len(p2)
    return p1, p2


def dtw_batch(feats1, off1, n1, feats2, off2, n2, path_stride, threads=1):
    """Batch front-end used by the CPU baseline. Returns (p1, p2, len, cells).
    threads > 1: OpenMP over the pairs (independent: identical results)."""
    feats1 = np.ascontiguousarray(feats1, dtype=np.float32)
    feats2 = np.ascontiguousarray(feats2, dtype=np.float32)
    off1 = np.ascontiguousarray(off1, dtype=np.int64)
    off2 = np.ascontiguousarray(off2, dtype=np.int64)
    n1 = np.ascontiguousarray(n1, dtype=np.int32)
    n2 = np.ascontiguousarray(n2, dtype=np.int32)
    P = len(n1)
    p1 = np.full((P, path_stride), -1, dtype=np.int32)
    p2 = np.full((P, path_stride), -1, dtype=np.int32)
    ln = np.zeros(P, dtype=np.int32)
    if threads > 1:
        cells = lib().abn_oracle_dtw_batch_mt(_p(feats1), _p(off1), _p(n1), _p(feats2),
                                              _p(off2), _p(n2), P, feats1.shape[1],
                                              _p(p1), _p(p2), _p(ln), path_stride, int(threads))
    else:
        cells = lib().abn_oracle_dtw_batch(_p(feats1), _p(off1), _p(n1), _p(feats2),
                                           _p(off2), _p(n2), P, feats1.shape[1],
                                           _p(p1), _p(p2), _p(ln), path_stride)
    return p1, p2, ln, cells


def dtw_path_py(d):
    """Pure-Python restatement of the same DP (small cases only); used by the
    tests to cross-check dtw.c."""
    d = np.asarray(d, dtype=np.float64)
    N, M = d.shape
    cost = np.zeros((N, M))
    for i in range(N):
        for j in range(M):
            if i == 0 and j == 0:
                cost[i, j] = d[0, 0]
            elif i == 0:
                cost[i, j] = d[i, j] + cost[i, j - 1]
            elif j == 0:
                cost[i, j] = d[i, j] + cost[i - 1, j]
            else:
                cost[i, j] = d[i, j] + min(cost[i - 1, j - 1], cost[i - 1, j],
                                           cost[i, j - 1])
    i, j = N - 1, M - 1
    p1, p2 = [i], [j]
    while i > 0 or j > 0:
        if i == 0:
            j -= 1
        elif j == 0:
            i -= 1
        else:
            k = int(np.argmin((cost[i - 1, j - 1], cost[i - 1, j], cost[i, j - 1])))
            if k == 0:
                i, j = i - 1, j - 1
            elif k == 1:
                i -= 1
            else:
                j -= 1
        p1.append(i)
        p2.append(j)
    return np.array(p1[::-1], dtype=np.int32), np.array(p2[::-1], dtype=np.int32)
