#!/usr/bin/env python3
"""abn_arccos_f32 (the HIP cell function's acosf) against the oracle's restatement of
glibc's acosf for EVERY float32 of [-1, 1], the band outside up to 1 + 2^-15, and the
huge / inf / NaN patterns.  Run on the GPU box; prints the mismatch count (expected 0)."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abnet3_amd import _lib              # noqa: E402
from oracle import dtw_oracle as O       # noqa: E402


def main():
    lib, L = _lib.load(), O.lib()
    L.abn_oracle_acosf_array.restype = None
    L.abn_oracle_acosf_array.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
    ranges = []
    for base in (0x00000000, 0x80000000):
        ranges += [(base, base + 0x3f800100), (base + 0x7f000000, base + 0x7fffffff)]
    chunk = 1 << 26
    bad = total = 0
    t0 = time.time()
    for over_pi in (0, 1, 2, 3):        # bit 1: the straight-line form of the first range where it applies
      for lo, hi in ranges:
        if over_pi & 2:
            lo, hi = max(lo, (lo & 0x80000000) + 0x32000000), min(hi, (lo & 0x80000000) + 0x3f000100)     # (around the range)
            if lo > hi:
                continue
        for s in range(lo, hi + 1, chunk):
            e = min(s + chunk, hi + 1)
            bits = np.arange(s, e, dtype=np.int64).astype(np.uint32)
            x = bits.view(np.float32)
            xd = torch.from_numpy(x).cuda()
            out = torch.empty_like(xd)
            _lib.check(lib.abn_arccos_f32(_lib.ptr(xd), xd.numel(), over_pi, _lib.ptr(out), _lib.stream()), 'abn_arccos_f32')
            ref = np.empty_like(x)
            L.abn_oracle_acosf_array(x.ctypes.data_as(ctypes.c_void_p), len(x), over_pi & 1, ref.ctypes.data_as(ctypes.c_void_p))
            got = out.cpu().numpy()
            nan = np.isnan(ref)
            m = (np.isnan(got) != nan) | ((got.view(np.uint32) != ref.view(np.uint32)) & ~nan)
            if m.any() and bad == 0:
                i = int(np.nonzero(m)[0][0])
                print('first mismatch: x bits %08x  device %08x  libm %08x' % (bits[i], got.view(np.uint32)[i], ref.view(np.uint32)[i]))
            bad += int(m.sum())
            total += len(x)
    print('%d evaluations (acosf and acosf / pi) compared on the device, %d mismatches (%.1f s)' % (total, bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
