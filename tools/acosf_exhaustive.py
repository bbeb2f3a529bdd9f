#!/usr/bin/env python3
"""Compares oracle/dtw.c's restatement of glibc's acosf with this machine's libm for
EVERY float32 bit pattern of [-1, 1] plus the band just outside (|x| up to 1 + 2^-15)
and the NaN / inf patterns.  Prints the number of mismatches (expected: 0)."""
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(rng):
    from oracle import dtw_oracle as D
    return D.acosf_vs_libm(rng[0], rng[1], 1)


def main():
    ranges = []
    for base in (0x00000000, 0x80000000):
        top = base + 0x3f800100                 # 1 + 2^-15
        step = (top - base) // 16 + 1
        lo = base
        while lo <= top:
            ranges.append((lo, min(lo + step - 1, top)))
            lo += step
        ranges.append((base + 0x7f000000, base + 0x7fffffff))     # huge, inf, NaN
    t = time.time()
    with ProcessPoolExecutor(max_workers=8) as ex:
        res = list(ex.map(run, ranges))
    total = sum(hi - lo + 1 for lo, hi in ranges)
    bad = sum(r[0] for r in res)
    print('%d float32 arguments compared, %d mismatches (%.1f s)' % (total, bad, time.time() - t))
    for (lo, hi), (n, first) in zip(ranges, res):
        if n:
            print('  range %08x-%08x: %d mismatches, first at %08x' % (lo, hi, n, first))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
