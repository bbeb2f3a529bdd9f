"""Randomised differential test of the operand-plane kernels (tools/fuzz_planes.py): 150 random towers -- widths that are
multiples of 4 up to 512, 0..4 hidden layers, every activation, BatchNorm on a quarter of them, 1 .. 2 100 rows, one or two
forward_once calls, chains and layer-per-launch kernels -- in the default arithmetic and in bf16 x 3 against the exact-fp32
mode: embeddings to 5e-5 of the tensor's largest entry, every gradient entry to 2e-4 (ReLU towers: every gradient tensor to
3e-2 in norm -- a pre-activation at ~0 takes either side of relu' in different arithmetics).  Needs an MI355X: -m gpu."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))


def test_random_towers_agree_with_the_exact_fp32_mode(monkeypatch):
    monkeypatch.setenv('ABN_FUSED_MIN_ROWS', '0')
    import fuzz_planes
    bad, lines = fuzz_planes.run(150, seed=7, verbose=False)
    assert bad == 0, '\n'.join(l for l in lines if l.startswith('BAD'))
    paths = {int(l.split('path')[1].split('|')[0]) for l in lines}
    assert {2, 5, 6} <= paths          # the chains, BatchNorm's layer launches, the layer-per-launch kernels all ran
