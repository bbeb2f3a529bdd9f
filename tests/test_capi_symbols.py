"""The C-ABI shared library loads without a GPU and exports every symbol that
include/abnet3_hip.h declares; host-side sizing entry points answer sanely.
No kernel is launched here."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, 'include', 'abnet3_hip.h')


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(abn_[a-z0-9_]+)\s*\(', text)))


@pytest.fixture(scope='module')
def lib():
    from abnet3_amd import build, _lib
    build.build()                      # hipcc cross-compiles gfx950 without a GPU
    return _lib.load()


def test_header_declares_what_the_binding_binds(lib):
    from abnet3_amd import _lib
    assert set(declared_symbols()) == set(_lib.SYMBOLS), \
        set(declared_symbols()) ^ set(_lib.SYMBOLS)


def test_every_declared_symbol_is_exported(lib):
    raw = ctypes.CDLL(lib._name)
    for name in declared_symbols():
        assert hasattr(raw, name), name
    from abnet3_amd import _lib as binding
    assert lib.abn_abi_version() == binding.ABI_VERSION


def test_nothing_undeclared_is_exported(lib):
    """The shared object's dynamic symbol table holds no abn_* entry point that the header does not declare:
    diagnostic builds (-DABN_DTW_STAMPS, tools/variants.sh) add theirs, the product library must not."""
    import subprocess
    nm = '/opt/rocm/lib/llvm/bin/llvm-nm' if os.path.exists('/opt/rocm/lib/llvm/bin/llvm-nm') else 'nm'
    out = subprocess.run([nm, '-D', '--defined-only', lib._name], stdout=subprocess.PIPE, text=True, check=True).stdout
    exported = set()
    for line in out.splitlines():
        parts = line.split()
        if len(parts) >= 3 and parts[1] in 'TtWw' and parts[2].startswith('abn_'):
            exported.add(parts[2])
    assert exported, out[:400]
    assert exported == set(declared_symbols()), exported ^ set(declared_symbols())


def test_tower_path_is_a_pure_query(lib):
    """abn_tower_path answers from the descriptor alone (no launch, no GPU): the C2 tower takes the operand-plane
    chain in its own arithmetic, an odd-width f16x2 tower the per-layer GEMMs in bf16x3."""
    from abnet3_amd import _lib
    d = _lib.TowerDesc()
    d.n_layers, d.act, d.last_act, d.batch_norm, d.precision = 4, 1, 1, 0, _lib.PRECISION['f16x2']
    for i, v in enumerate((40, 500, 500, 500, 100)):
        d.dims[i] = v
    for l in range(4):
        d.W[l] = d.b[l] = d.dW[l] = d.db[l] = 0x1000
    prec = ctypes.c_int32(-1)
    a = ctypes.c_void_p(0x10000)
    assert lib.abn_tower_path(ctypes.byref(d), a, a, 8192, 2, 1, a, 0, ctypes.byref(prec)) == _lib.PATH_PLANES
    assert prec.value == _lib.PRECISION['f16x2']
    assert lib.abn_tower_path(ctypes.byref(d), a, a, 8192, 2, 1, a, 1, None) == _lib.PATH_PLANES
    assert lib.abn_tower_path(ctypes.byref(d), a, a, 512, 2, 1, a, 0, None) == _lib.PATH_WIDE
    d.dims[1] = 501                                   # not a multiple of 4: the GEMM kernels, in bf16 x 3
    assert lib.abn_tower_path(ctypes.byref(d), a, a, 8192, 2, 1, a, 0, ctypes.byref(prec)) == _lib.PATH_PER_LAYER
    assert prec.value == _lib.PRECISION['bf16x3']
    d.n_layers = 99
    assert lib.abn_tower_path(ctypes.byref(d), a, a, 8192, 2, 1, a, 0, None) < 0


def test_descriptor_layout_matches_header():
    from abnet3_amd import _lib
    # 4 x int32 + 17 x int64 + 11 arrays of 16 pointers + precision, d_out_is_dz, defer_reduce, wpack_valid,
    # forward_only, reserved + wpack + drop_seed + drop_p, reserved2 + bn_sync_world, reserved3 + bn_sync_fn + bn_sync_ctx + n_valid
    # + bn_nbt (16 pointers) + sync_ws (ABI v18)
    assert ctypes.sizeof(_lib.TowerDesc) == 16 + 17 * 8 + 11 * 16 * 8 + 24 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 16 * 8 + 8 + 8 + 8 + 8      # (+ fwd_ws, fwd_calls, source: ABI v19)


def test_host_side_sizing_and_argument_errors(lib):
    from abnet3_amd import _lib
    d = _lib.TowerDesc()
    d.n_layers, d.act, d.last_act, d.batch_norm = 4, 1, 1, 0
    for i, v in enumerate((40, 500, 500, 500, 100)):
        d.dims[i] = v
    for l in range(4):
        d.W[l] = 0x1000
        d.b[l] = 0x1000
    rows = 8192
    ws = lib.abn_tower_ws_floats(ctypes.byref(d), rows, 2)
    assert ws >= rows * (40 + 500 * 3 + 100)
    off = lib.abn_tower_out_offset(ctypes.byref(d), rows, 2)
    assert 0 < off < ws and off % 64 == 0
    assert lib.abn_tower_bwd_scratch_floats(ctypes.byref(d), rows) >= 16 * 571600
    assert lib.abn_tower_ws_floats(ctypes.byref(d), 8191, 2) == -1      # rows % n_calls
    assert b'divisible' in lib.abn_last_error()
    d.n_layers = 99
    assert lib.abn_tower_ws_floats(ctypes.byref(d), rows, 2) == -1
    assert lib.abn_pair_loss_ws_bytes(4096) == 512 * 8 + 8      # per-workgroup partial sums + the ticket counter
    assert lib.abn_linear_wgrad_scratch_floats(8192, 500, 500) == 16 * 250560
    # argument validation happens before any launch
    assert lib.abn_pair_loss(None, None, None, 2, 4, 4, 0, 0.5, 1, None, None, None, None, None) == -1
    assert lib.abn_stack_frames(None, 10, 40, 4, None, None) == -1      # even nframes
    assert b'odd' in lib.abn_last_error()
    assert lib.abn_fbank(None, 1, 100, 400, 160.0, 1000, 40, 0.97, None, None, None, 1, None, None) == -1


def test_dtw_workspace_planning(lib):
    import numpy as np
    n1 = np.array([300, 50, 0], dtype=np.int32)
    n2 = np.array([280, 600, 5], dtype=np.int32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    ws = lib.abn_dtw_ws_bytes(p(n1), p(n2), 3, 1000, 1000)
    # the cost matrix is never materialised: per (padded) cell only 2 bits of back-pointers
    # -- bands of 32 rows x ceil((n2 + 31) / 32) rounds of 32 diagonals, 16 diagonals to a
    # dword per row -- plus, for the two slots of each of the 2 workgroups, two band boundary rows
    # (float64: the longest token 2 rounded up to 32 + 64 of slack) and the token-2 norms (float32), and metadata
    dirs = (10 * 10 + 2 * 20) * 32 * 2 * 4
    bound = 4 * 2 * (608 + 64) * 8
    norms = 4 * (608 + 64) * 4
    assert dirs + bound + norms <= ws <= dirs + bound + norms + 9 * 256 + 3 * 64
    assert ws < 1.0 * (300 * 280 + 50 * 600)            # under 1 B per cell even for two pairs (the matrix alone was 4.25 B / cell)
    assert lib.abn_dtw_host_stage_bytes(p(n1), p(n2), 3) >= 3 * 56 + 3 * 4
    assert lib.abn_dtw_ws_bytes(None, None, 3, 0, 0) == -1


def test_the_library_is_renamed_into_place_with_its_digest():
    """abnet3_amd/build.py links beside the target and renames (ranks that find the library stale at the same moment must never
    dlopen a half-written file): after a build the recorded digest is the file's, nothing temporary is left, and the kernel
    trace bench.py reads its in-step figures from says which build it is of."""
    import os
    from abnet3_amd import build
    lib = build.build()
    assert build._digest(lib) == build._recorded_digest()
    assert not [f for f in os.listdir(os.path.dirname(lib)) if f.endswith('.tmp')]
    import bench
    assert bench._trace_binary(None) is None
    assert bench._trace_binary('profiles/r05_bench_kernel_stats.csv').startswith('unknown')
    assert 'library' in bench._trace_binary('profiles/r06_bench_kernel_stats.csv')
