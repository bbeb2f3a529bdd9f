"""Oracle DTW / cosine_distance / stack_fbanks / fbank against the golden
vectors and against each other.  CPU only."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import dtw_oracle as D
from oracle import features_np as F


@pytest.mark.parametrize('name', ['f32', 'one', 'zero', 'near'])
def test_cosine_distance_matches_reference(name):
    g = load_golden('cosdist.npz')
    d = D.cosine_distance(g[name + '.x'], g[name + '.y'])
    ref = g[name + '.d']
    assert d.dtype == np.float64 and d.shape == ref.shape
    # arccos is ill-conditioned at |cos| -> 1 (near-duplicate frames), so the
    # comparison is made on cos(pi d), where one float32 ulp of the dot product
    # is one ulp of the result; plus a plain bound away from the poles.
    assert np.abs(np.cos(np.pi * d) - np.cos(np.pi * ref)).max() < 6e-7
    far = np.abs(np.cos(np.pi * ref)) < 0.99
    assert np.abs(d - ref)[far].max() < 1e-6


def test_cosine_distance_zero_rows_and_nan_drop():
    g = load_golden('cosdist.npz')
    d = D.cosine_distance(g['zero.x'], g['zero.y'])
    assert (d[3] == np.where(np.arange(d.shape[1]) == 7, 0.0, 1.0)).all()
    assert (d[:, 7] == np.where(np.isin(np.arange(d.shape[0]), (3, 10)), 0.0, 1.0)).all()
    # identical rows push cos one ulp above 1 -> NaN -> the reference asserts
    # (fixture holds 'AssertionError'); the oracle must refuse the pair too
    assert str(g['pos.d']) == 'AssertionError'
    with pytest.raises(AssertionError):
        D.cosine_distance(g['pos.x'], g['pos.y'])


def test_acosf_restatement_equals_libm():
    """oracle/dtw.c restates glibc's acosf operation by operation; here every 257th
    float32 of [-1, 1] and of a band outside (tools/acosf_exhaustive.py runs ALL of
    them: 0 mismatches on this image's glibc 2.35)."""
    for lo, hi in ((0x00000000, 0x3f800100), (0x80000000, 0xbf800100),
                   (0x3effff00, 0x3f000100), (0x3f7fff00, 0x3f800000)):
        n, first = D.acosf_vs_libm(lo, hi, 257 if hi - lo > 0x10000 else 1)
        assert n == 0, hex(first)
    L = D.lib()
    xs = np.concatenate([np.linspace(-1, 1, 2001), [1e-9, -1e-9, 0.5, -0.5, 0.49999997, 1.0, -1.0]])
    got = np.array([L.abn_oracle_acosf(float(np.float32(x))) for x in xs])
    ref = np.arccos(xs.astype(np.float32).astype(np.float64))
    assert np.abs(got - ref).max() < 3e-7
    assert np.isnan(L.abn_oracle_acosf(1.0000001)) and np.isnan(L.abn_oracle_acosf(-1.0000001))


@pytest.mark.parametrize('Dm', [1, 3, 7, 8, 9, 13, 39, 40, 41, 127, 128, 129, 200, 280, 1000])
def test_row_norms_follow_numpy_summation_order(Dm):
    """x2 = np.sqrt(np.sum(x ** 2, axis=1)) (utils.py:43-44): numpy sums pairwise, and
    the order changes the last bit in most rows -- the restatement must reproduce
    numpy's own result exactly (numpy is the arbiter of its summation order)."""
    rng = np.random.default_rng(Dm)
    x = rng.standard_normal((64, Dm)).astype(np.float32)
    assert (D.row_norms(x) == np.sqrt(np.sum(x ** 2, axis=1))).all()


def _near_duplicate_pair(p):
    # the generator of tools/make_golden.py (near_duplicate_pair), restated
    rng = np.random.default_rng(5000 + p)
    n = 176
    x = rng.standard_normal((n, 40)).astype(np.float32)
    if p % 3 == 2:
        x = np.abs(x)
    eps = [0.0, 3e-4, 6e-4, 8e-4, 1e-3, 1.3e-3, 2e-3, 4e-3][p % 8]
    y = (x + np.float32(eps) * rng.standard_normal((n, 40)).astype(np.float32)).astype(np.float32)
    if p % 16 >= 8:
        y = (y * np.float32(1.0 + 0.37 * (p % 5))).astype(np.float32)
    return x, y


@pytest.mark.parametrize('name', ['big', 'bigpos', 'bigzero', 'bigdup', 'wide'])
def test_cosine_distance_bit_exact_vs_reference_on_libm_path(name):
    """G5L: the reference's cosine_distance run with numpy on its plain-libm path, on
    matrices that take OpenBLAS's regular sgemm kernel: every cell bit for bit."""
    g = load_golden('cosdist_libm.npz')
    d, bad = D.cosine_distance(g[name + '.x'], g[name + '.y'], check=False)
    ref = g[name + '.d32']
    if (name + '.dropped') in g:
        assert bad                      # the reference raised AssertionError
        return
    assert not bad
    d32 = d.astype(np.float32)
    assert (d32.astype(np.float64) == d).all()
    assert (d32.view(np.uint32) == ref.view(np.uint32)).all()


def test_drop_decisions_agree_with_reference_pair_by_pair():
    """48 near-duplicate token pairs: which ones the reference drops (cos rounds above 1
    -> arccos NaN -> AssertionError, utils.py:59, dataloader.py:188-191) and, for the
    kept ones, every distance bit (sha256 of the float32 matrix)."""
    import hashlib
    g = load_golden('cosdist_libm.npz')
    drop, sha = g['near.dropped'], g['near.sha256']
    assert 10 < drop.sum() < len(drop) - 10
    for p in range(len(drop)):
        x, y = _near_duplicate_pair(p)
        assert [float(x.astype(np.float64).sum()), float(y.astype(np.float64).sum())] == list(g['near.in_chk'][p])
        d, bad = D.cosine_distance(x, y, check=False)
        assert bad == bool(drop[p]), p
        if not bad:
            assert hashlib.sha256(d.astype(np.float32).tobytes()).hexdigest() == str(sha[p]), p


def test_cosine_distance_float64_inputs():
    """utils.py:41-42: float64 inputs are computed in float64 (G5 `f64`)."""
    g = load_golden('cosdist.npz')
    d = D.cosine_distance(g['f64.x'], g['f64.y'])
    assert d.dtype == np.float64
    assert np.abs(d - g['f64.d']).max() < 1e-15
    with pytest.raises(AssertionError):
        D.cosine_distance(g['f64.x'], g['f64.y'].astype(np.float32))


def test_dtw_c_matches_python_restatement():
    rng = np.random.default_rng(0)
    for t in range(60):
        N, M = rng.integers(1, 40, 2)
        d = rng.random((N, M))
        if t % 3 == 0:
            d = np.round(d * 4) / 4          # force ties
        if t % 7 == 0:
            d[:] = 0.25                      # all ties: pure diagonal-first
        a = D.dtw_path(d)
        b = D.dtw_path_py(d)
        assert (a[0] == b[0]).all() and (a[1] == b[1]).all()
        # path invariants the call site relies on (utils.py:151-153)
        assert a[0][0] == 0 and a[1][0] == 0 and a[0][-1] == N - 1 and a[1][-1] == M - 1
        assert max(N, M) <= len(a[0]) <= N + M - 1
        s1, s2 = np.diff(a[0]), np.diff(a[1])
        assert ((s1 >= 0) & (s1 <= 1) & (s2 >= 0) & (s2 <= 1) & (s1 + s2 >= 1)).all()


def test_dtw_tie_break_is_diagonal_then_up_then_left():
    d = np.zeros((3, 3))
    p1, p2 = D.dtw_path(d)
    assert list(p1) == [0, 1, 2] and list(p2) == [0, 1, 2]
    d = np.zeros((2, 4))
    p1, p2 = D.dtw_path(d)
    # from (1,3): diag -> (0,2), then row 0 can only go left
    assert list(zip(p1, p2)) == [(0, 0), (0, 1), (0, 2), (1, 3)]
    d = np.zeros((4, 2))
    p1, p2 = D.dtw_path(d)
    assert list(zip(p1, p2)) == [(0, 0), (1, 0), (2, 0), (3, 1)]


def test_dtw_identical_sequences_follow_diagonal():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((25, 40)).astype(np.float32)
    y = x + 1e-2 * rng.standard_normal((25, 40)).astype(np.float32)
    p1, p2 = D.get_dtw_alignment(x, y)
    assert (p1 == np.arange(25)).all() and (p2 == np.arange(25)).all()


def test_dtw_batch_front_end():
    rng = np.random.default_rng(2)
    n1 = np.array([5, 9, 1, 12], dtype=np.int32)
    n2 = np.array([7, 3, 1, 12], dtype=np.int32)
    f1 = rng.standard_normal((n1.sum(), 40)).astype(np.float32)
    f2 = rng.standard_normal((n2.sum(), 40)).astype(np.float32)
    o1 = np.concatenate(([0], np.cumsum(n1)[:-1]))
    o2 = np.concatenate(([0], np.cumsum(n2)[:-1]))
    p1, p2, ln, cells = D.dtw_batch(f1, o1, n1, f2, o2, n2, 32)
    assert cells == int((n1.astype(int) * n2).sum())
    for p in range(4):
        a, b = D.get_dtw_alignment(f1[o1[p]:o1[p] + n1[p]], f2[o2[p]:o2[p] + n2[p]])
        assert ln[p] == len(a)
        assert (p1[p, :ln[p]] == a).all() and (p2[p, :ln[p]] == b).all()


@pytest.mark.parametrize('name', ['t100_n7', 't100_n3', 't5_n7', 't2_n7', 't17_n1'])
def test_stack_fbanks_matches_reference(name):
    g = load_golden('stack.npz')
    out = F.stack_fbanks(g[name + '.in'], int(g[name + '.n']))
    assert out.dtype == g[name + '.out'].dtype
    assert out.shape == g[name + '.out'].shape
    assert (out == g[name + '.out']).all()       # pure data movement: bit-exact


def test_fbank_shape_and_sanity():
    fs = 16000
    rng = np.random.default_rng(3)
    t = np.arange(fs) / fs
    sig = (3000 * np.sin(2 * np.pi * 1000 * t) + 50 * rng.standard_normal(fs)).astype(np.int16)
    fb = F.fbank(sig, fs)
    assert fb.shape == (101, 40) and fb.dtype == np.float32
    bank = F.mel_filterbank(fs)
    assert bank.shape == (513, 40)
    # every filter is a non-empty triangle and 1 kHz lands in the right band
    assert (bank.sum(axis=0) > 0).all()
    peak_band = np.argmax(bank[int(round(1000 / (fs / 1024)))])
    assert np.argmax(fb[10:90].mean(axis=0)) == peak_band
    # silence hits the 1e-5 floor
    z = F.fbank(np.zeros(4000, dtype=np.int16), fs)
    assert np.allclose(z, np.log(1e-5))
    # float32 pipeline stays close to the float64 definition
    fb32 = F.fbank(sig, fs, dtype=np.float32)
    assert np.abs(fb32 - fb).max() < 1e-4


def test_deltas_definition():
    """9-tap regression slope: exact on a linear ramp away from the edges, zero on a
    constant, the padding uses frame 1 / frame T-2."""
    T, Dm = 30, 3
    ramp = np.arange(T, dtype=np.float32)[:, None] * np.array([1.0, -2.0, 0.5], dtype=np.float32)
    d = F.deltas(ramp)
    assert np.allclose(d[4:-4], [1.0, -2.0, 0.5], atol=1e-6)
    assert np.allclose(F.deltas(np.ones((10, 2), dtype=np.float32)), 0.0)
    x = np.random.default_rng(0).standard_normal((12, 2)).astype(np.float32)
    d = F.deltas(x)
    manual0 = sum(n * (x[n] - x[1]) for n in range(1, 5)) / 60.0       # t = 0: every x[t-n] is the pad = x[1]
    assert np.allclose(d[0], manual0, atol=1e-6)
    dd = F.fbank_with_deltas(np.zeros(4000, dtype=np.int16), 16000)
    assert dd.shape == (26, 120) and np.allclose(dd[:, 40:], 0.0)


@pytest.mark.parametrize('per_channel', [1, 0])
@pytest.mark.parametrize('name', ['global', 'per_file', 'global_vad', 'per_file_vad'])
def test_mvn_restatement_on_the_reference_tests_cases(name, per_channel):
    """G10: oracle/features_np.mvn + the host-side VAD windows (utils.read_vad_file, Features_Accessor) on the literal
    cases of the reference's test/test_features.py:37-281 -- kept frames, statistics and outputs as that test asserts
    them (the fixture's kept-frame counts come from the reference's own filter_vad_* functions)."""
    import os
    import tempfile
    from abnet3_amd.utils import read_vad_file, Features_Accessor
    g = load_golden('mvn.npz')
    items = ['file1', 'file2']
    feats = [g['%s.x%d' % (name, i)].astype(np.float64) for i in range(2)]
    times = [g['%s.t%d' % (name, i)] for i in range(2)]
    vad = None
    if str(g[name + '.vad']):
        with tempfile.NamedTemporaryFile('w', suffix='.vad', delete=False) as fh:
            fh.write(str(g[name + '.vad']))
        vad = read_vad_file(fh.name)
        os.unlink(fh.name)
        assert vad == {'file1': [[0.0025, 0.5], [0.7525, 1.0]]}
    keep = []
    for f, x, t in zip(items, feats, times):
        if vad is not None and f in vad:
            rows = np.concatenate([Features_Accessor.get_indices_between(t, s, e) for s, e in vad[f]])
            keep.append(x[rows])
        else:
            keep.append(x)
    tag = '%s.pc%d' % (name, per_channel)
    if int(g[name + '.per_file']):
        for i, x in enumerate(feats):
            assert keep[i].shape[0] == int(g['%s.kept%d' % (tag, i)])
            out, mean, std = F.mvn(x, bool(per_channel), stats_on=keep[i])
            assert np.array_equal(np.atleast_1d(mean), g['%s.mean%d' % (tag, i)])
            assert np.array_equal(np.atleast_1d(std), g['%s.std%d' % (tag, i)])
            assert np.array_equal(out, g['%s.out%d' % (tag, i)])
    else:
        allf = np.vstack(keep)
        assert allf.shape[0] == int(g[tag + '.kept'])
        for i, x in enumerate(feats):
            out, mean, std = F.mvn(x, bool(per_channel), stats_on=allf)
            assert np.allclose(np.atleast_1d(mean), g[tag + '.mean'], rtol=1e-15)
            assert np.allclose(np.atleast_1d(std), g[tag + '.std'], rtol=1e-15)
            assert np.allclose(out, g['%s.out%d' % (tag, i)], rtol=1e-14, atol=1e-15)
