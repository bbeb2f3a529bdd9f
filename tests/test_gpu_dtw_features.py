"""Parity of the HIP DTW / distance / gather / stacking / filterbank kernels
(through the C-ABI) with the C / numpy oracle and the golden vectors.
DTW path indices must be BIT-EXACT.  Needs an MI355X: run with -m gpu."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def synth_pairs(rng, P, lo=5, hi=90, D=40, warp=True):
    n1 = rng.integers(lo, hi, P).astype(np.int32)
    n2 = rng.integers(lo, hi, P).astype(np.int32)
    f1 = rng.standard_normal((int(n1.sum()), D)).astype(np.float32)
    o1 = np.concatenate(([0], np.cumsum(n1)[:-1])).astype(np.int64)
    o2 = np.concatenate(([0], np.cumsum(n2)[:-1])).astype(np.int64)
    f2 = rng.standard_normal((int(n2.sum()), D)).astype(np.float32)
    if warp:    # second token = time-warped copy of the first + noise
        for p in range(P):
            src = np.rint(np.linspace(0, n1[p] - 1, n2[p])).astype(int)
            f2[o2[p]:o2[p] + n2[p]] = f1[o1[p] + src] + 0.1 * f2[o2[p]:o2[p] + n2[p]]
    return f1, o1, n1, f2, o2, n2


@pytest.mark.parametrize('name', ['f32', 'one', 'zero', 'near'])
def test_cosine_distance_matches_reference_and_oracle(name):
    from abnet3_amd.utils import cosine_distance
    from oracle import dtw_oracle as O
    g = load_golden('cosdist.npz')
    d = cosine_distance(g[name + '.x'], g[name + '.y'])
    assert d.dtype == np.float64
    assert (d == O.cosine_distance(g[name + '.x'], g[name + '.y'])).all()      # bit-exact
    # this fixture was produced with numpy's AVX512 dispatch (SVML arccos, OpenBLAS's
    # small-matrix sgemm): comparable to a few ulp of the cosine only
    ref = g[name + '.d']
    assert np.abs(np.cos(np.pi * d) - np.cos(np.pi * ref)).max() < 6e-7


@pytest.mark.parametrize('name', ['big', 'bigpos', 'bigzero', 'bigdup', 'wide'])
def test_cosine_distance_bit_exact_vs_reference_libm_path(name):
    """G5L: outputs of the reference's cosine_distance on numpy's plain-libm path
    (regular sgemm kernel): the HIP distance reproduces every cell bit for bit."""
    from abnet3_amd.utils import cosine_distance
    g = load_golden('cosdist_libm.npz')
    if (name + '.dropped') in g:
        with pytest.raises(AssertionError):
            cosine_distance(g[name + '.x'], g[name + '.y'])
        return
    d = cosine_distance(g[name + '.x'], g[name + '.y'])
    d32 = d.astype(np.float32)
    assert (d32.astype(np.float64) == d).all()
    assert (d32.view(np.uint32) == g[name + '.d32'].view(np.uint32)).all()


def test_drop_decisions_agree_with_reference_pair_by_pair():
    """48 near-duplicate token pairs through abn_dtw_batched (the tiled MFMA distance):
    path_len == 0 exactly for the pairs the reference's cosine_distance refuses, and the
    standalone distance of every kept pair equals the reference's, bit for bit."""
    import hashlib
    from abnet3_amd.utils import cosine_distance, dtw_align_batch
    from test_oracle_dtw_features import _near_duplicate_pair
    g = load_golden('cosdist_libm.npz')
    drop, sha = g['near.dropped'], g['near.sha256']
    xs, ys = zip(*[_near_duplicate_pair(p) for p in range(len(drop))])
    f1 = torch.from_numpy(np.concatenate(xs)).cuda()
    f2 = torch.from_numpy(np.concatenate(ys)).cuda()
    n = np.array([len(x) for x in xs], dtype=np.int32)
    off = np.concatenate(([0], np.cumsum(n)[:-1]))
    res = dtw_align_batch(f1, off, n, f2, off, n)
    ln = res.path_len.cpu().numpy()
    assert ((ln == 0) == drop).all(), np.nonzero((ln == 0) != drop)[0]
    for p in np.nonzero(~drop)[0]:
        d = cosine_distance(xs[p], ys[p])
        assert hashlib.sha256(d.astype(np.float32).tobytes()).hexdigest() == str(sha[p]), p


def test_cosine_distance_float64_inputs():
    """utils.py:41-42: float64 inputs are computed in float64 (G5 `f64`); mixed
    precisions are refused like in the reference."""
    from abnet3_amd.utils import cosine_distance
    g = load_golden('cosdist.npz')
    d = cosine_distance(g['f64.x'], g['f64.y'])
    assert d.dtype == np.float64
    assert np.abs(d - g['f64.d']).max() < 2e-15
    with pytest.raises(AssertionError):
        cosine_distance(g['f64.x'], g['f64.y'].astype(np.float32))


def test_device_arccos_equals_libm_acosf():
    """abn_arccos_f32 (the cell function's acosf) against the oracle's restatement of
    glibc's acosf -- itself compared with libm for every float32 -- on every 61st
    float32 of [-1, 1], all of the three range boundaries, and arguments outside."""
    from abnet3_amd import _lib
    from oracle import dtw_oracle as O
    lib = _lib.load()
    L = O.lib()
    bits = np.concatenate([np.arange(0, 0x3f800100, 61, dtype=np.int64),
                           np.arange(0x3effff00, 0x3f000100, dtype=np.int64),
                           np.arange(0x3f7ffe00, 0x3f800100, dtype=np.int64),
                           np.arange(0x32700000, 0x32900000, 997, dtype=np.int64)])
    bits = np.concatenate([bits, bits + 0x80000000]).astype(np.uint32)
    x = bits.view(np.float32)
    xd = torch.from_numpy(x.copy()).cuda()
    import ctypes
    L.abn_oracle_acosf_array.restype = None
    L.abn_oracle_acosf_array.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
    # acosf(x), and acosf(x) / float32(pi) as utils.py:53 divides; bit 1: the straight-line statements of the first range
    # (2^-26 < |x| < 0.5) the gang kernel takes when a wavefront's cells all lie in it -- the same bits
    for over_pi in (0, 1, 2, 3):
        out = torch.empty_like(xd)
        _lib.check(lib.abn_arccos_f32(_lib.ptr(xd), xd.numel(), over_pi, _lib.ptr(out), _lib.stream()), 'abn_arccos_f32')
        got = out.cpu().numpy()
        ref = np.empty_like(x)
        L.abn_oracle_acosf_array(x.ctypes.data_as(ctypes.c_void_p), len(x), over_pi & 1, ref.ctypes.data_as(ctypes.c_void_p))
        nan = np.isnan(ref)
        assert (np.isnan(got) == nan).all()
        assert (got.view(np.uint32)[~nan] == ref.view(np.uint32)[~nan]).all()


@pytest.mark.parametrize('D', [40, 280, 13])
def test_dtw_paths_bit_exact_vs_oracle(D):
    from abnet3_amd.utils import dtw_align_batch
    from oracle import dtw_oracle as O
    rng = np.random.default_rng(D)
    f1, o1, n1, f2, o2, n2 = synth_pairs(rng, 96, D=D)
    # edge cases: 1-frame tokens, a long one, an exact tile multiple
    n1[0], n2[0] = 1, 1
    n1[1], n2[1] = 1, 40
    n1[2], n2[2] = 64, 64
    res = dtw_align_batch(torch.from_numpy(f1).cuda(), o1, n1, torch.from_numpy(f2).cuda(), o2, n2)
    got = res.to_lists()
    cost = res.total_cost.cpu().numpy()
    for p in range(len(n1)):
        a = f1[o1[p]:o1[p] + n1[p]]
        b = f2[o2[p]:o2[p] + n2[p]]
        d = O.cosine_distance(a, b)
        p1, p2 = O.dtw_path(d)
        assert got[p] is not None
        assert (got[p][0] == p1).all() and (got[p][1] == p2).all(), p
        assert cost[p] == O.dtw_cost(d), p                        # float64 DP: bit-exact too


def test_dtw_ties_and_long_tokens():
    from abnet3_amd.utils import dtw_align_batch
    from oracle import dtw_oracle as O
    rng = np.random.default_rng(5)
    # quantised features => many exactly equal distances => tie-break matters
    n1 = np.array([300, 600, 1000, 17, 128], dtype=np.int32)
    n2 = np.array([280, 590, 50, 1000, 129], dtype=np.int32)
    f1 = rng.integers(-2, 3, (int(n1.sum()), 8)).astype(np.float32)
    f2 = rng.integers(-2, 3, (int(n2.sum()), 8)).astype(np.float32)
    f1[f1.sum(1) == 0] += 1
    o1 = np.concatenate(([0], np.cumsum(n1)[:-1]))
    o2 = np.concatenate(([0], np.cumsum(n2)[:-1]))
    got = dtw_align_batch(torch.from_numpy(f1).cuda(), o1, n1, torch.from_numpy(f2).cuda(), o2, n2).to_lists()
    for p in range(len(n1)):
        a, b = f1[o1[p]:o1[p] + n1[p]], f2[o2[p]:o2[p] + n2[p]]
        try:
            ref = O.get_dtw_alignment(a, b)
        except AssertionError:
            assert got[p] is None
            continue
        assert (got[p][0] == ref[0]).all() and (got[p][1] == ref[1]).all(), p


def test_dtw_tokens_longer_than_1024_frames():
    """The reference aligns tokens of any length (abnet3/utils.py:147-153).  Round 1 refused a
    whole batch when one token exceeded 1024 frames; the banded kernel has no limit: 1500 x 1100
    and 40 x 2300 next to ordinary pairs, both frame widths' kernels, bit-exact vs the oracle."""
    from abnet3_amd.utils import dtw_align_batch
    from oracle import dtw_oracle as O
    for D in (40, 24):
        rng = np.random.default_rng(1500 + D)
        n1 = np.array([1500, 40, 70, 1025, 33], dtype=np.int32)
        n2 = np.array([1100, 2300, 65, 31, 1024], dtype=np.int32)
        f1 = rng.standard_normal((int(n1.sum()), D)).astype(np.float32)
        f2 = rng.standard_normal((int(n2.sum()), D)).astype(np.float32)
        o1 = np.concatenate(([0], np.cumsum(n1)[:-1]))
        o2 = np.concatenate(([0], np.cumsum(n2)[:-1]))
        src = np.rint(np.linspace(0, n1[0] - 1, n2[0])).astype(int)        # pair 0: a warped noisy copy
        f2[:n2[0]] = f1[src] + 0.1 * f2[:n2[0]]
        res = dtw_align_batch(torch.from_numpy(f1).cuda(), o1, n1, torch.from_numpy(f2).cuda(), o2, n2)
        got, cost = res.to_lists(), res.total_cost.cpu().numpy()
        for p in range(len(n1)):
            d = O.cosine_distance(f1[o1[p]:o1[p] + n1[p]], f2[o2[p]:o2[p] + n2[p]])
            q1, q2 = O.dtw_path(d)
            assert got[p] is not None and np.array_equal(got[p][0], q1) and np.array_equal(got[p][1], q2), (D, p)
            assert cost[p] == O.dtw_cost(d), (D, p)


def test_dtw_dropped_pair_and_empty_token():
    from abnet3_amd.utils import dtw_align_batch
    g = load_golden('cosdist.npz')
    x, y = g['pos.x'], g['pos.y']            # holds identical rows -> NaN -> dropped
    f = torch.from_numpy(np.concatenate([x, y])).cuda()
    res = dtw_align_batch(f, [0, 0, 0], [30, 30, 0], f, [30, 40, 30], [30, 10, 5])
    ln = res.path_len.cpu().numpy()
    assert ln[0] == 0 and ln[2] == 0 and ln[1] > 0


def test_load_frames_from_pairs_matches_reference_fixture():
    """G7: OriginalDataLoader.load_frames_from_pairs on the toy batch (the
    fixture was produced by the reference loader with the oracle DTW)."""
    from abnet3_amd.dataloader import OriginalDataLoader
    from abnet3_amd.utils import group_pairs
    g = load_golden('frames.npz')
    feats = {k[5:]: v for k, v in g.items() if k.startswith('feat.')}
    times = {k: np.arange(len(v)) * 0.01 + 0.0025 for k, v in feats.items()}

    def parse(line):
        t = str(line).split(' ')
        return (t[0], float(t[1]), float(t[2]), t[3], float(t[4]), float(t[5]), t[6])
    pairs = [parse(l) for l in g['pairs']]
    empty = parse(g['pairs_empty'])
    for align in (0, 1):
        dl = OriginalDataLoader('unused', 'unused', align_different_words=bool(align))
        dl.set_data(feats, times)
        X1, X2, Y = dl.load_frames_from_pairs(group_pairs(pairs + ([] if align else [empty])))
        assert Y.dtype == np.float64 and X1.dtype == np.float32
        assert (X1 == g['align%d.X1' % align]).all()
        assert (X2 == g['align%d.X2' % align]).all()
        assert (Y == g['align%d.Y' % align]).all()


def test_frames_dataloader_batches_and_trainer_epoch(tmp_path):
    """FramesDataLoader (exact frame-pair batches) driving TrainerSiamese for
    two epochs: losses finite, decreasing machinery runs, best model saved."""
    from abnet3_amd.dataloader import FramesDataLoader
    from abnet3_amd.model import SiameseNetwork
    from abnet3_amd.loss import coscos2
    from abnet3_amd.trainer import TrainerSiamese
    rng = np.random.default_rng(3)
    feats = {'u%d' % i: rng.standard_normal((400, 40)).astype(np.float32) for i in range(4)}
    times = {k: np.arange(400) * 0.01 + 0.0025 for k in feats}
    pairs = []
    for i in range(40):
        a, b = rng.integers(0, 4, 2)
        s1, s2 = rng.uniform(0, 3.0, 2).round(2)
        pairs.append(('u%d' % a, s1, s1 + 0.4, 'u%d' % b, s2, s2 + 0.5, 'same' if i % 2 else 'diff'))
    dl = FramesDataLoader('unused', 'unused', batch_size=256)
    dl.set_data(feats, times, pairs[:30], pairs[30:])
    np.random.seed(0)
    batches = list(dl.batch_iterator(True))
    assert all(b[0].shape == (256, 40) and b[0].is_cuda for b in batches)
    assert batches[0][2].dtype == torch.int64
    torch.manual_seed(0)
    net = SiameseNetwork(input_dim=40, num_hidden_layers=1, hidden_dim=64, output_dim=32,
                         p_dropout=0.0, activation_layer='sigmoid',
                         output_path=str(tmp_path / 'network'))
    tr = TrainerSiamese(network=net, loss=coscos2(avg=False), num_epochs=2, patience=5,
                        optimizer_type='adadelta', lr=0.1, dataloader=dl,
                        log_dir=str(tmp_path / 'runs'))
    tr.train()
    assert len(tr.train_losses) == 3 and np.isfinite(tr.train_losses).all()
    assert (tmp_path / 'network.pth').exists()
    sd = torch.load(str(tmp_path / 'network.pth'))
    assert list(sd.keys()) == ['input_emb.0.weight', 'input_emb.0.bias',
                               'hidden_layers.0.weight', 'hidden_layers.0.bias',
                               'output_layer.0.weight', 'output_layer.0.bias']


@pytest.mark.parametrize('name', ['t100_n7', 't100_n3', 't5_n7', 't2_n7', 't17_n1'])
def test_stack_fbanks_matches_reference(name):
    from abnet3_amd.features import FeaturesGenerator
    g = load_golden('stack.npz')
    out = FeaturesGenerator().stack_fbanks(g[name + '.in'], int(g[name + '.n']))
    assert out.dtype == g[name + '.out'].dtype and (out == g[name + '.out']).all()


@pytest.mark.parametrize('fs,n', [(16000, 16000), (16000, 12345), (22050, 3999), (16000, 150)])
def test_fbank_matches_oracle(fs, n):
    from abnet3_amd.features import FeaturesGenerator
    from oracle import features_np as F
    rng = np.random.default_rng(n)
    t = np.arange(n) / fs
    sig = (2000 * np.sin(2 * np.pi * 700 * t) + 900 * np.sin(2 * np.pi * 2300 * t + 1.0)
           + 200 * rng.standard_normal(n)).astype(np.int16)
    fg = FeaturesGenerator()
    fb = fg.fbank_from_samples(sig, fs).cpu().numpy()
    ref = F.fbank(sig, fs)
    assert fb.shape == ref.shape == (F.frame_count(n, fs), 40) and fb.dtype == np.float32
    # log energies ~ 10..25; fp32 FFT against the fp64 definition: 5e-5 absolute
    assert np.abs(fb - ref).max() < 5e-5, np.abs(fb - ref).max()
    # float input path and silence floor
    fbf = fg.fbank_from_samples(sig.astype(np.float32), fs).cpu().numpy()
    assert np.abs(fbf - fb).max() < 1e-5
    z = fg.fbank_from_samples(np.zeros(2000, dtype=np.int16), fs).cpu().numpy()
    assert np.allclose(z, np.log(1e-5))


@pytest.mark.parametrize('fs,nfilt', [(16000, 13), (14000, 40), (16000, 64)])
def test_fbank_wide_and_narrow_filters(fs, nfilt):
    """Bands wider than the sparse projection's 20 pieces of four bins (few filters) take the dense projection inside the same
    kernel; another sampling rate moves every band; 64 filters fill every lane."""
    from abnet3_amd.features import FeaturesGenerator
    from oracle import features_np as F
    n = 9000
    rng = np.random.default_rng(nfilt)
    t = np.arange(n) / fs
    sig = (1500 * np.sin(2 * np.pi * 500 * t) + 700 * np.sin(2 * np.pi * 1900 * t + 0.5) + 300 * rng.standard_normal(n)).astype(np.int16)
    fb = FeaturesGenerator(n_filters=nfilt).fbank_from_samples(sig, fs).cpu().numpy()
    ref = F.fbank(sig, fs, nfilt=nfilt)
    assert fb.shape == ref.shape == (F.frame_count(n, fs), nfilt)
    assert np.abs(fb - ref).max() < 5e-5, np.abs(fb - ref).max()


def test_fbank_general_kernel_and_deltas():
    """nfft != 1024 takes the general kernel; deltas / deltasdeltas (features.py:110-111)
    append the 9-tap regression slopes of oracle/features_np.py as further columns."""
    from abnet3_amd.features import FeaturesGenerator
    from oracle import features_np as F
    fs, n = 16000, 9000
    rng = np.random.default_rng(4)
    t = np.arange(n) / fs
    sig = (1500 * np.sin(2 * np.pi * 440 * t) + 300 * rng.standard_normal(n)).astype(np.int16)
    fg = FeaturesGenerator()
    for nfft in (512, 2048):
        fb = fg.fbank_from_samples(sig, fs, nfft=nfft).cpu().numpy()
        assert np.abs(fb - F.fbank(sig, fs, nfft=nfft)).max() < 2e-4
    for d, dd in ((True, False), (True, True), (False, True)):
        fg = FeaturesGenerator(deltas=d, deltasdeltas=dd)
        got = fg.fbank_from_samples(sig, fs).cpu().numpy()
        ref = F.fbank_with_deltas(sig, fs, do_deltas=d, do_deltasdeltas=dd)
        assert got.shape == ref.shape == (F.frame_count(n, fs), 40 * (1 + d + dd))
        assert np.abs(got - ref).max() < 5e-5
    # two-frame and one-frame inputs: the padding rule needs no neighbour that does not exist
    for m in (1, 170):
        short = fg.fbank_from_samples(sig[:m], fs).cpu().numpy()
        assert np.abs(short - F.fbank_with_deltas(sig[:m], fs, do_deltas=False, do_deltasdeltas=True)).max() < 5e-5


def test_embedder_matches_eval_forward():
    from abnet3_amd.embedder import EmbedderSiamese
    from abnet3_amd.model import SiameseNetwork
    g = load_golden('tower_sig.npz')
    import ast
    net = SiameseNetwork(**ast.literal_eval(str(g['kw'])))
    net.load_state_dict({k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('p.')})
    emb = EmbedderSiamese(network=net, network_path=None, feature_path=None, output_path=None,
                          batch_size=10)
    out = emb.embed_features([g['x1'], g['x2'].astype(np.float64), g['x1'][:0]])
    assert np.abs(out[0] - g['eval_e1']).max() < 1e-5 * np.abs(g['eval_e1']).max()
    assert np.abs(out[1] - g['eval_e2']).max() < 1e-5 * np.abs(g['eval_e2']).max()
    assert out[2].shape == (0, 50)
    with pytest.raises(ValueError):
        EmbedderSiamese(network=None)


@pytest.mark.parametrize('per_channel', [True, False])
@pytest.mark.parametrize('per_file', [True, False])
def test_mean_variance_normalisation(per_channel, per_file):
    """features.py:205-297: global / per-file, per-channel / whole-spectrum,
    with and without VAD, against the numpy expressions the reference uses."""
    from abnet3_amd.features import FeaturesGenerator
    from oracle import features_np as F
    rng = np.random.default_rng(9)
    feats = {'a': (rng.standard_normal((700, 40)) * 3 + 12).astype(np.float32),
             'b': (rng.standard_normal((5000, 40)) * 2 + 9).astype(np.float32),
             'c': np.full((30, 40), 4.0, dtype=np.float32)}          # constant: std = 0 -> eps matters
    times = {k: np.arange(len(v)) * 0.01 + 0.0025 for k, v in feats.items()}
    vad = {'a': [[0.5, 2.0], [3.0, 4.5]], 'b': [[10.0, 30.0]]}
    fg = FeaturesGenerator(norm_per_file=per_file, norm_per_channel=per_channel)
    for use_vad in (False, True):
        out, stats = fg.normalize_features(feats, times, vad if use_vad else None)
        if per_file:
            for f, x in feats.items():
                sel = x
                if use_vad and f in vad:
                    sel = x[FeaturesGenerator._vad_rows(times[f], vad[f])]
                ref, m, s = F.mvn(x, per_channel, stats_on=sel)
                # constant file: ref is 0/eps-scaled noise; compare with an absolute bound there
                tol = 2e-5 * max(1.0, np.abs(ref).max())
                assert np.abs(out[f] - ref).max() <= tol, (f, use_vad)
        else:
            parts = [x[FeaturesGenerator._vad_rows(times[f], vad[f])] if (use_vad and f in vad) else x
                     for f, x in feats.items()]
            allf = np.vstack(parts)
            for f, x in feats.items():
                ref, m, s = F.mvn(x, per_channel, stats_on=allf)
                assert np.abs(out[f] - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), (f, use_vad)
            assert np.allclose(stats[0], np.mean(allf, axis=0 if per_channel else None), rtol=1e-5)
            assert np.allclose(stats[1], np.std(allf, axis=0 if per_channel else None), rtol=1e-5)
    if not per_file:       # load_mean_variance path
        params = {'mean': np.full(40, 10.0) if per_channel else 10.0,
                  'variance': np.full(40, 2.0) if per_channel else 2.0}
        out, _ = fg.normalize_features(feats, params={k: np.asarray(v, dtype=np.float32) for k, v in params.items()})
        ref = (feats['a'] - 10.0) / (2.0 + np.finfo(np.float32).eps)
        assert np.abs(out['a'] - ref).max() < 1e-5


def _mvn_case(g, name):
    from abnet3_amd.utils import read_vad_file
    items = ['file1', 'file2']
    feats = {k: g['%s.x%d' % (name, i)] for i, k in enumerate(items)}
    times = {k: g['%s.t%d' % (name, i)] for i, k in enumerate(items)}
    vad = None
    if str(g[name + '.vad']):
        import tempfile
        with tempfile.NamedTemporaryFile('w', suffix='.vad', delete=False) as fh:
            fh.write(str(g[name + '.vad']))
        vad = read_vad_file(fh.name)
        os.unlink(fh.name)
    return items, feats, times, vad, bool(int(g[name + '.per_file']))


@pytest.mark.parametrize('per_channel', [1, 0])
@pytest.mark.parametrize('name', ['global', 'per_file', 'global_vad', 'per_file_vad'])
def test_normalisation_cases_of_the_reference_test(golden, name, per_channel):
    """G10 (tests/golden/mvn.npz): the four cases of the reference's test/test_features.py:37-281 -- its literal
    inputs and VAD file, the frames its VAD keeps, the statistics and outputs it asserts -- through abn_mvn_stats /
    abn_mvn_apply (FeaturesGenerator.normalize_features).  The reference's bars: statistics pytest.approx (1e-6
    relative), outputs pytest.approx, and zero mean / unit deviation of the normalised data to 1e-6."""
    from abnet3_amd.features import FeaturesGenerator
    g = golden('mvn.npz')
    items, feats, times, vad, per_file = _mvn_case(g, name)
    tag = '%s.pc%d' % (name, per_channel)
    fg = FeaturesGenerator(normalization=True, norm_per_file=per_file, norm_per_channel=bool(per_channel))
    out, stats = fg.normalize_features(feats, times, vad)
    if per_file:
        assert [s[0] for s in stats] == items                       # meansvars[i][0] == 'file<i+1>'
        for i, (f, mean, std) in enumerate(stats):
            assert np.allclose(np.atleast_1d(mean), g['%s.mean%d' % (tag, i)], rtol=1e-6, atol=1e-7), (f, mean)
            assert np.allclose(np.atleast_1d(std), g['%s.std%d' % (tag, i)], rtol=1e-6), (f, std)
    else:
        assert np.allclose(np.atleast_1d(stats[0]), g[tag + '.mean'], rtol=1e-6, atol=1e-7)
        assert np.allclose(np.atleast_1d(stats[1]), g[tag + '.std'], rtol=1e-6)
    for i, f in enumerate(items):
        assert out[f].dtype == np.float32
        assert np.allclose(out[f], g['%s.out%d' % (tag, i)], rtol=1e-6, atol=1e-6), f
    if vad is None:        # "check that the new file has 0 mean and 1 variance" (per file, or over the whole set)
        sets = [out[f] for f in items] if per_file else [np.vstack([out[f] for f in items])]
        for data in sets:
            axis = 0 if per_channel else None
            assert np.allclose(np.mean(data.astype(np.float64), axis=axis), 0.0, atol=1e-6)
            assert np.allclose(np.std(data.astype(np.float64), axis=axis), 1.0, atol=1e-6)


def test_end_to_end_pipeline_trains(tmp_path, monkeypatch):
    """BASELINE.json configs[4] in miniature: fbank -> normalise -> stack -> DTW
    pair mining -> Siamese training -> embedding; the loss must go down."""
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('e2e', os.path.join(root, 'examples', 'end_to_end.py'))
    e2e = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(e2e)
    monkeypatch.setattr(sys, 'argv', ['end_to_end.py', '--utts', '12', '--words', '6', '--pairs', '120',
                                      '--epochs', '8', '--hidden', '128', '--batch', '256',
                                      '--out', str(tmp_path / 'e2e')])
    trainer, emb = e2e.main()
    assert len(trainer.train_losses) == 9 and np.isfinite(trainer.train_losses).all()
    assert trainer.train_losses[-1] < trainer.train_losses[0]
    assert emb[0].shape[1] == 100 and np.isfinite(emb[0]).all()


def test_dtw_full_size_properties():
    """BASELINE.json configs[3] at full size (10 000 pairs, ~300 frames, 9e8 cells)
    through properties that need no per-cell oracle: every path starts at (0, 0),
    ends at (N-1, M-1), advances by (1,0) / (0,1) / (1,1), has a length in
    [max(N, M), N+M-1]; the accumulated cost equals the sum of the distances along
    the returned path; aligning (y, x) costs the same as (x, y); and a sample of
    pairs is bit-identical to the C oracle."""
    import bench
    from abnet3_amd.utils import dtw_align_batch, cosine_distance
    from oracle import dtw_oracle as O
    P = 10000
    f1, o1, n1, f2, o2, n2 = bench.synth_dtw_pairs(P, seed=1234)
    d1, d2 = torch.from_numpy(f1).cuda(), torch.from_numpy(f2).cuda()
    res = dtw_align_batch(d1, o1, n1, d2, o2, n2)
    L = res.path_len.long()
    assert int((L == 0).sum()) == 0
    n1t, n2t = torch.from_numpy(n1).cuda().long(), torch.from_numpy(n2).cuda().long()
    p1, p2 = res.path1.long(), res.path2.long()
    stride = p1.shape[1]
    ar = torch.arange(P, device='cuda')
    # paths are right-aligned in their rows: entries [stride - L, stride)
    assert bool((p1[ar, stride - L] == 0).all() and (p2[ar, stride - L] == 0).all())
    assert bool((p1[:, -1] == n1t - 1).all() and (p2[:, -1] == n2t - 1).all())
    assert bool((L >= torch.maximum(n1t, n2t)).all() and (L <= n1t + n2t - 1).all())
    inside = res.mask()[:, :-1]                 # steps between two path entries
    s1, s2 = (p1[:, 1:] - p1[:, :-1])[inside], (p2[:, 1:] - p2[:, :-1])[inside]
    assert bool(((s1 == 0) | (s1 == 1)).all() and ((s2 == 0) | (s2 == 1)).all() and ((s1 + s2) >= 1).all())
    cost = res.total_cost.cpu().numpy()
    got = None
    rng = np.random.default_rng(0)
    for p in rng.choice(P, 12, replace=False):
        a, b = f1[o1[p]:o1[p] + n1[p]], f2[o2[p]:o2[p] + n2[p]]
        d = cosine_distance(a, b)                       # HIP distance matrix, float64 [N, M]
        i, j = p1[p, stride - L[p]:].cpu().numpy(), p2[p, stride - L[p]:].cpu().numpy()
        assert cost[p] == d[i, j].sum()                 # exact: float64 sums of float32 distances
        od = O.cosine_distance(a, b)
        q1, q2 = O.dtw_path(od)
        assert np.array_equal(i, q1) and np.array_equal(j, q2) and cost[p] == O.dtw_cost(od)
    # symmetry: swap the roles of the two tokens for the first 2000 pairs
    sw = dtw_align_batch(d2, o2[:2000], n2[:2000], d1, o1[:2000], n1[:2000])
    assert np.allclose(sw.total_cost.cpu().numpy(), cost[:2000], rtol=1e-6, atol=0)


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_dtw_randomised_shapes_bit_exact(seed):
    """Random feature widths (also odd, also < 4), token lengths 1..200, repeated rows
    (exact ties in the DP), zero rows (distance 1 / 0 rules) and tiny / huge scales:
    paths and accumulated costs bit-identical to the C oracle for every pair."""
    from abnet3_amd.utils import dtw_align_batch
    from oracle import dtw_oracle as O
    rng = np.random.default_rng(100 + seed)
    D = int(rng.choice([1, 3, 4, 13, 40, 47]))
    P = 60
    n1 = rng.integers(1, 200, P).astype(np.int32)
    n2 = rng.integers(1, 200, P).astype(np.int32)
    o1 = np.concatenate(([0], np.cumsum(n1)[:-1])).astype(np.int64)
    o2 = np.concatenate(([0], np.cumsum(n2)[:-1])).astype(np.int64)
    f1 = rng.standard_normal((int(n1.sum()), D)).astype(np.float32)
    f2 = rng.standard_normal((int(n2.sum()), D)).astype(np.float32)
    for p in range(P):
        a, b = f1[o1[p]:o1[p] + n1[p]], f2[o2[p]:o2[p] + n2[p]]
        kind = p % 6
        if kind == 1 and n1[p] > 3:            # runs of identical frames: ties
            a[1:n1[p] // 2] = a[0]
        elif kind == 2 and n2[p] > 2:          # a zero row on one side
            b[n2[p] // 2] = 0.0
        elif kind == 3:                        # zero rows on both sides
            a[0] = 0.0
            b[-1] = 0.0
        elif kind == 4:                        # scale must not matter to the path's validity
            a *= np.float32(1e-6)
            b *= np.float32(3e5)
        elif kind == 5 and n2[p] > 1:          # y a copy of x's frames: zero-distance cells
            b[:] = a[rng.integers(0, n1[p], n2[p])]
    res = dtw_align_batch(torch.from_numpy(f1).cuda(), o1, n1, torch.from_numpy(f2).cuda(), o2, n2)
    got = res.to_lists()
    cost = res.total_cost.cpu().numpy()
    dropped = 0
    for p in range(P):
        try:
            d = O.cosine_distance(f1[o1[p]:o1[p] + n1[p]], f2[o2[p]:o2[p] + n2[p]])
        except AssertionError:                 # cos rounded above 1 -> NaN: utils.py:59 asserts and
            assert got[p] is None, p           # the loader drops the pair; the oracle says the same
            dropped += 1
            continue
        q1, q2 = O.dtw_path(d)
        assert got[p] is not None, p
        assert np.array_equal(got[p][0], q1) and np.array_equal(got[p][1], q2), (p, D)
        assert cost[p] == O.dtw_cost(d), p
    assert dropped < P // 2


def adversarial_pairs(rng, P, D=40, longest=260):
    """P pairs of 40-d tokens whose lengths sit on and around the gang kernel's tile edges -- 1, 2, 31 .. 33, 63 .. 65, ... --
    half of the time, anywhere in 1 .. longest otherwise; every sixth pair with tied rows, zero rows, copied frames or
    extreme scales (what test_dtw_randomised_shapes_bit_exact does to single pairs)."""
    edges = np.array([1, 2, 3, 31, 32, 33, 63, 64, 65, 95, 96, 97, 127, 128, 129], dtype=np.int32)

    def lengths():
        return np.where(rng.random(P) < 0.5, rng.choice(edges, P), rng.integers(1, longest, P)).astype(np.int32)
    n1, n2 = lengths(), lengths()
    o1 = np.concatenate(([0], np.cumsum(n1)[:-1])).astype(np.int64)
    o2 = np.concatenate(([0], np.cumsum(n2)[:-1])).astype(np.int64)
    f1 = rng.standard_normal((int(n1.sum()), D)).astype(np.float32)
    f2 = rng.standard_normal((int(n2.sum()), D)).astype(np.float32)
    for p in range(0, P, 3):
        a, b = f1[o1[p]:o1[p] + n1[p]], f2[o2[p]:o2[p] + n2[p]]
        kind = (p // 3) % 6
        if kind == 1 and n1[p] > 3:
            a[1:n1[p] // 2] = a[0]
        elif kind == 2 and n2[p] > 2:
            b[n2[p] // 2] = 0.0
        elif kind == 3:
            a[0] = 0.0
            b[-1] = 0.0
        elif kind == 4:
            a *= np.float32(1e-6)
            b *= np.float32(3e5)
        elif kind == 5 and n2[p] > 1:
            b[:] = a[rng.integers(0, n1[p], n2[p])]
    return f1, o1, n1, f2, o2, n2


@pytest.mark.parametrize('seed,P,wgs', [(0, 6000, None), (1, 2000, None), (2, 300, None), (3, 3000, '1')])
def test_dtw_many_pairs_per_gang_bit_exact(seed, P, wgs, monkeypatch):
    """The dealt schedule under load (round 6): with thousands of pairs every gang works through a STREAM of pairs -- bands of
    consecutive pairs in its two slots at once, coupled and uncoupled bands, sets of boundary rows changing hands, pairs of a
    single band, of one column -- and every path, every length and the dropped pairs must equal the C oracle's, pair by pair."""
    from abnet3_amd.utils import dtw_align_batch
    from oracle import dtw_oracle as O
    if wgs:
        monkeypatch.setenv('ABN_DTW_WGS', wgs)               # one gang per CU: a dozen pairs per stream
    rng = np.random.default_rng(7000 + seed)
    f1, o1, n1, f2, o2, n2 = adversarial_pairs(rng, P)
    res = dtw_align_batch(torch.from_numpy(f1).cuda(), o1, n1, torch.from_numpy(f2).cuda(), o2, n2)
    stride = res.path1.shape[1]
    q1, q2, ln, _ = O.dtw_batch(f1, o1, n1, f2, o2, n2, stride, threads=8)
    got_len = res.path_len.cpu().numpy()
    assert np.array_equal(got_len, ln), np.nonzero(got_len != ln)[0][:10]
    g1, g2 = res.path1.cpu().numpy(), res.path2.cpu().numpy()
    cols = np.arange(stride)[None, :]
    mine = cols >= (stride - ln)[:, None]                   # right-aligned here, left-aligned in the oracle's rows
    theirs = cols < ln[:, None]
    assert np.array_equal(g1[mine], q1[theirs]) and np.array_equal(g2[mine], q2[theirs])
    assert 0 < int((ln == 0).sum()) < P // 2                # (the copied-frames pairs: cos rounds above 1 somewhere)
    cost = res.total_cost.cpu().numpy()
    for p in rng.choice(np.nonzero(ln > 0)[0], 150, replace=False):
        d = O.cosine_distance(f1[o1[p]:o1[p] + n1[p]], f2[o2[p]:o2[p] + n2[p]])
        assert cost[p] == O.dtw_cost(d), p
    if seed == 2:                                           # round 4's schedule (a pair per slot) is one switch away: the same bits
        monkeypatch.setenv('ABN_DTW_SCHED', '0')
        old = dtw_align_batch(torch.from_numpy(f1).cuda(), o1, n1, torch.from_numpy(f2).cuda(), o2, n2)
        assert torch.equal(old.path_len, res.path_len) and torch.equal(old.total_cost, res.total_cost)
        assert np.array_equal(old.path1.cpu().numpy()[mine], g1[mine]) and np.array_equal(old.path2.cpu().numpy()[mine], g2[mine])


def test_dtw_traceback_beside_the_fill_equals_one_stream(monkeypatch):
    """abn_dtw_batched_overlap (the traceback on a second stream, polling the pairs' flags while the fill kernel runs)
    against the same call on one stream: paths, lengths, costs and dropped pairs bit for bit; 40-value frames (the gang
    kernel), token lengths 1..700, a NaN pair and an empty token among them, twice in a row on the same workspace and once
    from a stream that is not the default one."""
    from abnet3_amd import utils
    rng = np.random.default_rng(77)
    P = 1500
    n1 = rng.integers(1, 700, P).astype(np.int32)
    n2 = rng.integers(1, 700, P).astype(np.int32)
    n1[5], n2[9] = 0, 0
    o1 = np.concatenate(([0], np.cumsum(n1)[:-1])).astype(np.int64)
    o2 = np.concatenate(([0], np.cumsum(n2)[:-1])).astype(np.int64)
    f1 = rng.standard_normal((int(n1.sum()), 40)).astype(np.float32)
    f2 = rng.standard_normal((int(n2.sum()), 40)).astype(np.float32)
    f2[o2[17] + 3, 7] = np.nan                                  # pair 17 is dropped (utils.py:59)
    d1, d2 = torch.from_numpy(f1).cuda(), torch.from_numpy(f2).cuda()

    def run():
        r = utils.dtw_align_batch(d1, o1, n1, d2, o2, n2)
        torch.cuda.synchronize()
        return r.path1.cpu().numpy(), r.path2.cpu().numpy(), r.path_len.cpu().numpy(), r.total_cost.cpu().numpy()

    monkeypatch.setenv('ABN_DTW_OVERLAP', '0')
    a1, a2, al, ac = run()
    assert al[5] == 0 and al[9] == 0 and al[17] == 0 and (al > 0).sum() == P - 3
    monkeypatch.setenv('ABN_DTW_OVERLAP', '1')
    st = a1.shape[1]
    inside = np.arange(st)[None, :] >= (st - al)[:, None]
    for _ in range(2):
        b1, b2, bl, bc = run()
        assert np.array_equal(al, bl) and np.array_equal(ac, bc)
        assert np.array_equal(a1[inside], b1[inside]) and np.array_equal(a2[inside], b2[inside])
    with torch.cuda.stream(torch.cuda.Stream()):
        b1, b2, bl, bc = run()
    assert np.array_equal(al, bl) and np.array_equal(ac, bc)
    assert np.array_equal(a1[inside], b1[inside]) and np.array_equal(a2[inside], b2[inside])
