"""Host-side logic of the reference-shaped surfaces (no GPU needed)."""
import os

import numpy as np
import pytest
import torch


def test_siamese_network_keys_init_and_asserts():
    from abnet3_amd.model import SiameseNetwork, NetworkBuilder
    from oracle import torch_ref
    kw = dict(input_dim=40, num_hidden_layers=2, hidden_dim=24, output_dim=12,
              p_dropout=0.0, activation_layer='sigmoid')
    for bn in (False, True):
        for init in ('xavier_uni', 'xavier_normal', 'orthogonal'):
            torch.manual_seed(3)
            net = SiameseNetwork(batch_norm=bn, type_init=init, **kw)
            ref = torch_ref.build(3, batch_norm=bn, type_init=init, **kw)
            assert isinstance(net, NetworkBuilder) and isinstance(net, torch.nn.Module)
            assert list(net.state_dict().keys()) == list(ref.state_dict().keys())
            for (k, a), (_, b) in zip(net.state_dict().items(), ref.state_dict().items()):
                assert torch.equal(a, b), k          # same RNG consumption => same init
    keys = list(SiameseNetwork(batch_norm=True, **kw).state_dict().keys())
    assert 'hidden_layers.4.weight' in keys and 'hidden_layers.6.running_var' in keys
    keys = list(SiameseNetwork(batch_norm=False, **kw).state_dict().keys())
    assert 'hidden_layers.3.weight' in keys and 'output_layer.0.bias' in keys
    with pytest.raises(AssertionError):
        SiameseNetwork(activation_layer='softmax', **{k: v for k, v in kw.items() if k != 'activation_layer'})
    with pytest.raises(AssertionError):
        SiameseNetwork(**dict(kw, input_dim=40.0))
    with pytest.raises(AssertionError):
        SiameseNetwork(type_init='he', **kw)
    net = SiameseNetwork(output_path='/tmp/x', **kw)
    assert net.whoami()['class_name'] == 'SiameseNetwork' and net.output_path == '/tmp/x'


def test_flat_parameter_buffer_round_trip(tmp_path):
    from abnet3_amd.model import SiameseNetwork
    torch.manual_seed(0)
    net = SiameseNetwork(input_dim=10, num_hidden_layers=1, hidden_dim=7, output_dim=5,
                         p_dropout=0.0, activation_layer='relu', batch_norm=True,
                         output_path=str(tmp_path / 'net'))
    before = {k: v.clone() for k, v in net.state_dict().items()}
    flat = net.flat_parameters()
    assert flat.numel() % 64 == 0 and net._is_flat()
    for k, v in net.state_dict().items():
        assert torch.equal(v, before[k])
    # parameters are views: an in-place update of the flat buffer is visible
    flat.add_(1.0)
    assert torch.equal(net.input_emb[0].weight.data, before['input_emb.0.weight'] + 1.0)
    assert all(p.data_ptr() % 256 == flat.data_ptr() % 256 for p in net.parameters())
    net.save_network()
    other = SiameseNetwork(input_dim=10, num_hidden_layers=1, hidden_dim=7, output_dim=5,
                           p_dropout=0.0, activation_layer='relu', batch_norm=True)
    other.load_network(str(tmp_path / 'net.pth'))
    for (k, a), (_, b) in zip(net.state_dict().items(), other.state_dict().items()):
        assert torch.equal(a, b), k


def test_no_cpu_fallback_anywhere():
    from abnet3_amd._lib import HipLibraryError
    from abnet3_amd.model import SiameseNetwork
    from abnet3_amd.loss import coscos2, cosmargin
    net = SiameseNetwork(input_dim=10, num_hidden_layers=0, hidden_dim=8, output_dim=4,
                         p_dropout=0.0, activation_layer='tanh')
    x = torch.randn(3, 10)
    with pytest.raises(HipLibraryError):
        net(x, x)
    with pytest.raises(HipLibraryError):
        net.forward_once(x)
    for loss in (coscos2(), cosmargin()):
        with pytest.raises(HipLibraryError):
            loss(torch.randn(3, 4), torch.randn(3, 4), torch.ones(3))
    with pytest.raises(AssertionError):
        coscos2()(torch.randn(3, 4), torch.randn(2, 4), torch.ones(3))
    with pytest.raises(AssertionError):
        cosmargin(margin=1.5)
    assert coscos2().avg is True and cosmargin().margin == 0.5
    assert cosmargin(avg=False).whoami()['class_name'] == 'cosmargin'


def test_embedder_and_trainer_surface_errors():
    from abnet3_amd.embedder import EmbedderBuilder, EmbedderSiamese
    with pytest.raises(ValueError):
        EmbedderSiamese(network=None)
    with pytest.raises(NotImplementedError):
        EmbedderBuilder(network=object()).embed()
    e = EmbedderSiamese(network=object(), network_path='a', feature_path='b', output_path='c')
    assert (e.cuda, e.batch_size) == (True, 5000)
    from abnet3_amd.trainer import FlatOptimizer
    assert set(FlatOptimizer.HP) == {'sgd', 'adadelta', 'adam', 'adagrad', 'RMSprop'}


def test_pairs_file_round_trip_and_grouping(tmp_path):
    from abnet3_amd.utils import read_dataset, write_dataset, group_pairs, read_pairs, print_token
    pairs = [('s01', 0.5, 1.234, 's02', 10.0, 10.456, 'same'),
             ('s03', 3.14159, 3.5, 's01', 0.0, 0.2, 'diff')]
    f = tmp_path / 'dataset'
    write_dataset(str(f), pairs)
    lines = open(str(f)).read().splitlines()
    assert lines[0] == 's01 0.50 1.23 s02 10.00 10.46 same'       # %.2f, sampler.py:697-742
    back = read_dataset(str(f))
    assert back[1] == ('s03', 3.14, 3.5, 's01', 0.0, 0.2, 'diff')
    g = group_pairs(back)
    assert len(g['same']) == 1 and len(g['diff']) == 1 and len(g['same'][0]) == 6
    assert read_pairs(str(f)) == g
    assert print_token(('a', 1, 2.005)) in ('a 1.00 2.00', 'a 1.00 2.01')
    (tmp_path / 'bad').write_text('a 0 1 b 0 1 maybe\n')
    with pytest.raises(AssertionError):
        read_dataset(str(tmp_path / 'bad'))


def test_features_accessor_inclusive_window():
    from abnet3_amd.utils import Features_Accessor
    feats = {'u': np.arange(40, dtype=np.float64).reshape(10, 4)}
    times = {'u': np.arange(10) * 0.01 + 0.0025}
    acc = Features_Accessor(times, feats)
    assert acc.features['u'].dtype == np.float32           # cast like the reference
    got = acc.get('u', 0.02, 0.053)                         # t in {0.0225 .. 0.0525}
    assert got.shape == (4, 4) and got[0, 0] == 8 and got[-1, 0] == 20
    t = times['u']
    assert acc.get('u', t[3], t[5]).shape == (3, 4)         # both ends inclusive
    assert acc.get('u', 0.2, 0.3).shape == (0, 4)
    assert acc.get_between_frames('u', 2, 5).shape == (3, 4)


def test_mel_filterbank_equals_oracle_definition():
    from abnet3_amd.features import mel_filterbank, FeaturesGenerator
    from oracle import features_np as F
    for fs in (16000, 22050, 44100):
        assert np.abs(mel_filterbank(fs) - F.mel_filterbank(fs)).max() < 1e-15
    with pytest.raises(ValueError):
        mel_filterbank(8000)
    fg = FeaturesGenerator(deltas=True, deltasdeltas=True)     # accepted like in the reference (features.py:110-111)
    assert fg.deltas and fg.deltasdeltas


def test_reference_is_never_needed_at_run_time():
    """nothing under abnet3_amd/ imports the oracle or reads /root/reference"""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'abnet3_amd')
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(dirpath, f)).read()
                assert '/root/reference' not in text, f
                assert 'import oracle' not in text and 'from oracle' not in text, f


def test_multitask_surface_host_side(tmp_path):
    """SiameseMultitaskNetwork / weighted_loss_multi / same_speaker: everything
    that needs no GPU."""
    import ast
    from conftest import load_golden
    from abnet3_amd._lib import HipLibraryError
    from abnet3_amd.model import SiameseMultitaskNetwork, NetworkBuilder
    from abnet3_amd.loss import weighted_loss_multi, coscos2, cosmargin
    from abnet3_amd.dataloader import OriginalDataLoader, MultiTaskDataLoader
    g = load_golden('multitask_relu_bn.npz')
    kw = ast.literal_eval(str(g['kw']))
    torch.manual_seed(5)
    net = SiameseMultitaskNetwork(output_path=str(tmp_path / 'mt'), **kw)
    assert isinstance(net, NetworkBuilder)
    sd = net.state_dict()
    assert sorted(sd) == sorted(k[2:] for k in g if k.startswith('p.'))
    for k, v in sd.items():
        assert np.array_equal(v.numpy(), g['p.' + k]), k      # reference init from the same seed
    # the branches forward never calls stay out of the flat bucket
    live = net.live_parameters()
    dead = [p for k, p in net.named_parameters() if k.startswith(('hidden_layers_spk', 'hidden_layers_phn'))]
    assert dead and not any(any(p is q for q in live) for p in dead)
    assert len(live) + len(dead) == len(list(net.parameters()))
    flat = net.flat_parameters()
    assert net._is_flat() and flat.numel() == sum((p.numel() + 63) // 64 * 64 for p in live)
    for k, v in net.state_dict().items():
        assert np.array_equal(v.numpy(), g['p.' + k]), k
    net.save_network()
    other = SiameseMultitaskNetwork(**kw)
    other.load_network(str(tmp_path / 'mt.pth'))
    assert all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), other.state_dict().values()))
    with pytest.raises(HipLibraryError):
        net(torch.randn(3, 40), torch.randn(3, 40))
    with pytest.raises(AssertionError):
        SiameseMultitaskNetwork(**dict(kw, num_hidden_layers_spk=None))
    with pytest.raises(AssertionError):
        SiameseMultitaskNetwork(**dict(kw, activation_layer='softmax'))

    with pytest.raises(AssertionError):
        weighted_loss_multi(loss_spk=coscos2(), loss_phn=coscos2(), weight=1)      # int, not float
    with pytest.raises(AssertionError):
        weighted_loss_multi(loss_spk=coscos2(), loss_phn=coscos2(), weight=1.5)
    w = weighted_loss_multi(loss_spk=coscos2(avg=False), loss_phn=cosmargin())
    assert w.weight == 0.5 and w.avg is True and w.whoami()['class_name'] == 'weighted_loss_multi'

    dl = OriginalDataLoader('p', 'f')
    spk = {'u0': 'alice', 'u1': ''.join(['ali', 'ce']), 'u2': 'b', 'u3': 'b'}
    assert dl.same_speaker(spk, 'u0', 'u0') and not dl.same_speaker(spk, 'u0', 'u1')
    assert dl.same_speaker(spk, 'u2', 'u3') and not dl.same_speaker(spk, 'u1', 'u2')
    ml = MultiTaskDataLoader('p', 'f', fid2spk_file='x', speaker_match='equal')
    assert ml.same_speaker(spk, 'u0', 'u1') and not ml.same_speaker(spk, 'u0', 'u2')


def test_batch_norm_sync_callback_finds_the_buffer_a_pointer_lies_in():
    """parallel.BatchNormSync hands the library a C callback (abn_allreduce_fn); the library calls it with a raw device
    pointer into a buffer the object was shown.  Host logic only: the pointer -> tensor view lookup and its refusals
    (the collective itself runs in tests/test_gpu_dp.py under two ranks)."""
    import torch
    from abnet3_amd import parallel
    s = parallel.BatchNormSync()
    assert s.world == 1 and s.fn
    a, b = torch.zeros(64, dtype=torch.float32), torch.zeros(32, dtype=torch.float32)
    s.buffers = [a, b]
    assert s._allreduce(None, b.data_ptr() + 16, 4, None) == 0 and s.calls == 1        # 4 doubles = floats 4 .. 11 of b
    assert s._allreduce(None, a.data_ptr(), 32, None) == 0 and s.calls == 2            # the whole of a
    assert s._allreduce(None, a.data_ptr() + 8, 32, None) == 1                         # runs past the end of a
    assert s._allreduce(None, b.data_ptr() + 4, 2, None) == 1                          # not 8-byte aligned
    assert s._allreduce(None, 12345, 1, None) == 1 and s.calls == 2                    # in nobody's buffer


def test_align_cache_compacts_its_chunks_and_keeps_every_path():
    # (the iterator path adds a chunk per align_pairs call; a plan build first makes ONE of them)
    from abnet3_amd.dataloader import AlignCache
    cache = AlignCache()
    rng = np.random.RandomState(0)
    want = {}
    for c in range(5):
        n = [int(v) for v in rng.randint(0, 7, size=4)]
        g1 = torch.from_numpy(rng.randint(0, 1000, size=sum(n)).astype(np.int64))
        g2 = torch.from_numpy(rng.randint(0, 1000, size=sum(n)).astype(np.int64))
        chunk = cache.add_chunk(g1, g2)
        start = 0
        for k, m in enumerate(n):
            cache.put((c, k), chunk, start, m)
            want[(c, k)] = None if m == 0 else (g1[start:start + m].clone(), g2[start:start + m].clone())
            start += m
    early = cache[(1, [k for k in range(4) if want[(1, k)] is not None][0])]      # a view handed out before
    cache.compact()
    assert len(cache.chunks) == 1 and len(cache) == 20
    for key, w in want.items():
        got = cache[key]
        assert (got is None) == (w is None)
        if w is not None:
            assert torch.equal(got[0], w[0]) and torch.equal(got[1], w[1])
    assert early[0].numel() > 0          # (still readable: it keeps its own storage)
    cache.compact()                      # idempotent
    assert len(cache.chunks) == 1
