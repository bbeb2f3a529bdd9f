"""The resident BatchNorm tower (csrc/tower_bn_persist.h, ABN_PATH_BN_TOWER): a BatchNorm tower in training as ONE launch
per direction, the layers separated by grid barriers -- against the numpy oracle (Linear -> Dropout -> BatchNorm ->
activation, abnet3/model.py:136-141,194-195: per-call statistics, two momentum updates per Siamese forward), and against
the one-launch-per-layer path it replaces (ABN_BN_PERSIST=0) on the same inputs: whole and ragged workgroups per
forward_once call, one and two calls, every (blocks per wave, K-split) layer shape, dropout from mask tensors and
from the per-forward seed, padded batches (n_valid), the pair loss riding along, the input's gradient.
Needs an MI355X: run with -m gpu."""
import numpy as np
import pytest
import torch

from conftest import rel_err, check_grads, is_pre_bn_bias

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def planes_forced(monkeypatch):
    monkeypatch.setenv('ABN_FUSED_MIN_ROWS', '0')
    monkeypatch.setenv('ABN_PLANES', '1')
    monkeypatch.setenv('ABN_WIDE', '0')


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def build(kw, seed, precision):
    from abnet3_amd.model import SiameseNetwork
    torch.manual_seed(seed)
    net = SiameseNetwork(**kw).cuda()
    net.precision = precision
    return net


def shake(net, rng):
    """affine parameters away from 1 / 0, a bias that shifts the column means"""
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                n = m.num_features
                m.weight.copy_(dev(rng.uniform(0.5, 1.5, n).astype(np.float32)))
                m.bias.copy_(dev(rng.standard_normal(n).astype(np.float32) * 0.2))
            if isinstance(m, torch.nn.Linear):
                m.bias.copy_(dev(rng.standard_normal(m.out_features).astype(np.float32) * 3.0))


TOWER_SHAPES = [  # (input, hidden layers, hidden, output, B): >= 8 workgroups of 32 rows over the two calls
    (40, 2, 500, 100, 512),     # C2's widths: two blocks per wave, the K-split output layer
    (40, 2, 500, 100, 300),     # ragged: the last workgroup of each call holds 12 rows
    (32, 1, 64, 32, 129),       # widths that are multiples of 32 (the column of ones opens a block of its own); one row in the last workgroup
    (64, 1, 512, 128, 160),     # the widest layer
    (128, 3, 288, 36, 200),     # 9 blocks (an odd count on two-block waves), a 36-wide output
    (4, 0, 8, 4, 128),          # the smallest tower
]


@pytest.mark.parametrize('p_drop', [0.0, 0.25])
@pytest.mark.parametrize('act', ['sigmoid', 'tanh', 'relu'])
@pytest.mark.parametrize('shape', TOWER_SHAPES)
def test_resident_tower_against_the_oracle(shape, act, p_drop, split, monkeypatch):
    """Embeddings, loss, running statistics, num_batches_tracked and every gradient against the numpy oracle, with shared
    dropout masks (mask tensors)."""
    import abnet3_amd.loss as L
    from abnet3_amd import _lib
    from oracle import siamese_np as O
    d_in, nh, hid, d_out, B = shape
    kw = dict(input_dim=d_in, num_hidden_layers=nh, hidden_dim=hid, output_dim=d_out, activation_layer=act,
              p_dropout=p_drop, batch_norm=True)
    net = build(kw, seed=B + 3, precision=split)
    spec = O.TowerSpec(d_in, nh, hid, d_out, act, True)
    # (relu: another draw -- with B + nh one pre-activation of the 128 -> 288 x 3 tower sits within rounding of 0, the oracle and the
    # kernels -- this path and the layer launches alike -- put it on different sides, and every gradient below it moves by 1 %)
    rng = np.random.default_rng(B + nh + (11 if act == 'relu' else 0))
    shake(net, rng)
    p = {k: v.detach().cpu().numpy().copy() for k, v in net.state_dict().items()}
    x1 = rng.standard_normal((B, d_in)).astype(np.float32)
    x2 = (rng.standard_normal((B, d_in)) * 2.0 + 1.0).astype(np.float32)
    y = rng.choice([1, -1], B)
    widths = [hid] * (nh + 1) + [d_out]
    masks = None
    if p_drop:
        masks = [((rng.random((2 * B, w)) >= p_drop) / (1 - p_drop)).astype(np.float32) for w in widths]
        net._mask_override = [dev(m) for m in masks]
    net.train()
    e1, e2 = net(dev(x1), dev(x2))
    assert _lib.last_forward_path() == _lib.PATH_BN_TOWER
    lv = L.coscos2(avg=False)(e1, e2, dev(y))
    lv.backward()
    assert _lib.last_backward_path() == _lib.PATH_BN_TOWER
    # The yardstick is the oracle's statements evaluated in FLOAT64 (its working dtype is one module attribute): north_star's
    # 1e-5 on embeddings and loss, 2e-5 on the gradients.  The oracle's own float32 evaluation is judged beside it -- where
    # float32 itself sits further from float64 than the bar (BatchNorm divides by a batch's standard deviation: a tower
    # whose columns nearly cancel amplifies every rounding) the kernels get what float32 needs there, and no more.
    def evaluate(dtype):
        monkeypatch.setattr(O, 'F32', dtype)
        pp = {k: v.astype(dtype) if v.dtype.kind == 'f' else v.copy() for k, v in p0.items()}
        mk = [m.astype(dtype) for m in masks] if masks else None
        o1, c1 = O.tower_forward(pp, x1.astype(dtype), spec, True, masks=[m[:B] for m in mk] if mk else None)
        o2, c2 = O.tower_forward(pp, x2.astype(dtype), spec, True, masks=[m[B:] for m in mk] if mk else None)
        ol, d1, d2, _ = O.pair_loss(o1, o2, y, 'coscos2', 0.5, False)
        og = {}
        O.tower_backward(pp, c1, d1.astype(dtype), spec, og)
        O.tower_backward(pp, c2, d2.astype(dtype), spec, og)
        return o1, o2, ol, og, pp
    p0 = p
    t1, t2, tl, tg, tp = evaluate(np.float64)                 # the truth
    o1, o2, ol, og, op = evaluate(np.float32)                 # the oracle as the other tests use it
    g1, g2 = e1.detach().cpu().numpy(), e2.detach().cpu().numpy()
    assert np.isfinite(g1).all()
    f32_emb = max(rel_err(o1, t1), rel_err(o2, t2))
    assert max(rel_err(g1, t1), rel_err(g2, t2)) < max(1e-5, 2 * f32_emb), (rel_err(g1, t1), rel_err(g2, t2), f32_emb)
    f32_loss = abs(ol - tl) / abs(tl)
    assert abs(float(lv.detach()) - tl) <= max(1e-5, 2 * f32_loss) * abs(tl) + 1e-9, (float(lv.detach()), tl, f32_loss)
    sd = net.state_dict()
    for k in p0:                    # the oracle updated its running statistics in place, once per call
        if 'running' in k:
            assert rel_err(sd[k].cpu().numpy(), tp[k]) < 1e-5, k
        if 'num_batches' in k:
            assert int(sd[k]) == int(tp[k]) == 2, k
    grads = {k: q.grad.cpu().numpy() for k, q in net.named_parameters()}
    keys = [k for k in spec.param_keys() if not is_pre_bn_bias(k, not p_drop)]
    gmax = max(np.abs(tg[k]).max() for k in spec.param_keys())
    f32_grad = max(rel_err(og[k], tg[k], floor=1e-2 * gmax) for k in keys)
    check_grads(grads, tg, spec.param_keys(), not p_drop, tol=max(2e-5, 2 * f32_grad))


def _step(net, x1, x2, y, how, lname='coscos2', avg=False, need_dx=False, n_valid=None):
    """one forward + backward; returns (embeddings, loss, gradients, buffers, input gradients)"""
    import abnet3_amd.loss as L
    for q in net.parameters():
        q.grad = None
    if how == 'autograd':
        a, b = x1.clone().requires_grad_(need_dx), x2.clone().requires_grad_(need_dx)
        e1, e2 = net(a, b)
        lv = getattr(L, lname)(avg=avg)(e1, e2, y)
        lv.backward()
        emb = torch.cat([e1, e2]).detach()
        dx = (a.grad.clone(), b.grad.clone()) if need_dx else None
    else:                            # the trainer's direct path: the pair loss inside the backward
        emb, state = net.direct_forward(x1, x2, n_valid=n_valid)
        lv = net.direct_backward_loss(state, y, lname, 0.5, avg, n_valid=n_valid)
        assert lv is not None
        emb, dx = emb.clone(), None
    return (emb, float(lv.detach()), {k: q.grad.clone() for k, q in net.named_parameters()},
            {k: v.clone() for k, v in net.state_dict().items() if 'running' in k or 'tracked' in k}, dx)


def _compare(res_tower, res_layers, tol=5e-6, dropout=False):
    (e0, l0, g0, s0, dx0), (e1, l1, g1, s1, dx1) = res_tower, res_layers
    assert torch.isfinite(e0).all()
    assert float((e0 - e1).abs().max()) <= tol * float(e1.abs().max())
    assert abs(l0 - l1) <= 1e-6 * abs(l1) + 1e-12
    for k in s0:
        if 'tracked' in k:
            assert int(s0[k]) == int(s1[k]), k
        else:
            assert float((s0[k] - s1[k]).abs().max()) <= tol * max(float(s1[k].abs().max()), 1e-6), k
    gmax = max(float(g.abs().max()) for g in g1.values())
    for k in g1:
        if is_pre_bn_bias(k, True) and not dropout:  # a Linear bias in front of BatchNorm (no dropout in between): a mathematically zero gradient, rounding noise on both sides
            assert float(g0[k].abs().max()) <= 1e-4 * gmax and float(g1[k].abs().max()) <= 1e-4 * gmax, k
            continue
        # (sums over 8192 rows of terms of both signs: the two paths reach them through differently rounded dy)
        assert float((g0[k] - g1[k]).abs().max()) <= 3e-5 * max(float(g1[k].abs().max()), 1e-2 * gmax), k
    if dx1 is not None:
        for a, b in zip(dx0, dx1):
            assert float((a - b).abs().max()) <= 5e-6 * float(b.abs().max())


@pytest.mark.parametrize('how', ['autograd', 'direct'])
@pytest.mark.parametrize('shape,act', [((40, 2, 500, 100, 4096), 'sigmoid'), ((40, 2, 500, 100, 330), 'tanh'), ((280, 2, 500, 100, 1000), 'relu'),
                                       ((40, 1, 96, 48, 250), 'sigmoid')])
def test_resident_tower_equals_the_layer_launches(shape, act, how, monkeypatch, split):
    """The same step on the resident tower and on one launch per layer (ABN_BN_PERSIST=0): embeddings, loss, running
    statistics, counters, every gradient -- through autograd (d loss / d e handed in, d loss / d input asked for) and
    through the trainer's direct calls (the pair loss inside the backward launch)."""
    from abnet3_amd import _lib
    d_in, nh, hid, d_out, B = shape
    kw = dict(input_dim=d_in, num_hidden_layers=nh, hidden_dim=hid, output_dim=d_out, activation_layer=act, p_dropout=0.0, batch_norm=True)
    rng = np.random.default_rng(B)
    x1, x2 = dev(rng.standard_normal((B, d_in)).astype(np.float32)), dev(rng.standard_normal((B, d_in)).astype(np.float32) * 1.5)
    y = dev(rng.choice([1, -1, 0] if B == 1000 else [1, -1], B))
    res = []
    for persist in ('1', '0'):
        monkeypatch.setenv('ABN_BN_PERSIST', persist)
        _lib.reload_switches()
        net = build(kw, seed=7, precision=split)
        shake(net, np.random.default_rng(5))
        net.train()
        res.append(_step(net, x1, x2, y, how, need_dx=(how == 'autograd')))
        assert _lib.last_forward_path() == (_lib.PATH_BN_TOWER if persist == '1' else _lib.PATH_BN_LAYERS)
        assert _lib.last_backward_path() == (_lib.PATH_BN_TOWER if persist == '1' else _lib.PATH_BN_LAYERS)
    monkeypatch.delenv('ABN_BN_PERSIST')
    _lib.reload_switches()
    _compare(*res)


def test_resident_tower_on_padded_batches_and_with_the_seeded_dropout(monkeypatch, split):
    """A padded batch (n_valid) and the dropout drawn from the per-forward seed, against the layer launches given the same
    seed word: the multipliers are a hash of (seed, layer, row, feature), the same in both."""
    from abnet3_amd import _lib
    kw = dict(input_dim=40, num_hidden_layers=2, hidden_dim=200, output_dim=64, activation_layer='sigmoid', p_dropout=0.2, batch_norm=True)
    rng = np.random.default_rng(3)
    B, npad = 300, 320
    x12 = torch.zeros(2 * npad, 40, device='cuda')
    x12[:B], x12[npad:npad + B] = dev(rng.standard_normal((B, 40)).astype(np.float32)), dev(rng.standard_normal((B, 40)).astype(np.float32))
    yy = torch.zeros(npad, dtype=torch.int64, device='cuda')
    yy[:B] = dev(rng.choice([1, -1], B))
    nv = torch.tensor([B], dtype=torch.int32, device='cuda')
    res = []
    for persist in ('1', '0'):
        monkeypatch.setenv('ABN_BN_PERSIST', persist)
        _lib.reload_switches()
        net = build(kw, seed=11, precision=split)
        net.train()
        torch.manual_seed(1234)          # (the per-forward seed word is drawn from torch's generator)
        res.append(_step(net, x12[:npad], x12[npad:], yy, 'direct', n_valid=nv))
        assert _lib.last_forward_path() == (_lib.PATH_BN_TOWER if persist == '1' else _lib.PATH_BN_LAYERS)
        e = res[-1][0]
        assert float(e[B:npad].abs().sum()) == 0.0 and float(e[npad + B:].abs().sum()) == 0.0
    monkeypatch.delenv('ABN_BN_PERSIST')
    _lib.reload_switches()
    _compare(*res, dropout=True)


def test_resident_tower_on_a_single_call(monkeypatch, split):
    """forward_once alone (one call of 700 rows: 22 workgroups, the last one short): the statistics, the counters (+ 1) and the
    gradients against the layer launches."""
    from abnet3_amd import _lib
    kw = dict(input_dim=40, num_hidden_layers=1, hidden_dim=160, output_dim=40, activation_layer='tanh', p_dropout=0.0, batch_norm=True)
    rng = np.random.default_rng(8)
    x = dev(rng.standard_normal((700, 40)).astype(np.float32))
    outs = []
    for persist in ('1', '0'):
        monkeypatch.setenv('ABN_BN_PERSIST', persist)
        _lib.reload_switches()
        net = build(kw, seed=2, precision=split)
        shake(net, np.random.default_rng(6))
        net.train()
        e = net.forward_once(x)
        assert _lib.last_forward_path() == (_lib.PATH_BN_TOWER if persist == '1' else _lib.PATH_BN_LAYERS)
        (e * e).sum().backward()
        outs.append([e.detach().clone()] + [q.grad.clone() for q in net.parameters()]
                    + [v.clone() for k, v in net.state_dict().items() if 'running' in k or 'tracked' in k])
    monkeypatch.delenv('ABN_BN_PERSIST')
    _lib.reload_switches()
    gmax = max(float(v.abs().max()) for v in outs[1][1:])
    for u, v in zip(*outs):
        if v.dtype == torch.int64:
            assert int(u) == int(v) == 1
        elif float(v.abs().max()) > 1e-5 * gmax:
            assert float((u - v).abs().max()) <= 5e-6 * float(v.abs().max())


def test_resident_tower_steps_like_the_layer_launches(monkeypatch, split):
    """Five Adadelta steps of the C2 tower with BatchNorm through TrainerSiamese.train_step on both paths: the losses and the
    parameters stay together (the persistent weight image, the counters and the running statistics move as before)."""
    from abnet3_amd import _lib
    from abnet3_amd.loss import coscos2
    from abnet3_amd.trainer import TrainerSiamese
    kw = dict(input_dim=40, num_hidden_layers=2, hidden_dim=500, output_dim=100, activation_layer='sigmoid', p_dropout=0.0, batch_norm=True)
    rng = np.random.default_rng(21)
    batches = [(dev(rng.standard_normal((1024, 40)).astype(np.float32)), dev(rng.standard_normal((1024, 40)).astype(np.float32)),
                dev(rng.choice([1, -1], 1024))) for _ in range(5)]
    runs = []
    for persist in ('1', '0'):
        monkeypatch.setenv('ABN_BN_PERSIST', persist)
        _lib.reload_switches()
        net = build(dict(kw, output_path='/tmp/abnet3_bn_tower_%s' % persist), seed=4, precision=split)
        tr = TrainerSiamese(network=net, loss=coscos2(avg=False), optimizer_type='adadelta', lr=0.1, dataloader=None,
                            log_dir='/tmp/abnet3_bn_tower_runs')
        net.train()
        losses = [float(tr.train_step(b, True)) for b in batches]
        runs.append((losses, {k: v.clone() for k, v in net.state_dict().items()}))
    monkeypatch.delenv('ABN_BN_PERSIST')
    _lib.reload_switches()
    (l0, s0), (l1, s1) = runs
    assert all(np.isfinite(l0))
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 2e-5 * abs(b)
    for k in s0:
        if 'tracked' in k:
            assert int(s0[k]) == int(s1[k]) == 10, k
        elif is_pre_bn_bias(k, True):               # (steps on a zero gradient's rounding noise)
            assert float((s0[k] - s1[k]).abs().max()) <= 1e-5, k
        else:
            assert float((s0[k] - s1[k]).abs().max()) <= 1e-4 * max(float(s1[k].abs().max()), 1e-3), k


def test_a_tower_that_gives_up_drops_the_step_and_recovers(split):
    """The failure word of the sync buffer set by hand -- what a hand-over that is not met in time leaves behind (the grid was
    not all on the chip; a real give-up also turns the launch's outputs into NaN, which a test on an idle GPU cannot bring
    about): NO parameter, optimizer state or BatchNorm weight moves in the step that follows (the optimizer's launch drops it),
    SiameseNetwork.resident_tower_failed() reports and clears the word with a warning, and the next steps run on the layer
    launches with the numbers an untouched twin of the network gets there."""
    import warnings
    from abnet3_amd import _lib
    from abnet3_amd.loss import coscos2
    from abnet3_amd.trainer import TrainerSiamese
    kw = dict(input_dim=40, num_hidden_layers=2, hidden_dim=500, output_dim=100, activation_layer='sigmoid', p_dropout=0.0, batch_norm=True)
    rng = np.random.default_rng(5)
    B = 512
    x1, x2 = dev(rng.standard_normal((B, 40)).astype(np.float32)), dev(rng.standard_normal((B, 40)).astype(np.float32))
    y = dev(rng.choice([1, -1], B))

    def trainer(net):
        return TrainerSiamese(network=net, loss=coscos2(avg=False), optimizer_type='adadelta', lr=0.1, dataloader=None,
                              log_dir='/tmp/abnet3_bn_recover')
    net = build(kw, seed=9, precision=split)
    tr = trainer(net)
    net.train()
    l0 = float(tr.train_step((x1, x2, y), True))
    assert np.isfinite(l0) and _lib.last_forward_path() == _lib.PATH_BN_TOWER and net.resident_tower_failed() == 0
    before = {k: v.clone() for k, v in net.state_dict().items() if 'running' not in k and 'tracked' not in k}
    for seg in net._segment_list():
        seg.sync_fail_word().fill_(1)
    tr.train_step((x1, x2, y), True)
    for k, v in before.items():
        assert torch.equal(net.state_dict()[k], v), k
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        assert net.resident_tower_failed() == 1
    assert any('resident BatchNorm tower' in str(m.message) for m in w)
    assert net.resident_tower_failed() == 0                       # cleared, and off the resident path
    # a twin that took steps 0 and 2 on the layer launches from the start (the dropped step changed nothing but the
    # running statistics of the layers it got through: compare what the optimizer owns)
    l2 = float(tr.train_step((x1, x2, y), True))
    assert _lib.last_forward_path() == _lib.PATH_BN_LAYERS and np.isfinite(l2)
    twin = build(kw, seed=9, precision=split)
    tw = trainer(twin)
    twin.train()
    m0 = float(tw.train_step((x1, x2, y), True))
    m2 = float(tw.train_step((x1, x2, y), True))
    assert abs(m0 - l0) <= 1e-5 * abs(m0) and abs(m2 - l2) <= 2e-5 * abs(m2), (l0, m0, l2, m2)


def test_a_tower_that_gives_up_hands_the_all_reduce_a_zero_gradient(split):
    """The data-parallel form of the step: the backward writes the gradient itself (an all-reduce follows, the optimizer's
    launch cannot guard it).  With the failure word set the slab sum writes ZEROS -- weights, biases and the BatchNorm
    gradients the tower's backward had written itself -- so the rank contributes nothing to that step instead of poison."""
    from abnet3_amd import _lib
    import abnet3_amd.loss as L
    kw = dict(input_dim=40, num_hidden_layers=2, hidden_dim=500, output_dim=100, activation_layer='sigmoid', p_dropout=0.0, batch_norm=True)
    rng = np.random.default_rng(6)
    B = 512
    x1, x2 = dev(rng.standard_normal((B, 40)).astype(np.float32)), dev(rng.standard_normal((B, 40)).astype(np.float32))
    y = dev(rng.choice([1, -1], B))
    net = build(kw, seed=10, precision=split)
    net.train()
    e1, e2 = net(x1, x2)
    assert _lib.last_forward_path() == _lib.PATH_BN_TOWER
    L.coscos2(avg=False)(e1, e2, y).backward()
    assert _lib.last_backward_path() == _lib.PATH_BN_TOWER
    assert all(float(q.grad.abs().max()) > 0 for k, q in net.named_parameters() if not is_pre_bn_bias(k, True))
    for q in net.parameters():
        q.grad = None
    e1, e2 = net(x1, x2)
    for seg in net._segment_list():
        seg.sync_fail_word().fill_(1)
    L.coscos2(avg=False)(e1, e2, y).backward()
    for k, q in net.named_parameters():
        assert q.grad is not None and float(q.grad.abs().max()) == 0.0, k
    for seg in net._segment_list():
        seg.recover_sync()

