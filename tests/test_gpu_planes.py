"""The operand-plane kernels of csrc/tower_planes.h (precision bf16x3 / bf16, widths <= 512 and multiples of
4): the library takes them from 256 rows up by itself; here ABN_FUSED_MIN_ROWS = 0 forces them on small towers
whose shapes walk every branch -- one and two column blocks per wave, the K-split of narrow layers, widths that
are multiples of 32 (the column of ones then opens a block of its own), 512-wide layers, ragged row counts,
dropout, every activation -- against the numpy oracle, plus a decode of the operand images themselves.
BatchNorm towers: the inference forward (running statistics in the epilogue), the one-launch-per-layer training
forward and backward (batch statistics from per-workgroup sums), their dropout, bf16 and fallback cases.
Needs an MI355X: run with -m gpu."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import rel_err, check_grads, is_pre_bn_bias
from planes_decode import decode, decode_t

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def planes_forced(monkeypatch):
    monkeypatch.setenv('ABN_FUSED_MIN_ROWS', '0')
    monkeypatch.setenv('ABN_PLANES', '1')
    monkeypatch.setenv('ABN_WIDE', '0')          # (small batches would take the layer-per-launch kernels: tests/test_gpu_wide.py)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def build(kw, seed, precision):
    from abnet3_amd.model import SiameseNetwork
    from oracle import siamese_np as O
    torch.manual_seed(seed)
    net = SiameseNetwork(**kw).cuda()
    net.precision = precision
    spec = O.TowerSpec(kw['input_dim'], kw['num_hidden_layers'], kw['hidden_dim'], kw['output_dim'],
                       kw['activation_layer'], False, kw.get('last_non_linearity', 'default'))
    p = {k: v.detach().cpu().numpy().copy() for k, v in net.state_dict().items()}
    return net, spec, p


SHAPES = [  # (input, hidden layers, hidden, output, B)
    (40, 2, 500, 100, 48),      # C2's widths: two blocks per wave, K-split output layer
    (32, 1, 64, 32, 33),        # every width a multiple of 32: the ones column gets its own block; ragged rows
    (64, 1, 512, 128, 16),      # the widest layer the LDS image holds; K-split with 32 steps
    (4, 0, 8, 4, 5),            # the smallest tower the kernels accept
    (128, 3, 288, 36, 70),      # 9 blocks (odd count on two-block waves), 36-wide output
    (20, 14, 32, 12, 40),       # 16 Linear layers
]


@pytest.mark.parametrize('precision,tol,gtol', [('bf16x3', 1e-5, 1e-4), ('f16x2', 1e-5, 1e-4), ('bf16', 3e-2, None)])
@pytest.mark.parametrize('act', ['sigmoid', 'tanh', 'relu'])
@pytest.mark.parametrize('shape', SHAPES)
def test_forward_loss_and_gradients_against_the_oracle(shape, act, precision, tol, gtol):
    import abnet3_amd.loss as L
    from oracle import siamese_np as O
    d_in, nh, hid, d_out, B = shape
    if nh > 8 and (act == 'sigmoid' or precision == 'bf16'):
        pytest.skip('sixteen sigmoid layers at initialisation map every input to the same embedding (gradients are '
                    'rounding noise on both sides); sixteen bf16 layers are outside any tolerance worth writing')
    kw = dict(input_dim=d_in, num_hidden_layers=nh, hidden_dim=hid, output_dim=d_out, activation_layer=act,
              p_dropout=0.0, batch_norm=False)
    if act == 'relu':
        kw['last_non_linearity'] = None
    net, spec, p = build(kw, seed=B, precision=precision)
    rng = np.random.default_rng(B)
    x1 = rng.standard_normal((B, d_in)).astype(np.float32)
    x2 = rng.standard_normal((B, d_in)).astype(np.float32)
    y = rng.choice([1, -1], B)
    net.train()
    e1, e2 = net(dev(x1), dev(x2))
    lv = L.coscos2(avg=False)(e1, e2, dev(y))
    lv.backward()
    o1, c1 = O.tower_forward(p, x1, spec, True)
    o2, c2 = O.tower_forward(p, x2, spec, True)
    ol, d1, d2, _ = O.pair_loss(o1, o2, y, 'coscos2', 0.5, False)
    og = {}
    O.tower_backward(p, c1, d1, spec, og)
    O.tower_backward(p, c2, d2, spec, og)
    assert rel_err(e1.detach().cpu().numpy(), o1) < tol
    assert rel_err(e2.detach().cpu().numpy(), o2) < tol
    assert abs(float(lv.detach()) - ol) <= 10 * tol * abs(ol) + 1e-6
    grads = {k: q.grad.cpu().numpy() for k, q in net.named_parameters()}
    if precision != 'bf16':
        check_grads(grads, og, spec.param_keys(), False, tol=gtol)
    else:
        # 8-bit operands under a loss whose gradient is a difference of near-equal vectors: the
        # direction and size of every gradient tensor, not its entries
        # (the output layer's: further down a sigmoid tower at its initialisation the true gradient shrinks
        # below what 8-bit operands resolve; the bf16 x 3 cases pin the arithmetic, these the data paths)
        for k in spec.param_keys()[-2:]:
            a, b = grads[k].ravel().astype(np.float64), og[k].ravel().astype(np.float64)
            assert a @ b / (np.linalg.norm(a) * np.linalg.norm(b)) > 0.9, k
            assert abs(np.linalg.norm(a) / np.linalg.norm(b) - 1) < 0.3, k
    # the direct path (fused loss gradient + d_out_is_dz, deferred reduction) yields the same gradients
    from abnet3_amd.trainer import TrainerSiamese
    tr = TrainerSiamese(network=net, loss=L.coscos2(avg=False), optimizer_type='sgd', lr=0.0, dataloader=None,
                        log_dir='/tmp/abn_runs')
    tr.train_step((dev(x1), dev(x2), dev(y)), True)
    for k, q in net.named_parameters():
        assert rel_err(q.grad.cpu().numpy(), grads[k], floor=1e-30) < (1e-6 if precision != 'bf16' else 1e-5), k


def test_input_gradient_through_the_chain(split):
    """x.requires_grad: the data-gradient chain runs one product further (dX = dZ_0 W_0)."""
    import abnet3_amd.loss as L
    from oracle import siamese_np as O
    kw = dict(input_dim=40, num_hidden_layers=1, hidden_dim=72, output_dim=36, activation_layer='tanh',
              p_dropout=0.0, batch_norm=False)
    net, spec, p = build(kw, seed=6, precision=split)
    B = 45
    rng = np.random.default_rng(6)
    x1n, x2n = rng.standard_normal((B, 40)).astype(np.float32), rng.standard_normal((B, 40)).astype(np.float32)
    y = rng.choice([1, -1], B)
    x1, x2 = dev(x1n).requires_grad_(True), dev(x2n).requires_grad_(True)
    net.train()
    e1, e2 = net(x1, x2)
    L.coscos2(avg=False)(e1, e2, dev(y)).backward()
    o1, c1 = O.tower_forward(p, x1n, spec, True)
    o2, c2 = O.tower_forward(p, x2n, spec, True)
    _, d1, d2, _ = O.pair_loss(o1, o2, y, 'coscos2', 0.5, False)
    og = {}
    _, dx1 = O.tower_backward(p, c1, d1, spec, og, return_dx=True)
    _, dx2 = O.tower_backward(p, c2, d2, spec, og, return_dx=True)
    assert rel_err(x1.grad.cpu().numpy(), dx1) < 1e-5 and rel_err(x2.grad.cpu().numpy(), dx2) < 1e-5
    check_grads({k: q.grad.cpu().numpy() for k, q in net.named_parameters()}, og, spec.param_keys(), False, tol=1e-4)


def test_dropout_masks_scale_both_chains(split):
    """train mode with p_dropout: the masks multiply the pre-activations in the forward epilogue and the data
    gradient in the backward chain (abnet3/model.py:137,148,157 Dropout between Linear and the activation)."""
    import abnet3_amd.loss as L
    from oracle import siamese_np as O
    kw = dict(input_dim=40, num_hidden_layers=2, hidden_dim=100, output_dim=36, activation_layer='sigmoid',
              p_dropout=0.3, batch_norm=False)
    net, spec, p = build(kw, seed=9, precision=split)
    B = 50
    rng = np.random.default_rng(9)
    x1 = rng.standard_normal((B, 40)).astype(np.float32)
    x2 = rng.standard_normal((B, 40)).astype(np.float32)
    y = rng.choice([1, -1], B)
    widths = [100, 100, 100, 36]
    gen = torch.Generator(device='cuda').manual_seed(5)
    masks = [(torch.rand(2 * B, w, device='cuda', generator=gen) > 0.3).float() / 0.7 for w in widths]
    net._mask_override = masks
    net.train()
    e1, e2 = net(dev(x1), dev(x2))
    lv = L.cosmargin(avg=True)(e1, e2, dev(y))
    lv.backward()
    net._mask_override = None
    m = [t.cpu().numpy() for t in masks]
    o1, c1 = O.tower_forward(p, x1, spec, True, masks=[t[:B] for t in m])
    o2, c2 = O.tower_forward(p, x2, spec, True, masks=[t[B:] for t in m])
    ol, d1, d2, _ = O.pair_loss(o1, o2, y, 'cosmargin', 0.5, True)
    og = {}
    O.tower_backward(p, c1, d1, spec, og)
    O.tower_backward(p, c2, d2, spec, og)
    assert rel_err(e1.detach().cpu().numpy(), o1) < 1e-5 and rel_err(e2.detach().cpu().numpy(), o2) < 1e-5
    check_grads({k: q.grad.cpu().numpy() for k, q in net.named_parameters()}, og, spec.param_keys(), False, tol=1e-4)


@pytest.mark.parametrize('precision,NP', [('bf16x3', 3), ('f16x2', 2), ('bf16', 1)])
def test_operand_images_decode_to_their_tensors(precision, NP):
    """The packed weights (both orientations), the transposed planes of every layer input with their column of
    ones, and the transposed planes of every dZ, read back from the forward workspace / backward scratch."""
    from abnet3_amd import _lib, model as M
    kw = dict(input_dim=40, num_hidden_layers=1, hidden_dim=96, output_dim=32, activation_layer='sigmoid',
              p_dropout=0.0, batch_norm=False)
    net, spec, p = build(kw, seed=4, precision=precision)
    lib = _lib.load()
    fn = lib.abn_tower_image_offset
    fn.restype = ctypes.c_int64
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int]
    B = 37
    R = 2 * B
    dims = [40, 96, 96, 32]
    rng = np.random.default_rng(4)
    x1, x2 = dev(rng.standard_normal((B, 40)).astype(np.float32)), dev(rng.standard_normal((B, 40)).astype(np.float32))
    net.train()
    out, (seg, sv, gp) = net.direct_forward(x1, x2)
    dz_top = dev((rng.standard_normal((R, 32)) * 1e-3).astype(np.float32))
    _, _, pending = M._segment_backward(seg, sv, dz_top, gp, False, True, True)
    torch.cuda.synchronize()
    desc = pending[0]
    ws16 = sv.ws.view(torch.int16).cpu().numpy().view(np.uint16)
    w16 = seg._wpack.view(torch.int16).cpu().numpy().view(np.uint16) if desc.wpack else ws16      # the weight images' home
    sc16 = pending[2].view(torch.int16).cpu().numpy().view(np.uint16)
    Ws = [q.detach().double().cpu().numpy() for k, q in net.named_parameters() if k.endswith('weight')]
    bs = [q.detach().double().cpu().numpy() for k, q in net.named_parameters() if k.endswith('bias')]
    acts = [torch.cat([x1, x2]).double().cpu().numpy()]
    for l in range(3):
        acts.append(1 / (1 + np.exp(-(acts[-1] @ Ws[l].T + bs[l]))))
    dz = [None, None, dz_top.double().cpu().numpy()]
    for l in (2, 1):
        dz[l - 1] = (dz[l] @ Ws[l]) * acts[l] * (1 - acts[l])
    eps = 2.0 ** -22 if NP == 3 else (2.0 ** -21 if NP == 2 else 2.0 ** -8)    # three bf16 terms carry 24 bits, two fp16 22 (of the block's / row's largest), one bf16 8
    steps = lambda c: ((c + 15) // 16 + 3) // 4 * 4
    row_steps = (R + 31) // 32 * 2
    for l in range(3):
        d = decode(w16[2 * fn(ctypes.byref(desc), R, 2, 0, l):], (dims[l + 1] + 31) // 32, steps(dims[l]), NP)
        assert np.abs(d[:dims[l + 1], :dims[l]] - Ws[l]).max() <= eps * np.abs(Ws[l]).max()
        assert np.abs(d[dims[l + 1]:]).max(initial=0) == 0 and np.abs(d[:, dims[l]:]).max(initial=0) == 0
        d = decode(w16[2 * fn(ctypes.byref(desc), R, 2, 1, l):], (dims[l] + 31) // 32, steps(dims[l + 1]), NP)
        assert np.abs(d[:dims[l], :dims[l + 1]] - Ws[l].T).max() <= eps * np.abs(Ws[l]).max()
        d = decode_t(ws16[2 * fn(ctypes.byref(desc), R, 2, 2, l):], (dims[l] + 1 + 31) // 32, row_steps, NP)
        assert np.abs(d[:dims[l], :R] - acts[l].T).max() <= max(eps, 3e-7) * np.abs(acts[l]).max()
        assert np.array_equal(d[dims[l], :R], np.ones(R)) and np.abs(d[dims[l], R:]).max(initial=0) == 0
        assert np.abs(d[dims[l] + 1:]).max(initial=0) == 0
        d = decode_t(sc16[2 * fn(ctypes.byref(desc), R, 2, 3, l):], (dims[l + 1] + 31) // 32, row_steps, NP)
        tol = 3e-6 if NP >= 2 else 3e-2
        assert np.abs(d[:dims[l + 1], :R] - dz[l].T).max() <= tol * np.abs(dz[l]).max(), l
        assert np.abs(d[:, R:]).max(initial=0) == 0          # rows past the batch contribute nothing


@pytest.mark.parametrize('lname,avg', [('coscos2', False), ('coscos2', True), ('cosmargin', True), ('cosmargin', False)])
@pytest.mark.parametrize('B,ydtype,act,p_drop', [(4096, torch.int64, 'sigmoid', 0.0), (33, torch.float32, 'tanh', 0.0),
                                               (70, torch.int8, 'sigmoid', 0.25), (48, torch.float64, 'relu', 0.0)])
def test_pair_loss_inside_the_backward_equals_the_two_calls(lname, avg, B, ydtype, act, p_drop, monkeypatch, split):
    """abn_tower_backward_loss (TrainerSiamese.train_step's path): the data-gradient chain computes the pair loss
    and d loss / d z in its first phase.  Against abn_pair_loss_dz + abn_tower_backward on the same forward: loss
    and every gradient agree to rounding (the per-pair arithmetic is the same, fp64)."""
    import abnet3_amd.loss as L
    kw = dict(input_dim=40, num_hidden_layers=2, hidden_dim=500 if B > 1000 else 72, output_dim=100 if B > 1000 else 36,
              activation_layer=act, p_dropout=p_drop, batch_norm=False)
    if act == 'relu':
        kw['last_non_linearity'] = None
    net, spec, p = build(kw, seed=B, precision=split)
    loss = getattr(L, lname)(avg=avg) if lname == 'coscos2' else L.cosmargin(avg=avg, margin=0.3)
    rng = np.random.default_rng(B)
    x1, x2 = dev(rng.standard_normal((B, 40)).astype(np.float32)), dev(rng.standard_normal((B, 40)).astype(np.float32))
    y = dev(rng.choice([1, -1, 0] if B == 48 else [1, -1], B)).to(ydtype)
    net.train()
    if p_drop:
        gen = torch.Generator(device='cuda').manual_seed(1)
        widths = [kw['hidden_dim']] * 3 + [kw['output_dim']]
        net._mask_override = [(torch.rand(2 * B, w, device='cuda', generator=gen) > p_drop).float() / (1 - p_drop) for w in widths]
    res = []
    for fused in ('1', '0'):
        monkeypatch.setenv('ABN_LOSS_IN_BACKWARD', fused)
        for q in net.parameters():
            q.grad = None
        emb, state = net.direct_forward(x1, x2)
        info = net.direct_dz_info(state)
        lv = net.direct_backward_loss(state, y, lname, 0.3, avg) if fused == '1' else None
        assert (lv is not None) == (fused == '1')
        if lv is None:
            lv, dz = loss.value_and_dz(emb[:B], emb[B:], y, info[0], info[1])
            net.direct_backward(state, dz.view(2 * B, -1), d_out_is_dz=True)
        res.append((float(lv), [q.grad.clone() for q in net.parameters()]))
    net._mask_override = None
    assert abs(res[0][0] - res[1][0]) <= 1e-6 * abs(res[1][0]) + 1e-12
    gmax = max(float(g.abs().max()) for g in res[1][1])
    for (k, _), a, b in zip(net.named_parameters(), res[0][1], res[1][1]):
        assert float((a - b).abs().max()) <= 2e-6 * max(float(b.abs().max()), 1e-2 * gmax), k


@pytest.mark.parametrize('lname,avg', [('coscos2', False), ('cosmargin', True)])
@pytest.mark.parametrize('B,ydtype,act', [(4096, torch.int64, 'sigmoid'), (330, torch.float32, 'tanh'), (1000, torch.float64, 'relu')])
def test_pair_loss_inside_the_batchnorm_backward_equals_the_two_calls(lname, avg, B, ydtype, act, monkeypatch, split):
    """abn_tower_backward_loss on a BatchNorm tower: the launch that sums the output layer's dy and dy xhat per workgroup
    also computes the pair loss and d loss / d a from the embeddings (no loss launch, no d_out tensor).  Against
    abn_pair_loss + abn_tower_backward on the same forward: loss and every gradient (gamma / beta included) agree to
    rounding; 330 pairs: the last workgroup of each call holds 10 rows; labels 0 among them at 1000."""
    import abnet3_amd.loss as L
    kw = dict(input_dim=40, num_hidden_layers=2, hidden_dim=500 if B > 2000 else 72, output_dim=100 if B > 2000 else 36,
              activation_layer=act, p_dropout=0.0, batch_norm=True)
    net, spec, p = build(kw, seed=B, precision=split)
    loss = getattr(L, lname)(avg=avg) if lname == 'coscos2' else L.cosmargin(avg=avg, margin=0.3)
    rng = np.random.default_rng(B)
    x1, x2 = dev(rng.standard_normal((B, 40)).astype(np.float32)), dev(rng.standard_normal((B, 40)).astype(np.float32))
    y = dev(rng.choice([1, -1, 0] if B == 1000 else [1, -1], B)).to(ydtype)
    net.train()
    res = []
    for fused in ('1', '0'):
        monkeypatch.setenv('ABN_LOSS_IN_BACKWARD', fused)
        for q in net.parameters():
            q.grad = None
        emb, state = net.direct_forward(x1, x2)
        lv = net.direct_backward_loss(state, y, lname, 0.3, avg) if fused == '1' else None
        assert (lv is not None) == (fused == '1')
        if lv is None:
            lv, de = loss.value_and_grad(emb[:B], emb[B:], y)
            net.direct_backward(state, de.view(2 * B, -1))
        res.append((float(lv), [q.grad.clone() for q in net.parameters()]))
    assert abs(res[0][0] - res[1][0]) <= 1e-6 * abs(res[1][0]) + 1e-12
    gmax = max(float(g.abs().max()) for g in res[1][1])
    for (k, _), a, b in zip(net.named_parameters(), res[0][1], res[1][1]):
        assert float((a - b).abs().max()) <= 2e-6 * max(float(b.abs().max()), 1e-2 * gmax), k


@pytest.mark.parametrize('B,npad,act', [(300, 320, 'sigmoid'), (129, 160, 'tanh'), (4000, 4096, 'sigmoid'), (256, 256, 'relu'), (40, 64, 'sigmoid')])
def test_batchnorm_on_a_padded_batch_equals_the_unpadded_one(B, npad, act, split):
    """abn_tower_desc.n_valid: a BatchNorm tower in training on [B real pairs | zero rows up to npad] per tower, told B through a
    device word, against the same tower on the B pairs alone -- embeddings of the real rows, running statistics, the loss and
    every gradient (gamma / beta included) bit for bit (the gradients: where both forms cut the same row slabs, a bucket of
    the planned passes; to rounding otherwise): the workgroups hold the same rows either way, the statistics and
    the backward's sums span the real rows only, the padded rows get no gradient and give none."""
    kw = dict(input_dim=40, num_hidden_layers=2, hidden_dim=500 if B > 2000 else 72, output_dim=100 if B > 2000 else 36,
              activation_layer=act, p_dropout=0.0, batch_norm=True)
    rng = np.random.default_rng(B)
    xa, xb = rng.standard_normal((B, 40)).astype(np.float32), rng.standard_normal((B, 40)).astype(np.float32)
    y = dev(rng.choice([1, -1], B))
    res = []
    for padded in (False, True):
        net, spec, p = build(kw, seed=B, precision=split)
        net.train()
        if padded:
            x12 = torch.zeros(2 * npad, 40, device='cuda')
            x12[:B], x12[npad:npad + B] = dev(xa), dev(xb)
            yy = torch.zeros(npad, dtype=y.dtype, device='cuda')
            yy[:B] = y
            nv = torch.tensor([B], dtype=torch.int32, device='cuda')
            emb, state = net.direct_forward(x12[:npad], x12[npad:], n_valid=nv)
            lv = net.direct_backward_loss(state, yy, 'coscos2', 0.0, False, n_valid=nv)
            e = torch.cat([emb[:B], emb[npad:npad + B]])
            assert float(emb[B:npad].abs().sum()) == 0.0 and float(emb[npad + B:].abs().sum()) == 0.0
        else:
            emb, state = net.direct_forward(dev(xa), dev(xb))
            lv = net.direct_backward_loss(state, y, 'coscos2', 0.0, False)
            e = emb
        assert lv is not None
        res.append((e.clone(), float(lv), {k: v.clone() for k, v in net.state_dict().items() if 'running' in k or 'tracked' in k},
                    {k: q.grad.clone() for k, q in net.named_parameters()}))
    (e0, l0, s0, g0), (e1, l1, s1, g1) = res
    assert torch.equal(e0, e1) and l0 == l1
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k
    same_slabs = (2 * B + 63) // 64 * 64 == 2 * npad       # (the weight gradients' row slabs are cut by the row count rounded up to 64)
    gmax = max(float(g.abs().max()) for g in g0.values())
    for k in g0:
        if same_slabs:
            assert torch.equal(g0[k], g1[k]), k
        elif float(g0[k].abs().max()) > 1e-5 * gmax:       # (a Linear bias in front of BatchNorm: a mathematically zero gradient, rounding noise)
            assert float((g0[k] - g1[k]).abs().max()) <= 2e-6 * max(float(g0[k].abs().max()), 1e-2 * gmax), k


def test_n_valid_where_it_does_not_apply(split):
    """abn_tower_desc.n_valid outside the BatchNorm layer launches: a BatchNorm tower with a width that is no multiple of 4
    (the per-layer kernels: no real-row count inside their statistics) refuses it loudly, and so says the query the trainer
    asks first; a tower without BatchNorm never sees it (its rows do not see each other: the Python layer does not pass it on)."""
    from abnet3_amd import _lib
    nv = torch.tensor([40], dtype=torch.int32, device='cuda')
    x12 = torch.zeros(128, 40, device='cuda')
    x12[:40], x12[64:104] = torch.randn(40, 40, device='cuda'), torch.randn(40, 40, device='cuda')
    kw = dict(input_dim=40, num_hidden_layers=1, hidden_dim=72, output_dim=36, activation_layer='sigmoid', p_dropout=0.0)
    odd, spec, p = build(dict(kw, hidden_dim=74, batch_norm=True), seed=1, precision=split)
    odd.train()
    assert not odd.takes_padded_batch_norm(x12, 64)
    with pytest.raises(_lib.HipLibraryError, match='n_valid'):
        odd.direct_forward(x12[:64], x12[64:], n_valid=nv)
    net, spec, p = build(dict(kw, batch_norm=True), seed=1, precision=split)
    net.train()
    assert net.takes_padded_batch_norm(x12, 64)
    plain, spec, p = build(dict(kw, batch_norm=False), seed=1, precision=split)
    plain.train()
    e0, _ = plain.direct_forward(x12[:64], x12[64:])
    e1, st = plain.direct_forward(x12[:64], x12[64:], n_valid=nv)
    assert torch.equal(e0, e1) and st[1].n_valid is None


@pytest.mark.parametrize('oname', ['adadelta', 'sgd'])
def test_persistent_weight_image_follows_every_kind_of_update(oname, monkeypatch, split):
    """abn_tower_desc.wpack: the forward skips its pack launch while the image is known to match the
    parameters (repeated forwards with unchanged weights).  Optimizer launches, torch-side writes and
    state_dict loads must make the next forward rebuild it.  Every variant against the same run without the image."""
    import abnet3_amd.loss as L
    from abnet3_amd.trainer import TrainerSiamese
    kw = dict(input_dim=40, num_hidden_layers=1, hidden_dim=96, output_dim=32, activation_layer='sigmoid',
              p_dropout=0.0, batch_norm=False)
    rng = np.random.default_rng(2)
    B = 64
    batches = [(dev(rng.standard_normal((B, 40)).astype(np.float32)), dev(rng.standard_normal((B, 40)).astype(np.float32)),
                dev(rng.choice([1, -1], B))) for _ in range(3)]
    outs = []
    for wpack in ('1', '0'):
        monkeypatch.setenv('ABN_WPACK', wpack)
        net, _, _ = build(kw, seed=1, precision=split)
        tr = TrainerSiamese(network=net, loss=L.coscos2(avg=False), optimizer_type=oname, lr=0.05, dataloader=None,
                            log_dir='/tmp/abn_runs')
        net.train()
        seg = net._segment_list()[0]
        losses = []
        for s in range(2):
            losses.append(float(tr.train_step(batches[s], True)))            # fused step: the image is stale afterwards
        assert (getattr(seg, '_wpack', None) is not None) == (wpack == '1')
        if wpack == '1':
            assert seg._wpack_key is None
            net.eval()
            with torch.no_grad():
                a = net.forward_once(batches[0][0])
                assert seg._wpack_key == seg._weights_key()                    # rebuilt; the next forward will not pack
                b = net.forward_once(batches[0][0])
            assert torch.equal(a, b)
            net.train()
        with torch.no_grad():
            net.input_emb[0].weight.mul_(1.5)                                  # a torch-side write
        losses.append(float(tr.train_step(batches[2], True)))
        tr.optimizer.zero_grad()
        e1, e2 = net(batches[0][0], batches[0][1])                             # autograd path + plain optimizer launch
        lv = L.coscos2(avg=False)(e1, e2, batches[0][2])
        lv.backward()
        tr.optimizer.step()
        if wpack == '1':
            assert seg._wpack_key is None
        losses.append(float(lv.detach()))
        losses.append(float(tr.train_step(batches[1], True)))
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        net.load_state_dict({k: v * 0.5 for k, v in sd.items()})
        net.eval()
        with torch.no_grad():
            emb = net.forward_once(batches[2][0]).cpu().numpy()
        outs.append((losses, emb, {k: v.detach().cpu().numpy() for k, v in net.named_parameters()}))
    assert np.allclose(outs[0][0], outs[1][0], rtol=1e-6), (outs[0][0], outs[1][0])
    assert rel_err(outs[0][1], outs[1][1]) < 1e-6
    for k in outs[0][2]:
        assert rel_err(outs[0][2][k], outs[1][2][k]) < 1e-6, k


@pytest.mark.parametrize('precision,NP', [('bf16x3', 3), ('f16x2', 2)])
def test_dropout_drawn_inside_the_kernels(precision, NP, monkeypatch):
    """p_dropout > 0 without mask tensors (abn_tower_desc.drop_seed): the multipliers are a hash of (seed, layer,
    row, feature) evaluated in the forward epilogue and again in the backward.  Recovered from the activations
    (tanh(z m) = 0 exactly where m = 0), they must (i) be Bernoulli(1 - p) / (1 - p), (ii) reproduce the embeddings
    through the oracle, (iii) be the ones the backward used: its gradients equal the oracle's with those masks,
    (iv) differ from one forward to the next."""
    import abnet3_amd.loss as L
    from abnet3_amd import _lib
    from oracle import siamese_np as O
    pdrop = 0.3
    kw = dict(input_dim=40, num_hidden_layers=2, hidden_dim=96, output_dim=36, activation_layer='tanh',
              p_dropout=pdrop, batch_norm=False)
    net, spec, p = build(kw, seed=12, precision=precision)
    lib = _lib.load()
    fn = lib.abn_tower_image_offset
    fn.restype = ctypes.c_int64
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int]
    B = 200
    R = 2 * B
    dims = [40, 96, 96, 96, 36]
    rng = np.random.default_rng(12)
    x1n, x2n = rng.standard_normal((B, 40)).astype(np.float32), rng.standard_normal((B, 40)).astype(np.float32)
    y = rng.choice([1, -1], B)
    net.train()
    seen = []
    for rep in range(2):
        for q in net.parameters():
            q.grad = None
        emb, state = net.direct_forward(dev(x1n), dev(x2n))
        seg, sv, gp = state
        assert type(sv.masks).__name__ == '_DropSeed'
        lv = net.direct_backward_loss(state, dev(y), 'coscos2', 0.0, False)
        assert lv is not None
        torch.cuda.synchronize()
        desc = seg.descriptor(False)
        ws16 = sv.ws.view(torch.int16).cpu().numpy().view(np.uint16)
        row_steps = (R + 31) // 32 * 2
        masks = []
        for l in range(3):          # hidden activations live in the transposed image of the next layer's input
            d = decode_t(ws16[2 * fn(ctypes.byref(desc), R, 2, 2, l + 1):], (dims[l + 1] + 1 + 31) // 32, row_steps, NP)
            masks.append((d[:dims[l + 1], :R].T != 0).astype(np.float32) / (1 - pdrop))
        e = emb.cpu().numpy()
        masks.append((e != 0).astype(np.float32) / (1 - pdrop))
        for m in masks:
            frac = float((m == 0).mean())
            assert abs(frac - pdrop) < 4 * np.sqrt(pdrop * (1 - pdrop) / m.size) + 1e-3, frac
        seen.append(masks)
        o1, c1 = O.tower_forward(p, x1n, spec, True, masks=[m[:B] for m in masks])
        o2, c2 = O.tower_forward(p, x2n, spec, True, masks=[m[B:] for m in masks])
        assert rel_err(e[:B], o1) < 1e-5 and rel_err(e[B:], o2) < 1e-5
        ol, d1, d2, _ = O.pair_loss(o1, o2, y, 'coscos2', 0.5, False)
        assert abs(float(lv) - ol) <= 1e-5 * abs(ol) + 1e-6
        og = {}
        O.tower_backward(p, c1, d1, spec, og)
        O.tower_backward(p, c2, d2, spec, og)
        check_grads({k: q.grad.cpu().numpy() for k, q in net.named_parameters()}, og, spec.param_keys(), False, tol=1e-4)
    assert any((a != b).any() for a, b in zip(*seen))
    # the autograd path and the per-layer GEMM path (mask tensors) still train
    e1, e2 = net(dev(x1n), dev(x2n))
    L.coscos2(avg=False)(e1, e2, dev(y)).backward()
    monkeypatch.setenv('ABN_PLANES', '0')
    e1, e2 = net(dev(x1n), dev(x2n))
    L.coscos2(avg=False)(e1, e2, dev(y)).backward()
    assert all(torch.isfinite(q.grad).all() for q in net.parameters())


def test_two_seeded_runs_are_bit_identical():
    """A race detector (tools/soak.py in small): the same seeded C2-shaped training run twice -- dropout drawn in
    the kernels, fused loss, deferred slab sum -- must produce bit-identical losses and parameters (every reduction
    in these kernels has a fixed order; nothing accumulates through atomics)."""
    import abnet3_amd.loss as L
    from abnet3_amd.model import SiameseNetwork
    from abnet3_amd.trainer import TrainerSiamese
    kw = dict(input_dim=40, num_hidden_layers=2, hidden_dim=500, output_dim=100, activation_layer='sigmoid',
              p_dropout=0.1, batch_norm=False)
    rng = np.random.default_rng(5)
    B = 1024
    pool = [(dev(rng.standard_normal((B, 40)).astype(np.float32)), dev(rng.standard_normal((B, 40)).astype(np.float32)),
             dev(rng.choice([1, -1], B))) for _ in range(3)]
    outs = []
    for run in range(2):
        torch.manual_seed(0)
        torch.cuda.manual_seed(0)
        net = SiameseNetwork(output_path='/tmp/abn_det', **kw)
        tr = TrainerSiamese(network=net, loss=L.coscos2(avg=False), optimizer_type='adadelta', lr=0.1, dataloader=None,
                            log_dir='/tmp/abn_runs')
        net.train()
        losses = torch.stack([tr.train_step(pool[i % 3], True) for i in range(60)])
        assert torch.isfinite(losses).all()
        outs.append((losses, [q.detach().clone() for q in net.parameters()]))
    assert torch.equal(outs[0][0], outs[1][0])
    assert all(torch.equal(a, b) for a, b in zip(outs[0][1], outs[1][1]))


@pytest.mark.parametrize('precision,tol', [('bf16x3', 1e-5), ('f16x2', 1e-5), ('bf16', 3e-2)])
@pytest.mark.parametrize('act', ['sigmoid', 'tanh', 'relu'])
@pytest.mark.parametrize('shape', SHAPES[:5])
def test_batch_norm_inference_forward_against_the_oracle(shape, act, precision, tol, monkeypatch):
    """Embedding extraction from a BatchNorm tower (eval mode, no gradient wanted): the running statistics
    are a per-feature affine map in the epilogue of the same single launch.  Against the numpy oracle and
    against the per-layer kernels; a backward through it stays refused; train mode stays on the per-layer path."""
    from oracle import siamese_np as O
    from abnet3_amd import _lib
    d_in, nh, hid, d_out, B = shape
    kw = dict(input_dim=d_in, num_hidden_layers=nh, hidden_dim=hid, output_dim=d_out, activation_layer=act,
              p_dropout=0.1, batch_norm=True)
    net, _, _ = build(kw, seed=B + 1, precision=precision)
    spec = O.TowerSpec(d_in, nh, hid, d_out, act, True)
    rng = np.random.default_rng(B)
    with torch.no_grad():           # statistics and affine parameters away from their initial 0 / 1
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                n = m.num_features
                m.running_mean.copy_(dev(rng.standard_normal(n).astype(np.float32) * 0.3))
                m.running_var.copy_(dev(rng.uniform(0.2, 3.0, n).astype(np.float32)))
                m.weight.copy_(dev(rng.uniform(0.5, 1.5, n).astype(np.float32)))
                m.bias.copy_(dev(rng.standard_normal(n).astype(np.float32) * 0.2))
    p = {k: v.detach().cpu().numpy().copy() for k, v in net.state_dict().items()}
    x = rng.standard_normal((B, d_in)).astype(np.float32)
    net.eval()
    seg = net._segment_list()[0]
    lib = _lib.load()
    xd = dev(x)
    probe = seg.descriptor(with_grads=False, forward_only=True)
    assert lib.abn_tower_uses_planes(ctypes.byref(probe), B, _lib.ptr(xd), None, _lib.ptr(xd), 0) == 1
    assert lib.abn_tower_uses_planes(ctypes.byref(probe), B, _lib.ptr(xd), None, _lib.ptr(xd), 1) == 0
    probe = seg.descriptor(with_grads=False, forward_only=False)
    assert lib.abn_tower_uses_planes(ctypes.byref(probe), B, _lib.ptr(xd), None, _lib.ptr(xd), 0) == 0
    with torch.no_grad():
        e = net.forward_once(xd).cpu().numpy()
        assert _lib.last_forward_path() == 4
        e1, e2 = net(xd, dev(x[::-1].copy()))
        assert _lib.last_forward_path() == 4
    monkeypatch.setenv('ABN_PLANES', '0')
    with torch.no_grad():
        ref = net.forward_once(xd).cpu().numpy()
        assert _lib.last_forward_path() == 0
    monkeypatch.setenv('ABN_PLANES', '1')
    o, _ = O.tower_forward(p, x, spec, False)
    # (8-bit operands: the normalisation multiplies a layer's rounding error by up to gamma / sqrt(running_var))
    tol = tol if precision != 'bf16' else 3 * tol
    assert rel_err(e, o) < tol
    assert rel_err(e, ref) < tol
    assert rel_err(e1.cpu().numpy(), o) < tol and rel_err(e2.cpu().numpy(), o[::-1]) < tol
    # the running statistics are inputs here: untouched
    for k, v in net.state_dict().items():
        assert np.array_equal(v.cpu().numpy(), p[k]), k
    # with a gradient wanted the eval forward stays on the per-layer kernels (whose backward refuses, as before)
    xg = xd.clone().requires_grad_(True)
    eg = net.forward_once(xg)
    assert _lib.last_forward_path() == 0
    assert rel_err(eg.detach().cpu().numpy(), ref) == 0.0
    with pytest.raises(NotImplementedError):
        eg.sum().backward()


@pytest.mark.parametrize('precision', ['bf16x3', 'f16x2', 'bf16'])
@pytest.mark.parametrize('shape', SHAPES)
def test_inference_instantiation_equals_the_training_forward(shape, precision):
    """torch.no_grad() forwards run the instantiation without dropout and transposed images: same arithmetic,
    the same bits as the forward a backward will follow."""
    d_in, nh, hid, d_out, B = shape
    kw = dict(input_dim=d_in, num_hidden_layers=nh, hidden_dim=hid, output_dim=d_out, activation_layer='tanh',
              p_dropout=0.0, batch_norm=False)
    net, _, _ = build(kw, seed=B, precision=precision)
    x = dev(np.random.default_rng(B).standard_normal((B, d_in)).astype(np.float32))
    from abnet3_amd import _lib
    lib = _lib.load()
    for mode in (net.train, net.eval):
        mode()
        a = net.forward_once(x).detach()
        assert _lib.last_forward_path() == 2
        with torch.no_grad():
            b = net.forward_once(x)
        assert _lib.last_forward_path() == (2 if net.training else 3)
        assert torch.equal(a, b)


BN_SHAPES = [  # (input, hidden layers, hidden, output, B): whole and ragged workgroups per forward_once call
    (40, 2, 500, 100, 64),
    (32, 1, 64, 32, 33),
    (64, 1, 512, 128, 32),
    (128, 3, 288, 36, 70),
    (4, 0, 8, 4, 5),
    (40, 1, 96, 48, 250),
]


@pytest.mark.parametrize('p_drop', [0.0, 0.25])
@pytest.mark.parametrize('act', ['sigmoid', 'tanh', 'relu'])
@pytest.mark.parametrize('shape', BN_SHAPES)
def test_batch_norm_training_step_against_the_oracle(shape, act, p_drop, split):
    """Linear -> Dropout -> BatchNorm -> activation in training: one operand-plane launch per layer, the
    batch statistics from per-workgroup column sums.  Embeddings, loss, running statistics and every
    gradient against the numpy oracle, with shared dropout masks."""
    import abnet3_amd.loss as L
    from abnet3_amd import _lib
    from oracle import siamese_np as O
    d_in, nh, hid, d_out, B = shape
    kw = dict(input_dim=d_in, num_hidden_layers=nh, hidden_dim=hid, output_dim=d_out, activation_layer=act,
              p_dropout=p_drop, batch_norm=True)
    net, _, _ = build(kw, seed=B + 3, precision=split)
    spec = O.TowerSpec(d_in, nh, hid, d_out, act, True)
    rng = np.random.default_rng(B + nh)
    with torch.no_grad():           # affine parameters away from 1 / 0, a bias that shifts the column means
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                n = m.num_features
                m.weight.copy_(dev(rng.uniform(0.5, 1.5, n).astype(np.float32)))
                m.bias.copy_(dev(rng.standard_normal(n).astype(np.float32) * 0.2))
            if isinstance(m, torch.nn.Linear):
                m.bias.copy_(dev(rng.standard_normal(m.out_features).astype(np.float32) * 3.0))
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
    assert _lib.last_forward_path() in (5, 7)      # (7: the resident tower, from 8 workgroups on)
    lv = L.coscos2(avg=False)(e1, e2, dev(y))
    lv.backward()
    assert _lib.last_backward_path() in (5, 7)
    o1, c1 = O.tower_forward(p, x1, spec, True, masks=[m[:B] for m in masks] if masks else None)
    o2, c2 = O.tower_forward(p, x2, spec, True, masks=[m[B:] for m in masks] if masks else None)
    ol, d1, d2, _ = O.pair_loss(o1, o2, y, 'coscos2', 0.5, False)
    og = {}
    O.tower_backward(p, c1, d1, spec, og)
    O.tower_backward(p, c2, d2, spec, og)
    assert rel_err(e1.detach().cpu().numpy(), o1) < 2e-5
    assert rel_err(e2.detach().cpu().numpy(), o2) < 2e-5
    assert abs(float(lv.detach()) - ol) <= 1e-4 * abs(ol) + 1e-6
    sd = net.state_dict()
    for k in p:                     # the oracle updated its running statistics in place, once per call
        if 'running' in k:
            assert rel_err(sd[k].cpu().numpy(), p[k]) < 1e-5, k
        if 'num_batches' in k:
            assert int(sd[k]) == int(p[k]) == 2, k
    grads = {k: q.grad.cpu().numpy() for k, q in net.named_parameters()}
    check_grads(grads, og, spec.param_keys(), not p_drop, tol=2e-4)


def test_batch_norm_input_gradient_and_fallbacks(monkeypatch, split):
    """d loss / d input through the BatchNorm launches (layer 0's product with W_0), against the per-layer kernels,
    with ragged calls (50 rows each: a whole and a short workgroup per call); a single ragged forward_once call."""
    from abnet3_amd import _lib
    kw = dict(input_dim=40, num_hidden_layers=1, hidden_dim=96, output_dim=32, activation_layer='tanh', p_dropout=0.0,
              batch_norm=True)
    rng = np.random.default_rng(4)
    x1 = rng.standard_normal((50, 40)).astype(np.float32)
    x2 = rng.standard_normal((50, 40)).astype(np.float32)
    w = dev(rng.standard_normal((50, 32)).astype(np.float32))
    res = []
    for planes in ('1', '0'):
        monkeypatch.setenv('ABN_BN_PLANES', planes)
        net, _, _ = build(kw, seed=9, precision=split)
        net.train()
        a, b = dev(x1).requires_grad_(True), dev(x2).requires_grad_(True)
        e1, e2 = net(a, b)
        ((e1 * w).sum() + (e2 * e2 * w).sum()).backward()
        assert _lib.last_backward_path() in ((5, 7) if planes == '1' else (0,))
        res.append([a.grad.cpu().numpy(), b.grad.cpu().numpy()] + [q.grad.cpu().numpy() for q in net.parameters()])
    gmax = max(np.abs(v).max() for v in res[1])
    for u, v in zip(*res):
        if np.abs(v).max() < 1e-5 * gmax:           # the Linear biases in front of a BatchNorm: zero but for rounding
            assert np.abs(u).max() < 1e-5 * gmax
        else:
            assert rel_err(u, v, floor=1e-3 * np.abs(v).max()) < 1e-4
    outs = []
    for planes in ('1', '0'):
        monkeypatch.setenv('ABN_BN_PLANES', planes)
        net, _, _ = build(kw, seed=9, precision=split)
        net.train()
        e = net.forward_once(dev(x1[:37]))
        assert _lib.last_forward_path() in ((5, 7) if planes == '1' else (0,))
        (e * e).sum().backward()
        assert _lib.last_backward_path() in ((5, 7) if planes == '1' else (0,))
        outs.append([e.detach().cpu().numpy()] + [q.grad.cpu().numpy() for q in net.parameters()]
                    + [v.cpu().numpy() for k, v in net.state_dict().items() if 'running' in k])
    gmax = max(np.abs(v).max() for v in outs[1][1:])
    for u, v in zip(*outs):
        if np.abs(v).max() < 1e-5 * gmax:
            continue
        assert rel_err(u, v, floor=1e-3 * np.abs(v).max()) < 1e-4


def test_batch_norm_training_in_the_bf16_arithmetic(monkeypatch):
    """The single-plane instantiations of the BatchNorm launches: against the per-layer kernels in the same
    arithmetic (8-bit operands on both sides, different summation orders)."""
    import abnet3_amd.loss as L
    from abnet3_amd import _lib
    kw = dict(input_dim=40, num_hidden_layers=2, hidden_dim=500, output_dim=100, activation_layer='sigmoid', p_dropout=0.0,
              batch_norm=True)
    rng = np.random.default_rng(11)
    B = 128
    x1, x2 = dev(rng.standard_normal((B, 40)).astype(np.float32)), dev(rng.standard_normal((B, 40)).astype(np.float32))
    y = dev(rng.choice([1, -1], B))
    res = []
    for planes in ('1', '0'):
        monkeypatch.setenv('ABN_BN_PLANES', planes)
        net, _, _ = build(kw, seed=2, precision='bf16')
        net.train()
        e1, e2 = net(x1, x2)
        assert _lib.last_forward_path() in ((5, 7) if planes == '1' else (0,))
        lv = L.coscos2(avg=False)(e1, e2, y)
        lv.backward()
        res.append((e1.detach().cpu().numpy(), float(lv.detach()), {k: q.grad.cpu().numpy() for k, q in net.named_parameters()},
                    {k: v.cpu().numpy() for k, v in net.state_dict().items() if 'running' in k}))
    assert rel_err(res[0][0], res[1][0]) < 3e-2
    assert abs(res[0][1] - res[1][1]) < 3e-2 * abs(res[1][1])
    for k, v in res[1][3].items():
        assert rel_err(res[0][3][k], v) < 1e-2, k
    gmax = max(np.abs(v).max() for v in res[1][2].values())
    for k, v in res[1][2].items():
        if np.abs(v).max() < 1e-4 * gmax or is_pre_bn_bias(k, True):      # (zero but for rounding)
            continue
        a, b = res[0][2][k].ravel().astype(np.float64), v.ravel().astype(np.float64)
        assert a @ b / (np.linalg.norm(a) * np.linalg.norm(b)) > 0.98, k
        assert abs(np.linalg.norm(a) / np.linalg.norm(b) - 1) < 0.1, k


def test_batch_norm_with_dropout_drawn_inside_the_kernels(split):
    """p_dropout > 0 on a BatchNorm tower: the per-layer launches hash the same per-forward seed as the
    single-launch chains (forward epilogue, regenerated by the backward).  The masks are recovered from the
    pre-normalisation values the forward leaves (a dropped entry is an exact zero) and fed back as tensors:
    same embeddings, same gradients."""
    from abnet3_amd import _lib
    p_drop = 0.25
    kw = dict(input_dim=40, num_hidden_layers=1, hidden_dim=96, output_dim=32, activation_layer='tanh', p_dropout=p_drop,
              batch_norm=True)
    net, _, _ = build(kw, seed=3, precision=split)
    net.train()
    rng = np.random.default_rng(8)
    B = 128                                                     # 256 rows: whole workgroups for every call count
    x1, x2 = dev(rng.standard_normal((B, 40)).astype(np.float32)), dev(rng.standard_normal((B, 40)).astype(np.float32))
    d_out = dev(rng.standard_normal((2 * B, 32)).astype(np.float32))
    seg = net._segment_list()[0]
    lib = _lib.load()
    probe = seg.descriptor(with_grads=False)
    assert lib.abn_tower_uses_planes(ctypes.byref(probe), 2 * B, _lib.ptr(x1), _lib.ptr(x2), _lib.ptr(x1), 1) == 1
    emb, state = net.direct_forward(x1, x2)
    assert _lib.last_forward_path() in (5, 7)      # (7: the resident tower, from 8 workgroups on)
    sv = state[1]
    assert type(sv.masks).__name__ == '_DropSeed'
    net.direct_backward(state, d_out)
    assert _lib.last_backward_path() in (5, 7)
    emb = emb.clone()
    grads = {k: q.grad.clone() for k, q in net.named_parameters()}
    fn = lib.abn_tower_image_offset
    fn.restype = ctypes.c_int64
    desc = seg.descriptor(with_grads=False, masks=sv.masks)
    masks = []
    for l, w in enumerate((96, 96, 32)):
        off = fn(ctypes.byref(desc), ctypes.c_int64(2 * B), ctypes.c_int64(2), 4, l)
        assert off >= 0
        z = sv.ws[off:off + 2 * B * w].view(2 * B, w)
        m = (z != 0).float() / (1 - p_drop)
        frac = float((z == 0).float().mean())
        assert abs(frac - p_drop) < 0.03, (l, frac)
        masks.append(m)
    emb_b, state_b = net.direct_forward(x1, x2)
    assert not torch.equal(emb, emb_b)                          # another forward, another seed
    net._mask_override = masks
    for q in net.parameters():
        q.grad = None
    emb_t, state_t = net.direct_forward(x1, x2)
    net.direct_backward(state_t, d_out)
    assert rel_err(emb_t.cpu().numpy(), emb.cpu().numpy()) < 1e-6
    for k, q in net.named_parameters():
        assert rel_err(q.grad.cpu().numpy(), grads[k].cpu().numpy(), floor=1e-6 * float(grads[k].abs().max()) + 1e-30) < 1e-5, k


def test_batch_norm_dropout_train_mode_without_gradients(tmp_path, split):
    """The reference's standard configuration (batch_norm=True, p_dropout=0.1): TrainerBuilder.train() starts with
    optimize_model(do_training=False) -- net.train() under torch.no_grad() (abnet3/trainer.py:137,226-235).  A
    forward that keeps nothing for a backward cannot use the BatchNorm training launches, so the dropout probe
    must ask with the call's own forward_only: the step falls back to mask tensors on the per-layer kernels
    instead of failing with 'in-kernel dropout needs the operand-plane kernels'."""
    from abnet3_amd import _lib
    from abnet3_amd.loss import coscos2
    from abnet3_amd.trainer import TrainerSiamese
    kw = dict(input_dim=40, num_hidden_layers=1, hidden_dim=96, output_dim=32, activation_layer='sigmoid', p_dropout=0.1,
              batch_norm=True, output_path=str(tmp_path / 'net'))
    net, _, _ = build(kw, seed=5, precision=split)
    rng = np.random.default_rng(2)
    B = 160                                                    # 320 rows: the planes kernels take it by themselves
    batch = (dev(rng.standard_normal((B, 40)).astype(np.float32)), dev(rng.standard_normal((B, 40)).astype(np.float32)),
             dev(rng.choice([1.0, -1.0], B)))

    class Loader(object):
        def batch_iterator(self, train_mode=True):
            for _ in range(3):
                yield batch

        def whoami(self):
            return {'class_name': 'Loader'}
    tr = TrainerSiamese(network=net, loss=coscos2(avg=False), optimizer_type='adadelta', lr=0.1, num_epochs=2, patience=5,
                        dataloader=Loader(), log_dir=str(tmp_path / 'runs'))
    net.train()
    lib = _lib.load()
    v = tr.train_step(batch, False)                            # train mode, no gradients
    assert _lib.last_forward_path() == 0 and np.isfinite(float(v))
    v = tr.train_step(batch, True)                             # the same tower in a real step: the BatchNorm launches
    assert _lib.last_forward_path() in (5, 7) and np.isfinite(float(v))
    tr.train()                                                 # the reference's whole loop, first pass included
    assert len(tr.train_losses) == 3 and all(np.isfinite(tr.train_losses))


def test_persistent_weight_image_sees_writes_behind_torch(split):
    """Writes that do not bump the parameters' version counters -- init_weight_method's .data writes, a copy into
    the flat buffer (what parallel.broadcast_parameters does) -- still invalidate the persistent weight image."""
    kw = dict(input_dim=40, num_hidden_layers=1, hidden_dim=64, output_dim=32, activation_layer='tanh', p_dropout=0.0)
    net, _, _ = build(kw, seed=11, precision=split)
    net.eval()
    x = dev(np.random.default_rng(0).standard_normal((64, 40)).astype(np.float32))
    with torch.no_grad():
        a = net.forward_once(x).clone()
        assert torch.equal(net.forward_once(x), a)             # image reused
        torch.manual_seed(99)
        net.apply(net.init_weight_method)                      # layer.weight.data / bias.data writes
        b = net.forward_once(x).clone()
        assert not torch.equal(a, b)
        fresh, _, _ = build(kw, seed=11, precision=split)
        fresh.eval()
        fresh.load_state_dict(net.state_dict())
        assert torch.equal(fresh.forward_once(x), b)
        net.flat_parameters().copy_(net.flat_parameters() * 0.5)   # behind every parameter's back
        c = net.forward_once(x).clone()
        fresh.load_state_dict(net.state_dict())
        assert torch.equal(fresh.forward_once(x), c) and not torch.equal(b, c)


def test_two_part_backward_is_refused_where_it_cannot_run(split):
    """abn_tower_desc.wgrad_part (the data-parallel backward in two calls) exists on the operand-plane launches of towers
    without BatchNorm; a BatchNorm tower's backward says so instead of running half a backward."""
    from abnet3_amd import _lib, model as M
    kw = dict(input_dim=40, num_hidden_layers=1, hidden_dim=96, output_dim=32, activation_layer='sigmoid',
              p_dropout=0.0, batch_norm=True)
    net, _, _ = build(kw, seed=21, precision=split)
    rng = np.random.default_rng(21)
    x1, x2 = dev(rng.standard_normal((160, 40)).astype(np.float32)), dev(rng.standard_normal((160, 40)).astype(np.float32))
    net.train()
    out, (seg, sv, gp) = net.direct_forward(x1, x2)
    plain = seg.descriptor

    def two_part(*a, **k):
        d = plain(*a, **k)
        d.wgrad_part, d.wgrad_split = 1, 1
        return d
    seg.descriptor = two_part
    with pytest.raises(_lib.HipLibraryError, match='wgrad_part'):
        M._segment_backward(seg, sv, torch.randn_like(out), gp, False)


@pytest.mark.parametrize('rows', [300, 4096])        # the layer-per-launch kernels / the single-launch chains
@pytest.mark.parametrize('act', ['relu', 'sigmoid'])
def test_f16x2_scales_follow_the_data(rows, act, split, monkeypatch):
    """The split arithmetics under extreme ranges -- fp16 x 2 lives on its power-of-two scales (per batch row, per 32-row
    weight block, per slab of the weight gradient; the column of ones that yields the bias gradient keeps a scale of its
    own).  Inputs whose rows differ by ten orders of magnitude, zero rows, weight blocks six orders of magnitude apart
    under an unbounded activation (activations up to 1e9), a loss gradient of 1e-9: embeddings (per row) and every
    gradient tensor against a float64 evaluation, as close as at ordinary ranges."""
    monkeypatch.setenv('ABN_WIDE', '1' if rows < 1000 else '0')
    kw = dict(input_dim=40, num_hidden_layers=2, hidden_dim=288, output_dim=64, activation_layer=act,
              p_dropout=0.0, batch_norm=False)
    if act == 'relu':
        kw['last_non_linearity'] = None
    rng = np.random.default_rng(rows)
    x = rng.standard_normal((rows, 40)).astype(np.float32)
    # rows from 1e-6 to 1e4 (a sigmoid is not driven into saturation: its derivative a (1 - a) from a float32 a is then
    # noise in the reference's arithmetic too -- rows up to 3)
    x *= (10.0 ** rng.uniform(-6, 4 if act == 'relu' else 0.5, size=(rows, 1))).astype(np.float32)
    x[::17] = 0.0                                                             # and zero rows
    net, _, _ = build(kw, seed=3, precision=split)
    with torch.no_grad():
        for k, q in net.named_parameters():
            if k.endswith('weight') and act == 'relu':
                sc = torch.ones(q.shape[0], 1, device=q.device)
                sc[32:64] = 1e-3
                sc[64:96] = 1e3
                q.mul_(sc)
        net.weights_changed_behind_torch()
    g = (rng.standard_normal((rows, 64)) * 1e-9).astype(np.float32)
    net.train()
    e = net.forward_once(dev(x))
    e.backward(dev(g))
    # the same network in float64 on the CPU
    Ws = [q.detach().double().cpu() for k, q in net.named_parameters() if k.endswith('weight')]
    bs = [q.detach().double().cpu() for k, q in net.named_parameters() if k.endswith('bias')]
    for t in Ws + bs:
        t.requires_grad_(True)
    h = torch.from_numpy(x).double()
    for l in range(4):
        h = h @ Ws[l].T + bs[l]
        if act == 'sigmoid':
            h = torch.sigmoid(h)
        elif l < 3:
            h = torch.relu(h)
    h.backward(torch.from_numpy(g).double())
    ref = h.detach().numpy()
    got = e.detach().cpu().numpy().astype(np.float64)
    assert np.isfinite(got).all()
    rmax = np.abs(ref).max(axis=1, keepdims=True)          # a row of 1e-6 inputs is judged against its own size
    ok = rmax[:, 0] > 0
    assert (np.abs(got - ref)[ok] / rmax[ok]).max() < 1e-5
    grads = [q.grad.double().cpu().numpy() for k, q in net.named_parameters()]
    refs = [t for pair in zip(Ws, bs) for t in pair]
    for (k, _), mine, r in zip(net.named_parameters(), grads, refs):
        d = np.abs(mine - r.grad.numpy()) / np.abs(r.grad.numpy()).max()
        # (ReLU: a pre-activation that cancels to ~0 may take the other side of relu' in another arithmetic -- one row's
        # term in one feature's gradients; the bias is zero here, such rows exist)
        allowed = max(1, d.size // 2000) if act == 'relu' else 0
        assert (d > 2e-5).sum() <= allowed, (k, d.max(), int((d > 2e-5).sum()))
        assert d.max() < 2e-2, (k, d.max())


@pytest.mark.parametrize('seed', range(10))
def test_two_part_backward_is_bit_identical_for_random_towers(seed, split, monkeypatch):
    """abn_tower_desc.wgrad_part on random towers (1 .. 5 layers, random widths, the chains and the layer-per-launch kernels,
    every cut 0 .. n_layers): the two calls leave the loss and every gradient of the one call, bit for bit."""
    import abnet3_amd.loss as L
    from abnet3_amd.trainer import TrainerSiamese
    rng = np.random.default_rng(100 + seed)
    nh = int(rng.integers(0, 4))
    kw = dict(input_dim=int(rng.integers(2, 80)) * 4, num_hidden_layers=nh, hidden_dim=int(rng.integers(2, 128)) * 4,
              output_dim=int(rng.integers(2, 40)) * 4, activation_layer=str(rng.choice(['sigmoid', 'tanh'])), p_dropout=0.0, batch_norm=False)
    B = int(rng.choice([48, 300, 1700]))
    monkeypatch.setenv('ABN_WIDE', str(int(rng.integers(0, 2))))
    net, _, _ = build(kw, seed=seed, precision=split)
    tr = TrainerSiamese(network=net, loss=L.coscos2(avg=False), optimizer_type='sgd', lr=0.0, dataloader=None, log_dir='/tmp/abn_runs')
    x1, x2 = dev(rng.standard_normal((B, kw['input_dim'])).astype(np.float32)), dev(rng.standard_normal((B, kw['input_dim'])).astype(np.float32))
    y = dev(rng.choice([1.0, -1.0], B))
    net.train()
    n_layers = nh + 2

    def run(cut):
        emb, state = net.direct_forward(x1, x2)
        tr.optimizer.zero_grad()
        loss = net.direct_backward_loss(state, y, 'coscos2', 0.0, False, defer_reduce=False, wgrad_split=cut)
        assert loss is not None
        if cut is not None:
            net.direct_backward_lower()
        return loss.clone(), [p.grad.clone() for p in net.parameters()]
    l0, g0 = run(None)
    for cut in range(n_layers + 1):
        l1, g1 = run(cut)
        assert torch.equal(l0, l1), cut
        for a, b in zip(g0, g1):
            assert torch.equal(a, b), cut
