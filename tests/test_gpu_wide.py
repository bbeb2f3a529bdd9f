"""The layer-per-launch kernels of csrc/tower_wide.h (small batches in the bf16x3 / bf16 arithmetic: a layer's
output blocks dealt over several workgroups per 32 rows, every forward_once call padded to whole workgroups):
against the numpy oracle on the shapes that walk the operand-plane branches, against the single-launch chains,
ragged and multiple calls, the input gradient, dropout from the per-forward seed, and the bit-identity of a
batch with its zero-padded form (what the trainer's captured steps rely on).  Needs an MI355X: -m gpu."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import rel_err, check_grads
from test_gpu_planes import SHAPES, build, dev

pytestmark = pytest.mark.gpu


def _lib():
    from abnet3_amd import _lib as L
    return L, L.load()


@pytest.mark.parametrize('precision,tol,gtol', [('bf16x3', 1e-5, 1e-4), ('f16x2', 1e-5, 1e-4), ('bf16', 6e-2, None)])      # (bf16: ~3 digits per product, four 280..500-wide layers)
@pytest.mark.parametrize('act', ['sigmoid', 'tanh', 'relu'])
@pytest.mark.parametrize('shape', SHAPES + [(280, 2, 500, 100, 300), (40, 2, 500, 100, 40), (64, 1, 512, 128, 97)])
def test_forward_loss_and_gradients_against_the_oracle(shape, act, precision, tol, gtol):
    import abnet3_amd.loss as L
    from abnet3_amd.trainer import TrainerSiamese
    from oracle import siamese_np as O
    d_in, nh, hid, d_out, B = shape
    if nh > 8 and (act == 'sigmoid' or precision == 'bf16'):
        pytest.skip('see tests/test_gpu_planes.py')
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
    lib = _lib()[1]
    e1, e2 = net(dev(x1), dev(x2))
    assert _lib()[0].last_forward_path() == 6
    lv = L.coscos2(avg=False)(e1, e2, dev(y))
    lv.backward()
    assert _lib()[0].last_backward_path() == 6
    o1, c1 = O.tower_forward(p, x1, spec, True)
    o2, c2 = O.tower_forward(p, x2, spec, True)
    ol, d1, d2, _ = O.pair_loss(o1, o2, y, 'coscos2', 0.5, False)
    og = {}
    O.tower_backward(p, c1, d1, spec, og)
    O.tower_backward(p, c2, d2, spec, og)
    assert rel_err(e1.detach().cpu().numpy(), o1) < tol and rel_err(e2.detach().cpu().numpy(), o2) < tol
    assert abs(float(lv.detach()) - ol) <= 10 * tol * abs(ol) + 1e-6
    grads = {k: q.grad.cpu().numpy() for k, q in net.named_parameters()}
    if precision != 'bf16':
        check_grads(grads, og, spec.param_keys(), False, tol=gtol)
    else:
        for k in spec.param_keys()[-2:]:
            a, b = grads[k].ravel().astype(np.float64), og[k].ravel().astype(np.float64)
            assert a @ b / (np.linalg.norm(a) * np.linalg.norm(b)) > 0.9, k
    # the trainer's direct step (pair loss inside the top data-gradient launch, deferred reduction): same gradients
    tr = TrainerSiamese(network=net, loss=L.coscos2(avg=False), optimizer_type='sgd', lr=0.0, dataloader=None, log_dir='/tmp/abn_runs')
    lv2 = tr.train_step((dev(x1), dev(x2), dev(y)), True)
    assert _lib()[0].last_backward_path() == 6
    assert abs(float(lv2) - float(lv.detach())) <= 1e-6 * abs(float(lv2))
    step_grads = {k: q.grad.cpu().numpy() for k, q in net.named_parameters()}
    if precision != 'bf16':      # the step's own weight-gradient launch (fp16 x 2: tower_wgrad_step.h, all rows in one sum) against the oracle too
        check_grads(step_grads, og, spec.param_keys(), False, tol=gtol)
    for k in step_grads:
        # (two launches that add the same products in different orders and scale their operands over different row ranges --
        # one power of two per image for all rows here, one per 128-row slab there: each is judged against the oracle
        # above; between themselves a cancelled sum like the output layer's gradient sits 1e-5 of its largest entry apart)
        assert rel_err(step_grads[k], grads[k], floor=1e-30) < 3e-5, k


@pytest.mark.parametrize('rows', [1, 31, 33, 100, 257])
def test_forward_once_and_inference(rows, split):
    """One call of any row count; torch.no_grad() (nothing kept for a backward) gives the same bits."""
    kw = dict(input_dim=40, num_hidden_layers=2, hidden_dim=100, output_dim=36, activation_layer='tanh', p_dropout=0.0)
    net, spec, p = build(kw, seed=rows, precision=split)
    from oracle import siamese_np as O
    x = np.random.default_rng(rows).standard_normal((rows, 40)).astype(np.float32)
    lib = _lib()[1]
    for mode in (net.train, net.eval):
        mode()
        a = net.forward_once(dev(x)).detach()
        assert _lib()[0].last_forward_path() == 6
        with torch.no_grad():
            b = net.forward_once(dev(x))
        assert torch.equal(a, b)
        o, _ = O.tower_forward(p, x, spec, False)
        assert rel_err(a.cpu().numpy(), o) < 1e-5


@pytest.mark.parametrize('B,hid,d_in', [(333, 500, 280), (1024, 128, 280), (1200, 128, 40), (640, 500, 280), (130, 64, 40)])
def test_agrees_with_the_single_launch_chains(B, hid, d_in, monkeypatch, split):
    """The same step on the chains (ABN_WIDE=0) and on the layer-per-launch kernels (8, 4, 3 and 2 workgroups per
    row block): equal up to the order of the sums."""
    kw = dict(input_dim=d_in, num_hidden_layers=2, hidden_dim=hid, output_dim=100, activation_layer='sigmoid', p_dropout=0.0)
    rng = np.random.default_rng(2)
    x1, x2 = dev(rng.standard_normal((B, d_in)).astype(np.float32)), dev(rng.standard_normal((B, d_in)).astype(np.float32))
    y = dev(rng.choice([1.0, -1.0], B))
    monkeypatch.setenv('ABN_FUSED_MIN_ROWS', '0')
    res = []
    for wide in ('1', '0'):
        monkeypatch.setenv('ABN_WIDE', wide)
        net, _, _ = build(kw, seed=7, precision=split)
        net.train()
        emb, st = net.direct_forward(x1, x2)
        assert _lib()[0].last_forward_path() == (6 if wide == '1' else 2)
        loss = net.direct_backward_loss(st, y, 'coscos2', 0.0, False)
        res.append((emb.clone(), float(loss), {k: q.grad.clone() for k, q in net.named_parameters()}))
    (ea, la, ga), (eb, lb, gb) = res
    assert rel_err(ea.cpu().numpy(), eb.cpu().numpy()) < 2e-6 and abs(la - lb) <= 2e-6 * abs(lb)
    for k in ga:
        assert rel_err(ga[k].cpu().numpy(), gb[k].cpu().numpy(), floor=1e-6 * float(gb[k].abs().max()) + 1e-30) < 2e-4, k


def test_input_gradient(split):
    import abnet3_amd.loss as L
    from oracle import siamese_np as O
    kw = dict(input_dim=40, num_hidden_layers=1, hidden_dim=72, output_dim=36, activation_layer='tanh', p_dropout=0.0)
    net, spec, p = build(kw, seed=6, precision=split)
    B = 45
    rng = np.random.default_rng(6)
    x1n, x2n = rng.standard_normal((B, 40)).astype(np.float32), rng.standard_normal((B, 40)).astype(np.float32)
    y = rng.choice([1, -1], B)
    x1, x2 = dev(x1n).requires_grad_(True), dev(x2n).requires_grad_(True)
    net.train()
    e1, e2 = net(x1, x2)
    L.coscos2(avg=False)(e1, e2, dev(y)).backward()
    assert _lib()[0].last_backward_path() == 6
    o1, c1 = O.tower_forward(p, x1n, spec, True)
    o2, c2 = O.tower_forward(p, x2n, spec, True)
    _, d1, d2, _ = O.pair_loss(o1, o2, y, 'coscos2', 0.5, False)
    og = {}
    _, dx1 = O.tower_backward(p, c1, d1, spec, og, return_dx=True)
    _, dx2 = O.tower_backward(p, c2, d2, spec, og, return_dx=True)
    assert rel_err(x1.grad.cpu().numpy(), dx1) < 1e-5 and rel_err(x2.grad.cpu().numpy(), dx2) < 1e-5
    check_grads({k: q.grad.cpu().numpy() for k, q in net.named_parameters()}, og, spec.param_keys(), False, tol=1e-4)


def test_one_layer_tower_and_a_linear_output(split):
    import abnet3_amd.loss as L
    from oracle import siamese_np as O
    for nh, last in ((0, 'default'), (1, None)):
        kw = dict(input_dim=40, num_hidden_layers=nh, hidden_dim=64, output_dim=32, activation_layer='sigmoid', p_dropout=0.0,
                  last_non_linearity=last)
        net, spec, p = build(kw, seed=3, precision=split)
        B = 70
        rng = np.random.default_rng(3)
        x1, x2 = rng.standard_normal((B, 40)).astype(np.float32), rng.standard_normal((B, 40)).astype(np.float32)
        y = rng.choice([1, -1], B)
        net.train()
        e1, e2 = net(dev(x1), dev(x2))
        L.cosmargin(avg=True)(e1, e2, dev(y)).backward()
        o1, c1 = O.tower_forward(p, x1, spec, True)
        o2, c2 = O.tower_forward(p, x2, spec, True)
        _, d1, d2, _ = O.pair_loss(o1, o2, y, 'cosmargin', 0.5, True)
        og = {}
        O.tower_backward(p, c1, d1, spec, og)
        O.tower_backward(p, c2, d2, spec, og)
        assert rel_err(e1.detach().cpu().numpy(), o1) < 1e-5
        check_grads({k: q.grad.cpu().numpy() for k, q in net.named_parameters()}, og, spec.param_keys(), False, tol=1e-4)


@pytest.mark.parametrize('avg', [False, True])
@pytest.mark.parametrize('n', [150, 97, 320])
def test_a_padded_batch_is_bit_identical_to_the_batch(n, avg, split):
    """n pairs, and the same pairs followed by zero rows up to the next multiple of 32 with the count of real
    pairs in a device word: the same loss and the same gradients, bit for bit (every call is padded to whole
    32-row workgroups anyway, so the two run the same arithmetic in the same order)."""
    kw = dict(input_dim=280, num_hidden_layers=2, hidden_dim=500, output_dim=100, activation_layer='sigmoid', p_dropout=0.0)
    net, _, _ = build(kw, seed=1, precision=split)
    net.train()
    rng = np.random.default_rng(n)
    npad = (n + 31) // 32 * 32
    x1, x2 = [dev(rng.standard_normal((n, 280)).astype(np.float32)) for _ in range(2)]
    y = dev(rng.choice([1.0, -1.0], n))
    emb, st = net.direct_forward(x1, x2)
    assert _lib()[0].last_forward_path() == 6
    loss_a = net.direct_backward_loss(st, y, 'coscos2', 0.0, avg, defer_reduce=False)
    emb_a = emb.clone()
    ga = {k: q.grad.clone() for k, q in net.named_parameters()}
    x12 = torch.zeros(2 * npad, 280, device='cuda')
    x12[:n], x12[npad:npad + n] = x1, x2
    yp = torch.zeros(npad, dtype=torch.float64, device='cuda')
    yp[:n] = y
    yp[n:] = -1.0
    nv = torch.tensor([n], dtype=torch.int32, device='cuda')
    emb, st = net.direct_forward(x12[:npad], x12[npad:])
    loss_b = net.direct_backward_loss(st, yp, 'coscos2', 0.0, avg, defer_reduce=False, n_valid=nv)
    assert torch.equal(loss_a, loss_b)
    assert torch.equal(emb[:n], emb_a[:n]) and torch.equal(emb[npad:npad + n], emb_a[n:])
    for k, q in net.named_parameters():
        assert torch.equal(q.grad, ga[k]), k


def test_dropout_from_the_seed_matches_its_masks(monkeypatch, split):
    """p_dropout > 0 without mask tensors: the multipliers are a hash of (seed, layer, row, feature) in the forward
    epilogues and again in the data-gradient launches.  The masks are recovered from the row-major outputs (a dropped
    tanh unit is an exact zero) and fed back as tensors (which run on the single-launch chains): same embeddings,
    same gradients."""
    from abnet3_amd import _lib as LIB
    lib = LIB.load()
    p_drop = 0.25
    kw = dict(input_dim=40, num_hidden_layers=1, hidden_dim=96, output_dim=32, activation_layer='tanh', p_dropout=p_drop)
    net, _, _ = build(kw, seed=3, precision=split)
    net.train()
    rng = np.random.default_rng(8)
    B = 128
    x1, x2 = dev(rng.standard_normal((B, 40)).astype(np.float32)), dev(rng.standard_normal((B, 40)).astype(np.float32))
    d_out = dev(rng.standard_normal((2 * B, 32)).astype(np.float32))
    seg = net._segment_list()[0]
    emb, state = net.direct_forward(x1, x2)
    assert _lib()[0].last_forward_path() == 6
    sv = state[1]
    assert type(sv.masks).__name__ == '_DropSeed'
    net.direct_backward(state, d_out)
    assert _lib()[0].last_backward_path() == 6
    emb = emb.clone()
    grads = {k: q.grad.clone() for k, q in net.named_parameters()}
    fn = lib.abn_tower_image_offset
    fn.restype = ctypes.c_int64
    desc = seg.descriptor(with_grads=False, masks=sv.masks)
    masks = []
    for l, w in enumerate((96, 96, 32)):
        off = fn(ctypes.byref(desc), ctypes.c_int64(2 * B), ctypes.c_int64(2), 5, l)
        a = sv.ws[off:off + 2 * B * w].view(2 * B, w)            # (B is a multiple of 32: virtual rows = rows)
        frac = float((a == 0).float().mean())
        assert abs(frac - p_drop) < 0.03, (l, frac)
        masks.append((a != 0).float() / (1 - p_drop))
    monkeypatch.setenv('ABN_FUSED_MIN_ROWS', '0')
    net._mask_override = masks
    for q in net.parameters():
        q.grad = None
    emb_t, state_t = net.direct_forward(x1, x2)
    assert _lib()[0].last_forward_path() == 2
    net.direct_backward(state_t, d_out)
    assert rel_err(emb_t.cpu().numpy(), emb.cpu().numpy()) < 2e-6
    for k, q in net.named_parameters():
        assert rel_err(q.grad.cpu().numpy(), grads[k].cpu().numpy(), floor=1e-6 * float(grads[k].abs().max()) + 1e-30) < 1e-4, k


def test_eight_calls_of_ragged_length(split):
    """abn_tower_forward's n_calls (up to 8 forward_once calls in one launch sequence), each padded on its own."""
    from abnet3_amd import _lib as LIB, model as M
    from oracle import siamese_np as O
    kw = dict(input_dim=40, num_hidden_layers=1, hidden_dim=64, output_dim=32, activation_layer='sigmoid', p_dropout=0.0)
    net, spec, p = build(kw, seed=2, precision=split)
    net.train()
    x = np.random.default_rng(0).standard_normal((8 * 19, 40)).astype(np.float32)
    seg = net._segment_list()[0]
    out, sv = M._segment_forward(seg, None, 8, dev(x), None)
    assert LIB.last_forward_path() == 6
    o, _ = O.tower_forward(p, x, spec, True)
    assert rel_err(out.cpu().numpy(), o) < 1e-5


@pytest.mark.parametrize('oname,lr,tol', [('sgd', 0.05, 1e-5), ('adadelta', 0.1, 3e-5), ('adagrad', 0.01, 3e-4), ('RMSprop', 0.001, 3e-4),
                                          ('adam', 0.001, 3e-4)])
def test_the_one_launch_step_applies_every_optimizer_rule(oname, lr, tol, monkeypatch):
    """tower_wgrad_step.h (round 6): every layer's weight gradient over all rows AND the optimizer's rule in one launch.  The
    rule is opt_rule.h's, the one the slab path's launch applies: three steps of each optimizer_type of abnet3/trainer.py:68-87
    end with the parameters of the two-launch path (ABN_WGRAD_STEP=0) to the tolerance the reference's own steps are held to
    (the two sum the same products in another order; Adam / RMSprop / Adagrad divide by sqrt(v) ~ |g|: rounding noise is
    amplified where a gradient is tiny)."""
    import abnet3_amd.loss as L
    from abnet3_amd.trainer import TrainerSiamese
    kw = dict(input_dim=280, num_hidden_layers=2, hidden_dim=500, output_dim=100, activation_layer='sigmoid', p_dropout=0.0, batch_norm=False)
    rng = np.random.default_rng(11)
    B = 333
    batch = (dev(rng.standard_normal((B, 280)).astype(np.float32)), dev(rng.standard_normal((B, 280)).astype(np.float32)),
             dev(rng.choice([1, -1], B)))

    def run(fused):
        monkeypatch.setenv('ABN_WGRAD_STEP', '1' if fused else '0')
        net, spec, p = build(kw, seed=3, precision='f16x2')
        tr = TrainerSiamese(network=net, loss=L.coscos2(avg=False), optimizer_type=oname, lr=lr, dataloader=None, log_dir='/tmp/abn_runs')
        net.train()
        losses = [float(tr.train_step(batch, True)) for _ in range(3)]
        assert _lib()[0].last_backward_path() == 6
        return losses, {k: q.detach().cpu().numpy().copy() for k, q in net.named_parameters()}
    l1, p1 = run(True)
    l0, p0 = run(False)
    assert np.allclose(l1, l0, rtol=2e-5), (l1, l0)
    for k in p0:
        if oname in ('sgd', 'adadelta'):
            assert rel_err(p1[k], p0[k]) < tol, (k, rel_err(p1[k], p0[k]))
        else:
            # these rules' first steps are lr * g / |g|: where a gradient is rounding noise its SIGN differs between two
            # summation orders and the parameter moves lr the other way -- a handful of elements; a wrong rule moves them all
            far = np.abs(p1[k] - p0[k]) > tol * np.abs(p0[k]).max()
            assert far.sum() <= max(3, 2e-3 * far.size), (k, int(far.sum()), far.size)
            assert np.median(np.abs(p1[k] - p0[k])) < 1e-6 * np.abs(p0[k]).max(), k
    assert any(not np.array_equal(p1[k], p0[k]) for k in p0)      # (another summation order: the switch did select another launch)
