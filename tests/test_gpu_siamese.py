"""Parity of the HIP tower + pair loss (through the C-ABI, behind the reference
class surface) with the golden vectors produced by the reference and with the
numpy oracle.  Needs an MI355X: run with -m gpu."""
import ast

import numpy as np
import pytest
import torch

from conftest import (load_golden, rel_err, check_grads, check_params,
                      check_loss_grads)

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=['bf16x3', 'f16x2', 'fp32'])
def tower_precision(request, monkeypatch):
    """Every test of this file runs on both parity-grade arithmetics of the tower GEMMs: the
    default bf16 x 3 split products and the exact-fp32 MFMA (SiameseNetwork.precision)."""
    monkeypatch.setenv('ABNET3_PRECISION', request.param)
    return request.param
TOL = 1e-5       # BASELINE.json: 1e-5 relative fp32 on embeddings and losses


def cuda_net(g, seed=None, prefix='p.'):
    from abnet3_amd.model import SiameseNetwork
    kw = ast.literal_eval(str(g['kw']))
    if seed is not None:
        torch.manual_seed(seed)
    net = SiameseNetwork(**kw)
    if prefix is not None:
        sd = {k[len(prefix):]: torch.from_numpy(v.copy()) for k, v in g.items()
              if k.startswith(prefix)}
        net.load_state_dict(sd)
    return net.cuda(), kw


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_library_is_the_hip_one():
    from abnet3_amd import _lib
    assert _lib.load().abn_abi_version() == _lib.ABI_VERSION
    assert torch.cuda.is_available()


@pytest.mark.parametrize('name', ['sig', 'sig_bn', 'relu_bn', 'tanh', 'sig_lin', 'relu_h2',
                                  'tanh_softmax', 'relu_bn_softmax'])
def test_tower_forward_matches_reference(name, forward_path):
    g = load_golden('tower_%s.npz' % name)
    net, kw = cuda_net(g)
    x1, x2 = dev(g['x1']), dev(g['x2'])
    net.eval()
    with torch.no_grad():
        e1, e2 = net(x1, x2)
        o1 = net.forward_once(x1)
    assert rel_err(e1.cpu().numpy(), g['eval_e1']) < TOL
    assert rel_err(e2.cpu().numpy(), g['eval_e2']) < TOL
    assert rel_err(o1.cpu().numpy(), g['eval_e1']) < TOL
    net.train()
    with torch.no_grad():
        e1, e2 = net(x1, x2)
    assert rel_err(e1.cpu().numpy(), g['train_e1']) < TOL
    assert rel_err(e2.cpu().numpy(), g['train_e2']) < TOL
    sd = net.state_dict()
    for k, v in g.items():
        if k.startswith('after.') and 'running' in k:
            assert rel_err(sd[k[6:]].cpu().numpy(), v) < TOL, k
        if k.startswith('after.') and 'num_batches_tracked' in k:
            assert int(sd[k[6:]]) == int(v) == 2


@pytest.mark.parametrize('name', ['tanh_softmax', 'relu_bn_softmax'])
def test_softmax_head_gradients(name):
    """last_non_linearity='softmax' (model.py:161-166): abn_softmax_rows after the
    linear-ended tower, and its backward, against the reference's autograd."""
    import abnet3_amd.loss as L
    g = load_golden('tower_%s.npz' % name)
    net, kw = cuda_net(g)
    net.train()
    e1, e2 = net(dev(g['x1']), dev(g['x2']))
    assert np.allclose(e1.detach().sum(dim=1).cpu().numpy(), 1.0, atol=1e-5)
    lv = L.cosmargin(avg=True)(e1, e2, dev(g['y']))
    lv.backward()
    ref = float(g['loss'])
    assert abs(float(lv.detach()) - ref) <= 1e-5 * abs(ref)
    keys = [k for k, _ in net.named_parameters()]
    grads = {k: p.grad.cpu().numpy() for k, p in net.named_parameters()}
    check_grads(grads, {k: g['grad.' + k] for k in keys}, keys, bool(kw['batch_norm']), tol=5e-5)
    assert net.grads_in_flat_buffer()


OPT = {'sgd': lambda p: torch.optim.SGD(p, lr=0.001, momentum=0.9),
       'adadelta': lambda p: torch.optim.Adadelta(p, lr=0.1),
       'adam': lambda p: torch.optim.Adam(p, lr=0.001),
       'adagrad': lambda p: torch.optim.Adagrad(p, lr=0.001),
       'RMSprop': lambda p: torch.optim.RMSprop(p, lr=0.001)}

CASES_C1 = [(l, a, o) for l in ('coscos2', 'cosmargin') for a in (1, 0)
            for o in ('sgd', 'adadelta')] + \
           [('coscos2', 0, o) for o in ('adam', 'adagrad', 'RMSprop')]


@pytest.mark.parametrize('bn', [0, 1])
@pytest.mark.parametrize('lname,avg,oname', CASES_C1)
def test_c1_grads_and_three_steps(bn, lname, avg, oname, forward_path):
    """C1 = 40->100->50, B=32 (BASELINE.json configs[0]); the statements of
    trainer.py:236-242 with torch.optim driving the HIP network's parameters."""
    import abnet3_amd.loss as L
    g = load_golden('train_c1_bn%d.npz' % bn)
    net, kw = cuda_net(g)
    loss_mod = getattr(L, lname)(avg=bool(avg)).cuda()
    opt = OPT[oname](net.parameters())
    x1, x2, y = dev(g['x1']), dev(g['x2']), dev(g['y'])
    tag = '%s.avg%d.%s' % (lname, avg, oname)
    keys = [k for k, _ in net.named_parameters()]
    net.train()
    losses = []
    for s in range(3):
        e1, e2 = net(x1, x2)
        lv = loss_mod(e1, e2, y)
        opt.zero_grad()
        lv.backward()
        if s == 0:
            assert rel_err(e1.detach().cpu().numpy(), g[tag + '.e1_0']) < TOL
            grads = {k: p.grad.cpu().numpy() for k, p in net.named_parameters()}
            check_grads(grads, {k: g['%s.grad0.%s' % (tag, k)] for k in keys}, keys, bool(bn))
            assert net.grads_in_flat_buffer()      # zero-copy into the flat bucket
        opt.step()
        losses.append(float(lv.detach()))
    assert np.allclose(losses, g[tag + '.losses'], rtol=1e-5, atol=1e-6)
    params = {k: p.detach().cpu().numpy() for k, p in net.named_parameters()}
    check_params(params, {k: g['%s.after.%s' % (tag, k)] for k in keys}, keys, bool(bn),
                 1e-5 if oname in ('sgd', 'adadelta') else 3e-4)


LOSS_TAGS = [(ls, ln, m, a) for ls in ('mixed', 'allsame', 'alldiff', 'f64')
             for (ln, m) in (('coscos2', None), ('cosmargin', None), ('cosmargin', 0.2))
             for a in (1, 0)]


@pytest.mark.parametrize('ls,lname,margin,avg', LOSS_TAGS)
def test_loss_edge_cases(ls, lname, margin, avg):
    import abnet3_amd.loss as L
    g = load_golden('loss_edge.npz')
    tag = '%s.%s%s.avg%d' % (ls, lname, '' if margin is None else '_m%g' % margin, avg)
    kwargs = {} if margin is None else {'margin': margin}
    mod = getattr(L, lname)(avg=bool(avg), **kwargs)
    e1 = dev(g['e1']).requires_grad_(True)
    e2 = dev(g['e2']).requires_grad_(True)
    lv = mod(e1, e2, dev(g['y.' + ls]))
    assert lv.dim() == 0
    lv.backward()
    ref = float(g[tag + '.loss'])
    assert abs(float(lv.detach()) - ref) <= 1e-5 * abs(ref) + 1e-6
    check_loss_grads(e1.grad.cpu().numpy(), e2.grad.cpu().numpy(),
                     g[tag + '.de1'], g[tag + '.de2'], lname, tag)


@pytest.mark.parametrize('dtype', [torch.int8, torch.int32, torch.int64, torch.float32, torch.float64])
def test_loss_label_dtypes_and_odd_width(dtype):
    """labels of any dtype (loss.py:60-63) and an embedding width that forces
    the scalar (non 16-byte) path."""
    import abnet3_amd.loss as L
    from oracle import siamese_np as O
    rng = np.random.default_rng(11)
    for D in (50, 37, 100, 3):
        e1 = rng.standard_normal((19, D)).astype(np.float32)
        e2 = rng.standard_normal((19, D)).astype(np.float32)
        y = rng.choice([1, -1, 0], 19)
        for kind in ('coscos2', 'cosmargin'):
            a = dev(e1).requires_grad_(True)
            b = dev(e2).requires_grad_(True)
            lv = getattr(L, kind)(avg=False)(a, b, torch.from_numpy(y).to(dtype).cuda())
            lv.backward()
            ol, o1, o2, _ = O.pair_loss(e1, e2, y, kind, 0.5, False)
            assert abs(float(lv.detach()) - ol) <= 1e-5 * abs(ol) + 1e-6
            assert rel_err(a.grad.cpu().numpy(), o1) < TOL
            assert rel_err(b.grad.cpu().numpy(), o2) < TOL


@pytest.mark.parametrize('bn', [0, 1])
def test_mid_width_against_oracle_and_reference(bn):
    """40->72x2->36, B=96: odd widths exercise every tile tail."""
    import abnet3_amd.loss as L
    from oracle import siamese_np as O
    g = load_golden('train_mid_bn%d.npz' % bn)
    net, kw = cuda_net(g)
    spec = O.TowerSpec(kw['input_dim'], kw['num_hidden_layers'], kw['hidden_dim'],
                       kw['output_dim'], kw['activation_layer'], kw['batch_norm'])
    p = {k[2:]: v.copy() for k, v in g.items() if k.startswith('p.')}
    keys = spec.param_keys()
    net.train()
    e1, e2 = net(dev(g['x1.0']), dev(g['x2.0']))
    lv = L.coscos2(avg=False)(e1, e2, dev(g['y.0']))
    lv.backward()
    ol, og, (oe1, oe2) = O.train_step(p, g['x1.0'], g['x2.0'], g['y.0'], spec,
                                      O.Optimizer('sgd', 0.0), avg=False)
    assert rel_err(e1.detach().cpu().numpy(), oe1) < TOL
    assert abs(float(lv.detach()) - ol) <= 1e-5 * abs(ol)
    grads = {k: q.grad.cpu().numpy() for k, q in net.named_parameters()}
    # tight against the oracle (fp64 loss gradient on both sides) ...
    check_grads(grads, og, keys, bool(bn), tol=5e-5)
    # ... and within the reference's own fp32 noise against the golden
    # (test_oracle_siamese.test_mid_five_steps explains the 1e-3)
    tag = 'coscos2.avg0.sgd'
    check_grads(grads, {k: g['%s.grad0.%s' % (tag, k)] for k in keys}, keys, bool(bn), tol=1e-3)


def check_all_embedding_rows(g, e1, e2, tol=1e-5):
    """Every one of the 2 x 4096 embedding rows of the first C2 forward against the reference's: the fixture holds
    each row's sum and sum |.| (float64) and each tensor's sum, sum |.|, max |.|; 1e-5 of the row's (tensor's) mass."""
    for name, e in (('e1', e1), ('e2', e2)):
        ed = e.detach().double().cpu()
        rowsum, rowabs = ed.sum(dim=1).numpy(), ed.abs().sum(dim=1).numpy()
        ref_sum, ref_abs = g[name + '_rowsum'], g[name + '_rowabs']
        assert rowsum.shape == ref_sum.shape == (4096,)
        assert np.all(np.abs(rowsum - ref_sum) <= tol * ref_abs), (name, np.abs(rowsum - ref_sum).max())
        assert np.all(np.abs(rowabs - ref_abs) <= tol * ref_abs), (name, np.abs(rowabs - ref_abs).max())
        tot = [float(ed.sum()), float(ed.abs().sum()), float(ed.abs().max())]
        assert abs(tot[0] - g[name + '_chk'][0]) <= tol * g[name + '_chk'][1]
        assert abs(tot[1] - g[name + '_chk'][1]) <= tol * g[name + '_chk'][1]
        assert abs(tot[2] - g[name + '_chk'][2]) <= tol * g[name + '_chk'][2]


@pytest.mark.parametrize('bn', [0, 1])
def test_c2_five_adadelta_steps(bn):
    """BASELINE.json configs[1]: 40->500x2->100, coscos2, B=4096, weights from
    torch.manual_seed(2) (same RNG consumption as the reference constructor)."""
    import abnet3_amd.loss as L
    from oracle import torch_ref
    g = load_golden('train_c2_bn%d.npz' % bn)
    net, kw = cuda_net(g, seed=2, prefix=None)
    for k, v in net.state_dict().items():
        v = v.double().cpu()
        assert np.allclose([float(v.sum()), float(v.abs().sum())], g['chk.' + k], rtol=1e-9), k
    opt = torch.optim.Adadelta(net.parameters(), lr=0.1)
    loss_mod = L.coscos2(avg=False)
    net.train()
    losses = []
    for s in range(5):
        x1, x2, y = torch_ref.make_inputs(4096, 40, 20 + s % 2)
        e1, e2 = net(x1.cuda(), x2.cuda())
        lv = loss_mod(e1, e2, torch.from_numpy(y).cuda())
        opt.zero_grad()
        lv.backward()
        if s == 0:
            from abnet3_amd import _lib
            # (this fixture is where the resident BatchNorm tower meets the reference's own numbers: it must be the path that ran)
            assert not bn or net.precision == 'fp32' or _lib.last_forward_path() == _lib.PATH_BN_TOWER, (net.precision, _lib.last_forward_path())
            assert rel_err(e1.detach().cpu().numpy()[:8], g['e1_rows']) < TOL
            assert rel_err(e2.detach().cpu().numpy()[-8:], g['e2_rows']) < TOL
            check_all_embedding_rows(g, e1, e2)
            for k, q in net.named_parameters():
                gg = q.grad.cpu().numpy().reshape(q.shape[0], -1)
                if bn and k.endswith('bias') and 'running' not in k and (
                        k.endswith('.0.bias') or int(k.split('.')[1]) % 4 == 0):
                    continue          # pre-BN bias: rounding noise on both sides
                assert rel_err(gg[:4], g['grow.' + k], 1e-6) < 3e-4, k
        opt.step()
        losses.append(float(lv.detach()))
    # the first step's loss is a function of the initial weights alone: north_star's 1e-5; later steps also carry
    # four rounds of Adadelta on gradients that are only 3e-4-reproducible at this initialisation (see above)
    assert abs(losses[0] - g['losses'][0]) <= 1e-5 * abs(g['losses'][0]), (losses[0], g['losses'][0])
    assert np.allclose(losses, g['losses'], rtol=2e-5)
    for k, v in net.state_dict().items():
        if 'num_batches' in k or (bn and k.endswith('bias')) or 'running_mean' in k:
            continue
        v = v.double().cpu()
        assert np.allclose([float(v.sum()), float(v.abs().sum())], g['after_chk.' + k],
                           rtol=1e-4, atol=1e-3), k


def test_second_backward_before_zero_grad_accumulates():
    """two forward_once calls + two backwards must SUM gradients (autograd
    semantics of the reference's two tower calls), not alias the flat buffer."""
    import abnet3_amd.loss as L
    g = load_golden('train_c1_bn0.npz')
    net, _ = cuda_net(g)
    net.train()
    x1, x2, y = dev(g['x1']), dev(g['x2']), dev(g['y'])
    e1, e2 = net(x1, x2)
    L.coscos2(avg=False)(e1, e2, y).backward()
    fused = [p.grad.clone() for p in net.parameters()]
    net.zero_grad()
    a = net.forward_once(x1)
    b = net.forward_once(x2)
    L.coscos2(avg=False)(a, b, y).backward()
    for f, p in zip(fused, net.parameters()):
        assert rel_err(p.grad.cpu().numpy(), f.cpu().numpy()) < 1e-5


def test_cpu_tensors_fail_loudly():
    from abnet3_amd.model import SiameseNetwork
    from abnet3_amd._lib import HipLibraryError
    net = SiameseNetwork(input_dim=10, num_hidden_layers=1, hidden_dim=10, output_dim=5,
                         p_dropout=0., activation_layer='relu')
    with pytest.raises(HipLibraryError):
        net(torch.randn(4, 10), torch.randn(4, 10))


@pytest.mark.parametrize('bn', [False, True])
def test_dropout_masks_forward_backward_vs_oracle(bn, forward_path):
    """nn.Dropout(p) sits between Linear and BatchNorm/activation
    (model.py:137,148,157).  torch's CPU RNG stream cannot be reproduced on the
    device, so the ARITHMETIC is pinned with masks shared by oracle and kernel,
    and the mask generator is checked statistically."""
    import abnet3_amd.loss as L
    from abnet3_amd.model import SiameseNetwork
    from oracle import siamese_np as O
    torch.manual_seed(5)
    kw = dict(input_dim=40, num_hidden_layers=1, hidden_dim=72, output_dim=36,
              activation_layer='relu', batch_norm=bn, p_dropout=0.25)
    net = SiameseNetwork(**kw).cuda()
    spec = O.TowerSpec(40, 1, 72, 36, 'relu', bn)
    p = {k: v.detach().cpu().numpy().copy() for k, v in net.state_dict().items()}
    rng = np.random.default_rng(0)
    B = 48
    x1 = rng.standard_normal((B, 40)).astype(np.float32)
    x2 = rng.standard_normal((B, 40)).astype(np.float32)
    y = rng.choice([1, -1], B)
    masks = [((rng.random((2 * B, w)) >= 0.25) / 0.75).astype(np.float32) for w in (72, 72, 36)]
    net._mask_override = [dev(m) for m in masks]
    net.train()
    e1, e2 = net(dev(x1), dev(x2))
    lv = L.coscos2(avg=False)(e1, e2, dev(y))
    lv.backward()
    o1, c1 = O.tower_forward(p, x1, spec, True, masks=[m[:B] for m in masks])
    o2, c2 = O.tower_forward(p, x2, spec, True, masks=[m[B:] for m in masks])
    ol, d1, d2, _ = O.pair_loss(o1, o2, y, 'coscos2', 0.5, False)
    og = {}
    O.tower_backward(p, c1, d1, spec, og)
    O.tower_backward(p, c2, d2, spec, og)
    assert rel_err(e1.detach().cpu().numpy(), o1) < TOL
    assert abs(float(lv.detach()) - ol) <= 1e-5 * abs(ol)
    grads = {k: q.grad.cpu().numpy() for k, q in net.named_parameters()}
    # with dropout between Linear and BN the pre-BN bias gradient is NOT zero
    # any more (the mask breaks the mean invariance): compare it like the rest
    check_grads(grads, og, spec.param_keys(), False, tol=5e-5)
    # eval mode ignores dropout entirely
    net.eval()
    with torch.no_grad():
        ev = net.forward_once(dev(x1))
    oe, _ = O.tower_forward(p, x1, spec, False)
    assert rel_err(ev.cpu().numpy(), oe) < TOL
    # the generator: zeros with probability p, survivors scaled by 1/(1-p)
    net._mask_override = None
    drawn = net._draw_mask_tensors(4096, torch.device('cuda'))          # (the tensor form: per-layer path)
    for m in drawn:
        frac = float((m == 0).float().mean())
        assert abs(frac - 0.25) < 0.01
        assert torch.all((m == 0) | ((m - 1 / 0.75).abs() < 1e-6))


def test_reference_style_update_all_weights_with_default_dropout():
    """test/test_model.py:43-96 re-enacted on the device (p_dropout=0.1 variant
    included): after one step every parameter tensor has changed."""
    import copy
    import abnet3_amd.loss as L
    from abnet3_amd.model import SiameseNetwork
    models = {
        'siamese_relu': dict(input_dim=10, num_hidden_layers=2, hidden_dim=10, output_dim=40,
                             p_dropout=0.1, activation_layer='relu'),
        'siamese_sig': dict(input_dim=10, num_hidden_layers=4, hidden_dim=10, type_init='orthogonal',
                            output_dim=15, p_dropout=0., activation_layer='sigmoid'),
        'siamese_batch': dict(input_dim=10, num_hidden_layers=4, hidden_dim=10, type_init='orthogonal',
                              output_dim=15, p_dropout=0., activation_layer='relu', batch_norm=True)}
    for name, kw in models.items():
        for loss_cls in (L.coscos2, L.cosmargin):
            torch.manual_seed(1)
            net = SiameseNetwork(**kw).cuda()
            x1, x2 = torch.randn(128, 10).cuda(), torch.randn(128, 10).cuda()
            y = torch.from_numpy(np.random.choice([-1, 1], 128)).cuda()
            before = copy.deepcopy([p_.detach().clone() for p_ in net.parameters()])
            net.train()
            opt = torch.optim.Adam(net.parameters(), lr=0.0005)
            o1, o2 = net(x1, x2)
            opt.zero_grad()
            loss_cls(avg=False)(o1, o2, y).backward()
            opt.step()
            for a, b in zip(before, net.parameters()):
                assert (a != b).any(), name


def test_graphed_step_equals_eager_step():
    """make_graphed_step (hipGraph replay, tuple batches and pack_batch blobs) takes
    the same steps as the eager TrainerSiamese.train_step, which the golden tests pin."""
    import abnet3_amd.loss as L
    from abnet3_amd.trainer import TrainerSiamese
    g = load_golden('train_mid_bn0.npz')
    rng = np.random.default_rng(3)
    batches = [(dev(rng.standard_normal((96, 40)).astype(np.float32)),
                dev(rng.standard_normal((96, 40)).astype(np.float32)),
                dev(rng.choice([1.0, -1.0], 96))) for _ in range(3)]
    results = []
    for mode in ('eager', 'graph', 'graph_packed'):
        net, _ = cuda_net(g)
        net.output_path = '/tmp/abn_graph_test'
        tr = TrainerSiamese(network=net, loss=L.coscos2(avg=False), optimizer_type='adadelta',
                            lr=0.1, dataloader=None, log_dir='/tmp/abn_runs')
        net.train()
        losses = []
        if mode == 'eager':
            for s in range(2):                      # the warm-up steps of the graphed variants
                tr.train_step(batches[0], True)
            for s in range(6):
                losses.append(float(tr.train_step(batches[s % 3], True)))
        else:
            step = tr.make_graphed_step(batches[0], warmup=2)
            feed = [tr.pack_batch(b) for b in batches] if mode == 'graph_packed' else batches
            for s in range(6):
                losses.append(float(step(feed[s % 3])))
        results.append((losses, {k: p.detach().cpu().numpy().copy() for k, p in net.named_parameters()}))
    for losses, params in results[1:]:
        assert np.allclose(losses, results[0][0], rtol=1e-6, atol=0)
        for k, v in params.items():           # (the flat buffer's alignment gaps hold no parameters)
            # (fp16 x 2, a small batch: the eager step sums its weight gradients over all rows in the optimizer's launch --
            # tower_wgrad_step.h --, the captured autograd backward in slabs: the same products in another order, eight steps on)
            assert rel_err(v, results[0][1][k]) < 5e-6, k


@pytest.mark.parametrize('fixture,bn', [('train_c2_bn0.npz', False), ('train_mid_bn1.npz', True)])
def test_bf16_throughput_mode_tracks_fp32(fixture, bn):
    """precision='bf16' (opt-in, NOT the parity path): operands of every tower GEMM are
    rounded to bf16, accumulation stays fp32.  It must stay within bf16's ~3 significant
    digits of the fp32 path in the forward, and train to the same loss curve."""
    import copy
    import abnet3_amd.loss as L
    g = load_golden(fixture)
    net32, kw = cuda_net(g, seed=2 if 'c2' in fixture else None, prefix=None if 'c2' in fixture else 'p.')
    net16 = copy.deepcopy(net32)
    net16.precision = 'bf16'
    net32.precision = 'fp32'
    rng = np.random.default_rng(1)
    B = 256
    x1 = dev(rng.standard_normal((B, kw['input_dim'])).astype(np.float32))
    x2 = dev(rng.standard_normal((B, kw['input_dim'])).astype(np.float32))
    y = dev(rng.choice([1, -1], B))
    for net in (net32, net16):
        net.eval()
    with torch.no_grad():
        a, b = net32.forward_once(x1), net16.forward_once(x1)
    err = rel_err(b.cpu().numpy(), a.cpu().numpy())
    assert 1e-6 < err < 3e-2, err                  # different arithmetic, same function
    curves = []
    for net in (net32, net16):
        net.train()
        opt = torch.optim.Adadelta(net.parameters(), lr=0.1)
        loss_mod = L.coscos2(avg=True)
        losses = []
        for s in range(8):
            e1, e2 = net(x1, x2)
            lv = loss_mod(e1, e2, y)
            opt.zero_grad()
            lv.backward()
            opt.step()
            losses.append(float(lv.detach()))
        curves.append(losses)
    assert np.allclose(curves[0], curves[1], rtol=3e-2), curves
    assert curves[1][-1] <= curves[1][0]           # (a sigmoid tower at its initialisation barely moves)


@pytest.mark.parametrize('planes', ['1', '0'])
@pytest.mark.parametrize('fixture,bn,B', [('train_c2_bn0.npz', False, 4096), ('train_mid_bn1.npz', True, 96),
                                          ('train_c2_bn1.npz', True, 1000)])
def test_bf16x3_precision_is_fp32_grade(fixture, bn, B, planes, monkeypatch):
    """precision='bf16x3': every tower GEMM sums six bf16 products per operand pair (operands
    split into hi + mid + lo bf16 terms; ABN_BF16X3_PLANES picks where the split happens).
    Embeddings, loss and every parameter gradient must agree with the exact-fp32 MFMA path
    to a few 1e-6 -- the level at which fp32 itself sits from a float64 evaluation."""
    import copy
    import abnet3_amd.loss as L
    from abnet3_amd.trainer import TrainerSiamese
    monkeypatch.setenv('ABN_BF16X3_PLANES', planes)
    g = load_golden(fixture)
    net32, kw = cuda_net(g, seed=2 if 'c2' in fixture else None, prefix=None if 'c2' in fixture else 'p.')
    net3 = copy.deepcopy(net32)
    net3.precision = 'bf16x3'
    net32.precision = 'fp32'
    rng = np.random.default_rng(3)
    x1 = dev(rng.standard_normal((B, kw['input_dim'])).astype(np.float32))
    x2 = dev(rng.standard_normal((B, kw['input_dim'])).astype(np.float32))
    y = dev(rng.choice([1, -1], B))
    out = []
    for net in (net32, net3):
        net.output_path = '/tmp/abn_x3'
        tr = TrainerSiamese(network=net, loss=L.coscos2(avg=False), optimizer_type='sgd', lr=0.0,
                            dataloader=None, log_dir='/tmp/abn_runs')
        net.eval()
        with torch.no_grad():
            e = net.forward_once(x1).cpu().numpy()
        net.train()
        loss = float(tr.train_step((x1, x2, y), True))
        out.append((e, loss, {k: p.grad.cpu().numpy().copy() for k, p in net.named_parameters()}))
    assert rel_err(out[1][0], out[0][0]) < 3e-6
    assert abs(out[1][1] - out[0][1]) <= 3e-6 * abs(out[0][1])
    gmax = max(np.abs(v).max() for v in out[0][2].values())
    from conftest import is_pre_bn_bias
    for k, v in out[1][2].items():
        if is_pre_bn_bias(k, bn):          # mathematically zero: rounding noise on both sides
            continue
        assert rel_err(v, out[0][2][k], floor=1e-2 * gmax) < 2e-5, (k, rel_err(v, out[0][2][k], floor=1e-2 * gmax))


class _ListLoader(object):
    """Minimal dataloader contract the trainer consumes (dataloader.py:263-312)."""

    def __init__(self, train, dev_batches):
        self.train, self.dev = train, dev_batches

    def batch_iterator(self, train_mode=True):
        return iter(self.train if train_mode else self.dev)

    def whoami(self):
        return {'class_name': 'ListLoader'}


def test_epoch_loop_auto_graph_equals_eager():
    """optimize_model serves recurring batch shapes from a captured hipGraph
    (train_step_auto): same losses and parameters as the all-eager loop, odd-shaped
    batches in between included."""
    import abnet3_amd.loss as L
    from abnet3_amd.trainer import TrainerSiamese
    g = load_golden('train_mid_bn1.npz')
    rng = np.random.default_rng(5)

    def batch(n):
        return (dev(rng.standard_normal((n, 40)).astype(np.float32)),
                dev(rng.standard_normal((n, 40)).astype(np.float32)), dev(rng.choice([1.0, -1.0], n)))
    train = [batch(64) for _ in range(4)] + [batch(37)] + [batch(64) for _ in range(5)] + [batch(21)]
    devb = [batch(64), batch(30)]
    out = []
    for graph_steps in (False, True):
        net, _ = cuda_net(g)
        net.output_path = '/tmp/abn_auto_graph'
        tr = TrainerSiamese(network=net, loss=L.cosmargin(avg=True), optimizer_type='sgd', lr=0.01,
                            momentum=0.9, dataloader=_ListLoader(train, devb), log_dir='/tmp/abn_runs')
        tr.graph_steps = graph_steps
        tr.train_losses, tr.dev_losses = [], []
        for _ in range(2):
            tr.optimize_model(do_training=True)
        assert (len(getattr(tr, '_graphs', {})) == 1) == graph_steps
        out.append((tr.train_losses, tr.dev_losses,
                    {k: p.detach().cpu().numpy().copy() for k, p in net.state_dict().items()}))
    assert np.allclose(out[0][0], out[1][0], rtol=1e-6) and np.allclose(out[0][1], out[1][1], rtol=1e-6)
    for k, v in out[1][2].items():
        assert rel_err(v, out[0][2][k]) < 1e-6, k          # BatchNorm running stats and counters included


def test_auto_graph_redraws_dropout_masks():
    """Dropout masks are drawn by torch's device generator inside the captured step:
    every replay must see fresh masks."""
    import abnet3_amd.loss as L
    from abnet3_amd.model import SiameseNetwork
    from abnet3_amd.trainer import TrainerSiamese
    torch.manual_seed(0)
    net = SiameseNetwork(input_dim=40, num_hidden_layers=1, hidden_dim=64, output_dim=32,
                         p_dropout=0.3, activation_layer='tanh', output_path='/tmp/abn_drop').cuda()
    tr = TrainerSiamese(network=net, loss=L.coscos2(avg=True), optimizer_type='sgd', lr=0.0,
                        dataloader=None, log_dir='/tmp/abn_runs')
    rng = np.random.default_rng(0)
    b = (dev(rng.standard_normal((128, 40)).astype(np.float32)),
         dev(rng.standard_normal((128, 40)).astype(np.float32)), dev(rng.choice([1.0, -1.0], 128)))
    net.train()
    tr.graph_steps = True
    losses = [float(tr.train_step_auto(b)) for _ in range(8)]      # lr = 0: only the masks change
    assert len(tr._graphs) == 1
    assert len(set(losses[3:])) == len(losses[3:]), losses
    assert all(np.isfinite(losses))


@pytest.mark.parametrize('fixture,opt', [('train_mid_bn0.npz', 'adadelta'), ('train_mid_bn1.npz', 'sgd'),
                                         ('tower_sig_lin.npz', 'adam')])
def test_direct_step_equals_autograd_step(fixture, opt):
    """TrainerSiamese.train_step drives forward / fused loss+gradient / backward itself
    (no autograd graph); with direct_steps = False it goes through torch.autograd like
    the reference's five statements.  Same gradients, same parameters, same p.grad
    layout -- variable batch sizes, BatchNorm, linear output head, cosmargin included."""
    import abnet3_amd.loss as L
    from abnet3_amd.trainer import TrainerSiamese
    g = load_golden(fixture)
    rng = np.random.default_rng(9)
    batches = [(dev(rng.standard_normal((n, 40)).astype(np.float32)),
                dev(rng.standard_normal((n, 40)).astype(np.float32)),
                dev(rng.choice([1.0, -1.0], n))) for n in (96, 33, 64, 7, 130)]
    out = []
    for direct in (False, True):
        net, _ = cuda_net(g)
        net.output_path = '/tmp/abn_direct'
        tr = TrainerSiamese(network=net, loss=L.cosmargin(avg=True, margin=0.3), optimizer_type=opt, lr=0.01,
                            dataloader=None, log_dir='/tmp/abn_runs')
        tr.direct_steps = direct
        assert tr._direct_ok() == direct
        net.train()
        losses = [float(tr.train_step(b, True)) for b in batches]
        assert net.grads_in_flat_buffer()
        out.append((losses, {k: p.detach().cpu().numpy().copy() for k, p in net.state_dict().items()},
                    {k: p.grad.cpu().numpy().copy() for k, p in net.named_parameters()}))
    assert np.allclose(out[0][0], out[1][0], rtol=1e-6, atol=0)
    for k, v in out[1][1].items():
        # (fp16 x 2, small batches: the direct step sums its weight gradients over all rows inside the optimizer's launch,
        # tower_wgrad_step.h; the autograd step in slabs: the same products, another order -- and Adadelta's step is the
        # gradient over its own running size: a bias that starts at zero carries the gradients' 1e-5 one to one)
        assert rel_err(v, out[0][1][k]) < 2e-5, k
    for k, v in out[1][2].items():
        assert rel_err(v, out[0][2][k], floor=1e-12) < 3e-5, k      # (the last step's gradients: see tests/test_gpu_wide.py on the two launches)


@pytest.mark.parametrize('rows,k,n,act', [(8192, 500, 500, 'sigmoid'), (200, 40, 72, 'tanh'), (1000, 500, 100, 'none')])
def test_linear_backward_entry_matches_the_two_single_gemm_entries(rows, k, n, act):
    """abn_linear_backward (wgrad + dgrad of one Linear as ONE grid, the way the tower
    backward issues them) against abn_linear_wgrad + abn_linear_dgrad, and those against
    float64 matmuls."""
    from abnet3_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(rows)
    dz = dev(rng.standard_normal((rows, n)).astype(np.float32))
    W = dev((rng.standard_normal((n, k)) * 0.05).astype(np.float32))
    a = dev(rng.uniform(0.05, 0.95, (rows, k)).astype(np.float32))
    code = _lib.ACT[act]
    sc_n = lib.abn_linear_wgrad_scratch_floats(rows, k, n)
    sc = torch.empty(sc_n, device='cuda')
    dW1, db1, dx1 = torch.empty(n, k, device='cuda'), torch.empty(n, device='cuda'), torch.empty(rows, k, device='cuda')
    dW2, db2, dx2 = torch.empty_like(dW1), torch.empty_like(db1), torch.empty_like(dx1)
    _lib.check(lib.abn_linear_backward(_lib.ptr(dz), _lib.ptr(W), _lib.ptr(a), rows, k, n, code, _lib.ptr(dW1),
                                       _lib.ptr(db1), _lib.ptr(dx1), _lib.ptr(sc), sc_n, _lib.stream()), 'bwd')
    _lib.check(lib.abn_linear_wgrad(_lib.ptr(dz), _lib.ptr(a), rows, k, n, _lib.ptr(dW2), _lib.ptr(db2),
                                    _lib.ptr(sc), sc_n, _lib.stream()), 'w')
    _lib.check(lib.abn_linear_dgrad(_lib.ptr(dz), _lib.ptr(W), rows, k, n, _lib.ptr(a) if code else None, code,
                                    _lib.ptr(dx2), _lib.stream()), 'd')
    assert torch.equal(dW1, dW2) and torch.equal(db1, db2) and torch.equal(dx1, dx2)   # same kernels bodies
    dz64, a64, W64 = dz.double().cpu().numpy(), a.double().cpu().numpy(), W.double().cpu().numpy()
    grad = {'sigmoid': a64 * (1 - a64), 'tanh': 1 - a64 * a64, 'none': np.ones_like(a64)}[act]
    assert rel_err(dW1.cpu().numpy(), dz64.T @ a64) < 1e-5
    assert rel_err(db1.cpu().numpy(), dz64.sum(0)) < 1e-5
    assert rel_err(dx1.cpu().numpy(), (dz64 @ W64) * grad) < 1e-5
