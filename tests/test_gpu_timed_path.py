"""The path bench.py times -- TrainerSiamese.train_step (autograd-free forward / fused
loss+gradient / backward / abn_optimizer_step) and make_graphed_step -- against the
numbers the REFERENCE produced (tests/golden, tools/make_golden.py): every
optimizer_type of abnet3/trainer.py:68-87 on C1, and the C2 configuration of
BASELINE.json (five Adadelta steps, full-tensor gradient checksums).
Needs an MI355X: run with -m gpu."""
import ast

import numpy as np
import pytest
import torch

from conftest import is_pre_bn_bias, load_golden, rel_err, check_grads, check_params

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=['bf16x3', 'f16x2', 'fp32'])
def tower_precision(request, monkeypatch):
    """Every test of this file runs on both parity-grade arithmetics of the tower GEMMs: the
    default bf16 x 3 split products and the exact-fp32 MFMA (SiameseNetwork.precision)."""
    monkeypatch.setenv('ABNET3_PRECISION', request.param)
    return request.param
TOL = 1e-5


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def cuda_net(g, seed=None, prefix='p.'):
    from abnet3_amd.model import SiameseNetwork
    kw = ast.literal_eval(str(g['kw']))
    if seed is not None:
        torch.manual_seed(seed)
    net = SiameseNetwork(output_path='/tmp/abn_timed_path', **kw)
    if prefix is not None:
        net.load_state_dict({k[len(prefix):]: torch.from_numpy(v.copy()) for k, v in g.items()
                             if k.startswith(prefix)})
    return net.cuda(), kw


def trainer(net, lname, avg, oname):
    import abnet3_amd.loss as L
    from abnet3_amd.trainer import TrainerSiamese, FlatOptimizer
    lr = {'sgd': 0.001, 'adadelta': 0.1, 'adam': 0.001, 'adagrad': 0.001, 'RMSprop': 0.001}[oname]
    tr = TrainerSiamese(network=net, loss=getattr(L, lname)(avg=bool(avg)), optimizer_type=oname, lr=lr,
                        momentum=0.9, dataloader=None, log_dir='/tmp/abn_runs')
    assert isinstance(tr.optimizer, FlatOptimizer) and tr._direct_ok()
    return tr


CASES_C1 = [(l, a, o) for l in ('coscos2', 'cosmargin') for a in (1, 0)
            for o in ('sgd', 'adadelta')] + \
           [('coscos2', 0, o) for o in ('adam', 'adagrad', 'RMSprop')]


@pytest.mark.parametrize('bn', [0, 1])
@pytest.mark.parametrize('lname,avg,oname', CASES_C1)
@pytest.mark.parametrize('mode', ['eager', 'graph'])
def test_c1_train_step_all_optimizers_vs_reference(bn, lname, avg, oname, mode):
    """G2 (C1 = 40->100->50, B=32): three steps of the direct path + abn_optimizer_step
    (sgd, adadelta, adam, adagrad, RMSprop) -- eager and replayed from a hipGraph --
    against the reference's losses, first-step gradients and parameters after 3 steps."""
    g = load_golden('train_c1_bn%d.npz' % bn)
    net, kw = cuda_net(g)
    tr = trainer(net, lname, avg, oname)
    batch = (dev(g['x1']), dev(g['x2']), dev(g['y']))
    tag = '%s.avg%d.%s' % (lname, avg, oname)
    keys = [k for k, _ in net.named_parameters()]
    net.train()
    if mode == 'eager':
        losses = [float(tr.train_step(batch, True))]
        grads = {k: p.grad.cpu().numpy().copy() for k, p in net.named_parameters()}
        check_grads(grads, {k: g['%s.grad0.%s' % (tag, k)] for k in keys}, keys, bool(bn))
        assert net.grads_in_flat_buffer()
        losses += [float(tr.train_step(batch, True)) for _ in range(2)]
    else:
        step = tr.make_graphed_step(batch, warmup=1)        # its warm-up step is step 1
        losses = [float(step.warmup_loss)] + [float(step(batch)) for _ in range(2)]
    assert tr.optimizer.step_count == 3
    assert np.allclose(losses, g[tag + '.losses'], rtol=1e-5, atol=1e-6), (losses, g[tag + '.losses'])
    params = {k: p.detach().cpu().numpy() for k, p in net.named_parameters()}
    check_params(params, {k: g['%s.after.%s' % (tag, k)] for k in keys}, keys, bool(bn),
                 1e-5 if oname in ('sgd', 'adadelta') else 3e-4)


@pytest.mark.parametrize('bn', [0, 1])
@pytest.mark.parametrize('mode', ['eager', 'graph'])
def test_c2_timed_step_five_adadelta_steps_vs_reference(bn, mode):
    """BASELINE.json configs[1] (40->500x2->100, coscos2(avg=False), B=4096, Adadelta 0.1)
    through the stepper bench.py times.  Losses of 5 steps at 2e-5, full-tensor checksums
    of the first-step gradients (`gchk`: sum, sum|.|, max|.|) and of the parameters after
    the 5 steps (`after_chk`), all produced by the reference (tools/make_golden.py g4)."""
    from oracle import torch_ref
    g = load_golden('train_c2_bn%d.npz' % bn)
    net, kw = cuda_net(g, seed=2, prefix=None)
    for k, v in net.state_dict().items():
        v = v.double().cpu()
        assert np.allclose([float(v.sum()), float(v.abs().sum())], g['chk.' + k], rtol=1e-9), k
    tr = trainer(net, 'coscos2', 0, 'adadelta')
    batches = []
    for s in range(2):
        x1, x2, y = torch_ref.make_inputs(4096, 40, 20 + s)
        batches.append((x1.cuda(), x2.cuda(), torch.from_numpy(y).cuda()))
    net.train()
    # the timed path's own forward (direct_forward: no autograd) on the first batch: all 2 x 4096 rows
    from test_gpu_siamese import check_all_embedding_rows
    emb, _ = net.direct_forward(batches[0][0], batches[0][1])
    check_all_embedding_rows(g, emb[:4096], emb[4096:])
    if bn:      # that forward moved BatchNorm's running statistics: start again from the seed
        net, kw = cuda_net(g, seed=2, prefix=None)
        tr = trainer(net, 'coscos2', 0, 'adadelta')
        net.train()

    def check_first_gradient():
        gmax = max(float(g['gchk.' + k][2]) for k, _ in net.named_parameters())
        for k, q in net.named_parameters():
            gg = q.grad.double().cpu()
            ref_sum, ref_abs, ref_max = [float(v) for v in g['gchk.' + k]]
            pre_bn_bias = bn and k.endswith('bias') and (k.endswith('.0.bias') or int(k.split('.')[1]) % 4 == 0)
            if pre_bn_bias:        # mathematically zero, rounding noise on both sides
                assert float(gg.abs().max()) <= 1e-4 * gmax, k
                continue
            # the reference's own fp32 gradient is 1e-5 .. 1.1e-4 away from an fp64
            # evaluation at this initialisation: 3e-4 of the
            # tensor's mass, and of its largest entry
            assert abs(float(gg.abs().sum()) - ref_abs) <= 3e-4 * ref_abs, k
            assert abs(float(gg.sum()) - ref_sum) <= 3e-4 * ref_abs, k
            assert abs(float(gg.abs().max()) - ref_max) <= 3e-4 * max(ref_max, 1e-2 * gmax), k
            assert rel_err(q.grad.cpu().numpy().reshape(q.shape[0], -1)[:4], g['grow.' + k], 1e-6) < 3e-4, k

    if mode == 'eager':
        losses = []
        for s in range(5):
            losses.append(float(tr.train_step(batches[s % 2], True)))
            if s == 0:
                from abnet3_amd import _lib
                # with BatchNorm this fixture pins the RESIDENT tower (one launch per direction) to the reference: on a
                # device that cannot hold the grid, or without the sync buffer, the test would silently judge the layer launches
                # (the split arithmetics; precision='fp32' runs the per-layer GEMM kernels)
                assert not bn or net.precision == 'fp32' or (_lib.last_forward_path() == _lib.PATH_BN_TOWER and
                                                              _lib.last_backward_path() == _lib.PATH_BN_TOWER), \
                    (net.precision, _lib.last_forward_path(), _lib.last_backward_path())
                check_first_gradient()
    else:
        step = tr.make_graphed_step(batches[0], warmup=1)     # its warm-up step is step 1
        losses = [float(step.warmup_loss)] + [float(step(batches[s % 2])) for s in range(1, 5)]
    assert abs(losses[0] - g['losses'][0]) <= 1e-5 * abs(g['losses'][0]), (losses[0], g['losses'][0])      # step 0: nothing has drifted yet
    assert np.allclose(losses, g['losses'], rtol=2e-5), (losses, g['losses'])
    for k, v in net.state_dict().items():
        if 'num_batches' in k or (bn and k.endswith('bias')) or 'running_mean' in k:
            continue
        v = v.double().cpu()
        ref_sum, ref_abs = [float(x) for x in g['after_chk.' + k]]
        # both checksums to 1e-3 of the tensor's MASS (a signed sum cancels: judged against sum |.|, not against itself;
        # the reference's own fp32 biases sit 5e-5 .. 5e-4 from a float64 run after these five steps): what "close"
        # means against the truth is test_c2_backward_judged_against_float64's business
        assert abs(float(v.sum()) - ref_sum) <= 1e-3 * ref_abs + 1e-9 and abs(float(v.abs().sum()) - ref_abs) <= 1e-3 * ref_abs + 1e-9, k


@pytest.mark.parametrize('bn', [0, 1])
def test_c2_backward_judged_against_float64(bn):
    """The C2 step's gradients and its parameters after five Adadelta steps, HIP and reference both measured
    against the SAME sequence of operations in float64 (oracle/torch_ref.run_steps: the float32 run there is what
    the reference executes -- its checksums are the fixture's -- the float64 run the truth both approximate).  At
    this initialisation (cos in [0.99998, 0.999997]) the reference's own fp32 gradients sit 6e-6 .. 3e-4 from the
    truth, the output layer's worst; the HIP gradients must be no further from it than the reference's are
    (factor 1.5 + 3e-6 of the tensor's largest entry for the tensors the reference gets almost exactly; factor 4
    with BatchNorm, whose float32 batch statistics add rounding of their own on both sides), and so must the
    parameters after the five steps."""
    from oracle import torch_ref
    g = load_golden('train_c2_bn%d.npz' % bn)
    kw = ast.literal_eval(str(g['kw']))
    batches = [torch_ref.make_inputs(4096, 40, 20 + s) for s in range(2)]
    l64, g64, p64 = torch_ref.run_steps(kw, 2, batches, 5, torch.float64)
    l32, g32, p32 = torch_ref.run_steps(kw, 2, batches, 5, torch.float32)
    for k in g32:        # the float32 run IS the reference's: the fixture's checksums (produced by the reference itself,
        if is_pre_bn_bias(k, bool(bn)):
            continue
        gg = g32[k].astype(np.float64)          # on another machine: another BLAS, another order of sums -- the reference's
        ref_sum, ref_abs, ref_max = [float(v) for v in g['gchk.' + k]]      # fp32 gradients move by 1e-4 of a tensor's mass)
        assert abs(gg.sum() - ref_sum) <= 3e-4 * ref_abs and abs(np.abs(gg).sum() - ref_abs) <= 3e-4 * ref_abs, k
        assert abs(np.abs(gg).max() - ref_max) <= 3e-4 * max(ref_max, 1e-2 * float(g['gchk.output_layer.0.weight'][2])), k
    net, _ = cuda_net(g, seed=2, prefix=None)
    tr = trainer(net, 'coscos2', 0, 'adadelta')
    dev_batches = [(a.cuda(), b.cuda(), torch.from_numpy(y).cuda()) for a, b, y in batches]
    net.train()
    losses = []
    for s in range(5):
        losses.append(float(tr.train_step(dev_batches[s % 2], True)))
        if s == 0:
            ghip = {k: q.grad.double().cpu().numpy() for k, q in net.named_parameters()}
    err = lambda a, t: float(np.abs(a.astype(np.float64) - t).max() / max(np.abs(t).max(), 1e-30))
    gmax = max(np.abs(v).max() for v in g64.values())
    worst, factor = 0.0, (4.0 if bn else 1.5)
    for k in g64:
        if is_pre_bn_bias(k, bool(bn)):           # mathematically zero: rounding noise on every side
            assert np.abs(ghip[k]).max() <= 1e-4 * gmax, k
            continue
        e_hip, e_ref = err(ghip[k], g64[k]), err(g32[k], g64[k])
        worst = max(worst, e_hip / max(e_ref, 1e-12))
        assert e_hip <= factor * e_ref + 3e-6, (k, e_hip, e_ref)
    # losses: each within the reference's own distance from the truth (+ 1e-6 relative)
    for a, b, c in zip(losses, l32, l64):
        assert abs(a - c) <= abs(b - c) + 2e-6 * abs(c), (a, b, c)
    sd = {k: v.double().cpu().numpy() for k, v in net.state_dict().items()}
    for k in p64:
        if 'num_batches' in k or is_pre_bn_bias(k, bool(bn)) or 'running' in k:
            continue
        e_hip, e_ref = err(sd[k], p64[k]), err(p32[k], p64[k])
        assert e_hip <= factor * e_ref + 3e-6, (k, e_hip, e_ref)


class _ListLoader(object):
    def __init__(self, train, dev_batches):
        self.train, self.dev = train, dev_batches

    def batch_iterator(self, train_mode=True):
        return iter(self.train if train_mode else self.dev)

    def whoami(self):
        return {'class_name': 'ListLoader'}


@pytest.mark.parametrize('oname', ['adam', 'adadelta'])
def test_auto_graph_with_interleaved_eager_steps_uses_fresh_gradients(oname):
    """A replayed step whose optimizer launch is NOT captured (Adam: host-side bias
    correction) must step on the gradient the replay just wrote, even when eager steps
    of other batch shapes ran in between and re-pointed p.grad at their own buffer."""
    import abnet3_amd.loss as L
    from abnet3_amd.trainer import TrainerSiamese
    g = load_golden('train_mid_bn0.npz')
    rng = np.random.default_rng(17)

    def batch(n):
        return (dev(rng.standard_normal((n, 40)).astype(np.float32)),
                dev(rng.standard_normal((n, 40)).astype(np.float32)), dev(rng.choice([1.0, -1.0], n)))
    train = [batch(64) for _ in range(4)] + [batch(37)] + [batch(64), batch(21), batch(64), batch(64)]
    out = []
    for graph_steps in (False, True):
        net, _ = cuda_net(g)
        tr = TrainerSiamese(network=net, loss=L.coscos2(avg=True), optimizer_type=oname, lr=0.01,
                            dataloader=_ListLoader(train, [batch(30)]), log_dir='/tmp/abn_runs')
        tr.graph_steps = graph_steps
        tr.train_losses, tr.dev_losses = [], []
        for _ in range(2):
            tr.optimize_model(do_training=True)
        assert (len(getattr(tr, '_graphs', {})) == 1) == graph_steps
        out.append((tr.train_losses, {k: p.detach().cpu().numpy().copy() for k, p in net.state_dict().items()}))
    assert np.allclose(out[0][0], out[1][0], rtol=1e-6)
    for k, v in out[1][1].items():
        # (fp16 x 2, small batches: eager steps sum their weight gradients over all rows in the optimizer's launch, captured
        # autograd steps in slabs -- the same products in another order, two epochs on)
        assert rel_err(v, out[0][1][k]) < 1e-5, k


def test_save_whoami_pickles_the_reference_dictionary(tmp_path):
    """TrainerBuilder.save_whoami (abnet3/trainer.py:106-108) pickles whoami(): the
    network / loss / dataloader descriptions under the reference's keys."""
    import pickle
    import abnet3_amd.loss as L
    from abnet3_amd.model import SiameseNetwork
    from abnet3_amd.trainer import TrainerSiamese
    net = SiameseNetwork(input_dim=40, num_hidden_layers=1, hidden_dim=32, output_dim=16, p_dropout=0.0,
                         activation_layer='sigmoid', output_path=str(tmp_path / 'network'))
    tr = TrainerSiamese(network=net, loss=L.cosmargin(avg=False, margin=0.4), optimizer_type='adadelta', lr=0.1,
                        dataloader=_ListLoader([], []), log_dir=str(tmp_path / 'runs'))
    x = torch.randn(8, 40).cuda()
    tr.train_step((x, x.flip(0), torch.ones(8).cuda()), True)          # HIP state exists by now
    tr.save_whoami()
    with open(str(tmp_path / 'network') + '.params', 'rb') as fh:
        info = pickle.load(fh)
    assert set(info) == {'params', 'network', 'loss', 'class_name', 'dataloader'}
    assert info['class_name'] == 'TrainerSiamese' and info['network']['class_name'] == 'SiameseNetwork'
    assert info['loss']['class_name'] == 'cosmargin' and info['loss']['params']['margin'] == 0.4
    p = info['network']['params']
    assert p['input_dim'] == 40 and p['hidden_dim'] == 32 and p['activation_layer'] == 'sigmoid'
    assert not any(k in p for k in SiameseNetwork._HIP_STATE)
    # the network still steps after having described itself, and a no-op .cuda() keeps
    # the flat buffer (and with it the optimizer state) in place
    flat_before = net.flat_parameters().data_ptr()
    net.cuda()
    assert net.flat_parameters().data_ptr() == flat_before
    tr.train_step((x, x.flip(0), torch.ones(8).cuda()), True)


def test_strided_inputs_are_accepted():
    """The reference takes any tensor nn.Linear takes, e.g. a column slice of stacked
    features (non-contiguous)."""
    from abnet3_amd.model import SiameseNetwork
    torch.manual_seed(0)
    net = SiameseNetwork(input_dim=40, num_hidden_layers=0, hidden_dim=24, output_dim=8, p_dropout=0.0,
                         activation_layer='tanh').cuda().eval()
    wide = torch.randn(33, 280).cuda()
    sl = wide[:, 120:160]
    assert not sl.is_contiguous()
    with torch.no_grad():
        a = net.forward_once(sl)
        b = net.forward_once(sl.contiguous())
    assert torch.equal(a, b)


@pytest.mark.parametrize('oname', ['sgd', 'adadelta', 'adam', 'adagrad', 'RMSprop'])
@pytest.mark.parametrize('fixture', ['train_c1_bn0.npz', 'train_c1_bn1.npz', 'train_mid_bn1.npz'])
def test_fused_reduce_and_step_equals_the_two_launches_bit_for_bit(oname, fixture, monkeypatch):
    """abn_tower_reduce_step (split-K reduction + optimizer update in one launch, what
    train_step uses in a single process without BatchNorm) against slab_reduce followed by
    abn_optimizer_step: same summation order, same update arithmetic -> identical bits in
    the gradients, the parameters and the optimizer state after three steps."""
    g = load_golden(fixture)          # (with BatchNorm: its gamma / beta are stepped by the same launch)
    sfx = '' if 'x1' in g else '.0'
    batch = (dev(g['x1' + sfx]), dev(g['x2' + sfx]), dev(g['y' + sfx]))
    runs = []
    for fused in ('1', '0'):
        monkeypatch.setenv('ABN_FUSED_STEP', fused)
        net, _ = cuda_net(g)
        tr = trainer(net, 'coscos2', 0, oname)
        net.train()
        for s in range(3):
            tr.train_step(batch, True)
            pending = getattr(net, '_pending_reduce', None)
            assert pending is None                       # step() consumed it
        assert net.grads_in_flat_buffer()
        runs.append(([p.grad.clone() for p in net.parameters()], [p.detach().clone() for p in net.parameters()],
                     tr.optimizer._s1.clone(), tr.optimizer._s2.clone(), net._offsets, net))
    (ga, pa, s1a, s2a, offs, net), (gb, pb, s1b, s2b, _, _) = runs
    for a, b in zip(ga + pa, gb + pb):
        assert torch.equal(a, b)
    for p, off in zip(net.live_parameters(), offs):      # state of the real elements (padding is never touched fused)
        n = p.numel()
        assert torch.equal(s1a[off:off + n], s1b[off:off + n]) and torch.equal(s2a[off:off + n], s2b[off:off + n])
