"""Ragged / odd / tiny shapes through the HIP path against the numpy oracle (which
the golden vectors pin on regular shapes): widths that are not multiples of 4
(scalar-load GEMM path, no fused forward), row counts that are not multiples of any
tile, one row, empty batches, maximum layer count."""
import numpy as np
import pytest
import torch

from conftest import rel_err, check_grads

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


def run_case(kw, B, seed, loss_kind='coscos2', avg=False, tol=TOL):
    import abnet3_amd.loss as L
    from abnet3_amd.model import SiameseNetwork
    from oracle import siamese_np as O
    torch.manual_seed(seed)
    net = SiameseNetwork(p_dropout=0.0, **kw).cuda()
    spec = O.TowerSpec(kw['input_dim'], kw['num_hidden_layers'], kw['hidden_dim'], kw['output_dim'],
                       kw['activation_layer'], kw.get('batch_norm', False),
                       kw.get('last_non_linearity', 'default'))
    p = {k: v.detach().cpu().numpy().copy() for k, v in net.state_dict().items()}
    rng = np.random.default_rng(seed)
    x1 = rng.standard_normal((B, kw['input_dim'])).astype(np.float32)
    x2 = rng.standard_normal((B, kw['input_dim'])).astype(np.float32)
    y = rng.choice([1, -1], B)
    net.train()
    e1, e2 = net(dev(x1), dev(x2))
    lv = getattr(L, loss_kind)(avg=avg)(e1, e2, dev(y))
    lv.backward()
    o1, c1 = O.tower_forward(p, x1, spec, True)
    o2, c2 = O.tower_forward(p, x2, spec, True)
    ol, d1, d2, _ = O.pair_loss(o1, o2, y, loss_kind, 0.5, avg)
    og = {}
    O.tower_backward(p, c1, d1, spec, og)
    O.tower_backward(p, c2, d2, spec, og)
    assert rel_err(e1.detach().cpu().numpy(), o1) < tol
    assert rel_err(e2.detach().cpu().numpy(), o2) < tol
    assert abs(float(lv.detach()) - ol) <= tol * abs(ol) + 1e-6
    grads = {k: q.grad.cpu().numpy() for k, q in net.named_parameters()}
    check_grads(grads, og, spec.param_keys(), spec.batch_norm, tol=1e-4)
    net.eval()
    with torch.no_grad():
        ev = net.forward_once(dev(x1))
    oe, _ = O.tower_forward(p, x1, spec, False)
    assert rel_err(ev.cpu().numpy(), oe) < tol


@pytest.mark.parametrize('act', ['sigmoid', 'tanh', 'relu'])
@pytest.mark.parametrize('bn', [False, True])
def test_odd_widths_and_ragged_rows(act, bn):
    # 39 -> 50 -> 50 -> 7: nothing is a multiple of 4, 33 rows per tower
    run_case(dict(input_dim=39, num_hidden_layers=1, hidden_dim=50, output_dim=7,
                  activation_layer=act, batch_norm=bn), B=33, seed=1)


@pytest.mark.parametrize('B', [1, 2, 63, 65, 130])
def test_row_counts_around_tile_edges(B, forward_path):
    run_case(dict(input_dim=40, num_hidden_layers=1, hidden_dim=72, output_dim=36,
                  activation_layer='tanh', batch_norm=False), B=B, seed=B, loss_kind='cosmargin', avg=True)


def test_wide_and_deep_tower():
    # wider than the fused-forward image (600 > 512) and 14 hidden layers (16 Linear in all)
    run_case(dict(input_dim=40, num_hidden_layers=0, hidden_dim=600, output_dim=24,
                  activation_layer='relu', batch_norm=False), B=40, seed=3)
    run_case(dict(input_dim=24, num_hidden_layers=14, hidden_dim=32, output_dim=16,
                  activation_layer='tanh', batch_norm=True), B=24, seed=4,
             tol=5e-5)      # 16 BatchNorm'd layers deep: fp32 summation-order noise adds up
    from abnet3_amd.model import SiameseNetwork
    with pytest.raises(AssertionError):
        SiameseNetwork(input_dim=24, num_hidden_layers=15, hidden_dim=32, output_dim=16,
                       activation_layer='tanh')


def test_empty_batch_is_refused_or_empty():
    """nn.Linear on zero rows returns zero rows in the reference; the HIP path must
    not launch on an empty grid: it returns empty embeddings."""
    from abnet3_amd.model import SiameseNetwork
    net = SiameseNetwork(input_dim=40, num_hidden_layers=0, hidden_dim=16, output_dim=8,
                         p_dropout=0.0, activation_layer='sigmoid').cuda()
    net.eval()
    x = torch.zeros(0, 40, device='cuda')
    with torch.no_grad():
        e = net.forward_once(x)
    assert e.shape == (0, 8)
    torch.cuda.synchronize()


@pytest.mark.parametrize('seed', range(12))
def test_random_tower_configurations(seed, forward_path):
    """Random depth / widths (multiples of 4 or not) / activation / BatchNorm / output
    head / batch size / loss, forward + loss + every gradient against the numpy oracle."""
    rng = np.random.default_rng(1000 + seed)
    act = str(rng.choice(['sigmoid', 'tanh', 'relu']))
    hidden = int(rng.choice([8, 30, 64, 100, 257]))
    kw = dict(input_dim=int(rng.choice([5, 16, 40, 77])), num_hidden_layers=int(rng.integers(0, 4)),
              hidden_dim=hidden, output_dim=int(rng.choice([3, 20, 64])), activation_layer=act,
              batch_norm=bool(rng.integers(0, 2)))
    if rng.integers(0, 3) == 0:
        kw['last_non_linearity'] = None
    B = int(rng.integers(2, 150))
    run_case(kw, B=B, seed=seed, loss_kind=str(rng.choice(['coscos2', 'cosmargin'])), avg=bool(rng.integers(0, 2)),
             tol=3e-5)
