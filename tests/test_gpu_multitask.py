"""SiameseMultitaskNetwork + weighted_loss_multi +
TrainerSiameseMultitask on the HIP path (through the C-ABI, behind the reference
class surface) against the golden vectors the reference produced
(tools/make_golden.py G8) and against the numpy oracle.  Needs an MI355X."""
import ast

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err, check_grads, check_params

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=['bf16x3', 'f16x2', 'fp32'])
def tower_precision(request, monkeypatch):
    """Every test of this file runs on both parity-grade arithmetics of the tower GEMMs: the
    default bf16 x 3 split products and the exact-fp32 MFMA (SiameseNetwork.precision)."""
    monkeypatch.setenv('ABNET3_PRECISION', request.param)
    return request.param
TOL = 1e-5
GTOL = {'sig': 1e-4, 'relu_bn': 2e-5, 'tanh0': 2e-5}    # see tests/test_oracle_multitask.py
RUNS = [('sgd', 0.3, 0.001), ('adadelta', 0.5, 0.1), ('adam', 1.0, 0.001)]
OPT = {'sgd': lambda p: torch.optim.SGD(p, lr=0.001, momentum=0.9),
       'adadelta': lambda p: torch.optim.Adadelta(p, lr=0.1),
       'adam': lambda p: torch.optim.Adam(p, lr=0.001)}


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def cuda_net(g):
    from abnet3_amd.model import SiameseMultitaskNetwork
    kw = ast.literal_eval(str(g['kw']))
    net = SiameseMultitaskNetwork(**kw)
    net.load_state_dict({k[2:]: torch.from_numpy(v.copy()) for k, v in g.items()
                         if k.startswith('p.')})
    return net.cuda(), kw


def make_loss(weight):
    import abnet3_amd.loss as L
    return L.weighted_loss_multi(loss_spk=L.coscos2(avg=False),
                                 loss_phn=L.cosmargin(avg=True, margin=0.4), weight=weight).cuda()


@pytest.mark.parametrize('name', ['sig', 'relu_bn', 'tanh0'])
def test_same_seed_same_initial_weights(name):
    """Construction consumes the RNG like the reference: one torch.manual_seed
    gives the reference's state_dict, the never-called branches included."""
    from abnet3_amd.model import SiameseMultitaskNetwork
    g = load_golden('multitask_%s.npz' % name)
    torch.manual_seed(5)
    net = SiameseMultitaskNetwork(**ast.literal_eval(str(g['kw'])))
    sd = net.state_dict()
    assert sorted(sd) == sorted(k[2:] for k in g if k.startswith('p.'))
    for k, v in sd.items():
        assert np.array_equal(v.numpy(), g['p.' + k]), k


@pytest.mark.parametrize('name', ['sig', 'relu_bn', 'tanh0'])
def test_multitask_forward_eval(name):
    g = load_golden('multitask_%s.npz' % name)
    net, _ = cuda_net(g)
    net.eval()
    with torch.no_grad():
        out = net(dev(g['x1']), dev(g['x2']))
        once = net.forward_once(dev(g['x1']))
    for mine, key in zip(out, ('spk1', 'phn1', 'spk2', 'phn2')):
        assert rel_err(mine.cpu().numpy(), g['eval_' + key]) < TOL, key
    assert rel_err(once[0].cpu().numpy(), g['eval_spk1']) < TOL
    assert rel_err(once[1].cpu().numpy(), g['eval_phn1']) < TOL


@pytest.mark.parametrize('name', ['sig', 'relu_bn', 'tanh0'])
@pytest.mark.parametrize('oname,weight,lr', RUNS)
def test_multitask_grads_and_three_steps(name, oname, weight, lr):
    """trainer.py:236-242 around TrainerSiameseMultitask.give_batch_to_network
    (trainer.py:266-279), torch.optim driving the HIP network's parameters."""
    g = load_golden('multitask_%s.npz' % name)
    net, kw = cuda_net(g)
    bn = bool(kw['batch_norm'])
    loss_mod = make_loss(weight)
    opt = OPT[oname](net.parameters())
    x1, x2 = dev(g['x1']), dev(g['x2'])
    y_spk, y_phn = dev(g['y_spk']), dev(g['y_phn'])
    tag = '%s.w%g' % (oname, weight)
    live = [k for k, p in net.named_parameters()
            if not k.startswith(('hidden_layers_spk', 'hidden_layers_phn'))]
    dead = [k for k, p in net.named_parameters() if k not in live]
    net.train()
    losses = []
    for s in range(3):
        emb = net(x1, x2)
        lv = loss_mod(emb[0], emb[1], emb[2], emb[3], y_spk, y_phn)
        opt.zero_grad()
        lv.backward()
        if s == 0:
            for mine, key in zip(emb, ('spk1', 'phn1', 'spk2', 'phn2')):
                assert rel_err(mine.detach().cpu().numpy(), g['%s.%s_0' % (tag, key)]) < TOL, key
            named = dict(net.named_parameters())
            for k in dead:               # like the reference: no gradient at all
                assert named[k].grad is None, k
            grads = {k: named[k].grad.cpu().numpy() for k in live}
            ref = {k: g['%s.grad0.%s' % (tag, k)] for k in live}
            cmp_keys = live
            if weight == 1.0:
                for k in live:
                    if k.startswith('output_layer_phn'):
                        assert not np.any(grads[k]) and not np.any(ref[k]), k
                cmp_keys = [k for k in live if not k.startswith('output_layer_phn')]
            check_grads(grads, ref, cmp_keys, bn, GTOL[name])
            assert net.grads_in_flat_buffer()      # three segments, ONE flat bucket
        opt.step()
        losses.append(float(lv.detach()))
    assert np.allclose(losses, g[tag + '.losses'], rtol=1e-5, atol=1e-6)
    params = {k: p.detach().cpu().numpy() for k, p in net.named_parameters()}
    check_params(params, {k: g['%s.after.%s' % (tag, k)] for k in live}, live, bn,
                 max(5 * GTOL[name], 3e-4) if oname == 'adam' else
                 5 * GTOL[name] if name == 'sig' else 1e-5)
    for k in dead:
        assert np.array_equal(params[k], g['p.' + k]), k
    if bn and oname != 'adam':
        sd = net.state_dict()
        for k, v in sd.items():
            if 'running' in k:
                assert rel_err(v.cpu().numpy(), g['%s.after.%s' % (tag, k)]) < TOL, k
            if 'num_batches_tracked' in k:
                assert int(v) == int(g['%s.after.%s' % (tag, k)]), k


@pytest.mark.parametrize('name', ['sig', 'relu_bn'])
def test_trainer_multitask_step_matches_oracle(name):
    """TrainerSiameseMultitask.train_step (fused flat optimizer) and its captured
    hipGraph replay against the numpy oracle, five Adadelta steps."""
    from abnet3_amd.trainer import TrainerSiameseMultitask
    from oracle import siamese_np as O
    g = load_golden('multitask_%s.npz' % name)
    kw = ast.literal_eval(str(g['kw']))
    spec = O.MultitaskSpec(kw['input_dim'], kw['num_hidden_layers_shared'],
                           kw['num_hidden_layers_spk'], kw['num_hidden_layers_phn'],
                           kw['hidden_dim'], kw['output_dim'], kw['activation_layer'],
                           kw['batch_norm'])
    batch = (dev(g['x1']), dev(g['x2']), dev(g['y_spk']), dev(g['y_phn']))
    for graphed in (False, True):
        net, _ = cuda_net(g)
        net.output_path = '/tmp/abn_multitask_test'
        tr = TrainerSiameseMultitask(network=net, loss=make_loss(0.3), optimizer_type='adadelta',
                                     lr=0.1, dataloader=None, log_dir='/tmp/abn_runs')
        p = {k[2:]: v.copy() for k, v in g.items() if k.startswith('p.')}
        oopt = O.Optimizer('adadelta', 0.1)
        net.train()
        n_warm = 0
        if graphed:
            step = tr.make_graphed_step(batch, warmup=2)
            n_warm = 2
        for _ in range(n_warm):
            O.multitask_train_step(p, g['x1'], g['x2'], g['y_spk'], g['y_phn'], spec, oopt, 0.3,
                                   spk=('coscos2', 0.5, False), phn=('cosmargin', 0.4, True))
        for s in range(5):
            lv = step(batch) if graphed else tr.train_step(batch, True)
            ref, _, _ = O.multitask_train_step(
                p, g['x1'], g['x2'], g['y_spk'], g['y_phn'], spec, oopt, 0.3,
                spk=('coscos2', 0.5, False), phn=('cosmargin', 0.4, True))
            assert abs(float(lv) - ref) <= 1e-5 * abs(ref) + 1e-6, (graphed, s)
        mine = {k: v.detach().cpu().numpy() for k, v in net.named_parameters()}
        keys = spec.live_param_keys()
        check_params(mine, p, keys, spec.batch_norm, 5 * GTOL[name] if name == 'sig' else 2e-5)
        for k in spec.dead:
            assert np.array_equal(mine[k + '.weight'], g['p.' + k + '.weight'])


def test_multitask_loader_speaker_labels():
    """load_frames_from_pairs(fid2spk=...) (dataloader.py:166-261): frames, phone
    labels and the reference's `is`-comparison of speaker ids, against the fixture
    the reference produced; and MultiTaskDataLoader.batch_iterator's 4-tuples."""
    import os
    import tempfile
    from abnet3_amd.dataloader import OriginalDataLoader, MultiTaskDataLoader, DeviceCorpus
    from abnet3_amd.utils import group_pairs, read_spkid_file, write_dataset
    g = load_golden('multitask_frames.npz')
    feats = {k[5:]: v for k, v in g.items() if k.startswith('feat.')}
    times = {k: (np.arange(len(v)) * 0.01 + 0.0025) for k, v in feats.items()}
    pairs = []
    for line in g['pairs']:
        f1, s1, e1, f2, s2, e2, t = str(line).split(' ')
        pairs.append((f1, float(s1), float(e1), f2, float(s2), float(e2), t))
    with tempfile.TemporaryDirectory() as d:
        spk = os.path.join(d, 'fid2spk')
        open(spk, 'w').write(str(g['spk_file']))
        fid2spk = read_spkid_file(spk)
        dl = OriginalDataLoader('unused', 'unused', align_different_words=False)
        dl.features = DeviceCorpus(feats, times)
        X1, X2, y_spk, y_phn = dl.load_frames_from_pairs(group_pairs(pairs), fid2spk=fid2spk)
        assert np.array_equal(X1, g['X1']) and np.array_equal(X2, g['X2'])
        assert np.array_equal(y_phn, g['y_phn'])
        assert np.array_equal(y_spk, g['y_spk'])          # 'alice' in two files: -1, like the reference
        assert y_spk.dtype == np.float64
        dl.speaker_match = 'equal'
        y_eq = dl.load_frames_from_pairs(group_pairs(pairs), fid2spk=fid2spk)[2]
        assert (y_eq == 1).sum() > (y_spk == 1).sum()

        for sub in ('train_pairs', 'dev_pairs'):
            os.makedirs(os.path.join(d, sub))
            write_dataset(os.path.join(d, sub, 'dataset'), pairs)
        ml = MultiTaskDataLoader(d, 'unused', fid2spk_file=spk, batch_size=2,
                                 align_different_words=False)
        ml.features = DeviceCorpus(feats, times)
        batches = list(ml.batch_iterator(train_mode=True))
        assert len(batches) == 2 and all(len(b) == 4 for b in batches)
        assert sum(len(b[2]) for b in batches) == len(g['y_spk'])
        assert all(b[0].is_cuda and b[2].dtype == torch.float64 for b in batches)


def test_multitask_embedder():
    from abnet3_amd.embedder import EmbedderSiameseMultitask
    g = load_golden('multitask_relu_bn.npz')
    net, _ = cuda_net(g)
    emb = EmbedderSiameseMultitask(network=net, feature_path=None, output_path=None)
    spk, phn = emb.embed_features([g['x1'], g['x2'].astype(np.float64), np.zeros((0, 40), np.float32)])
    assert rel_err(spk[0], g['eval_spk1']) < TOL and rel_err(phn[0], g['eval_phn1']) < TOL
    assert rel_err(spk[1], g['eval_spk2']) < TOL and rel_err(phn[1], g['eval_phn2']) < TOL
    assert spk[2].shape == (0, 24) and phn[2].shape == (0, 24)
