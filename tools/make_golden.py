#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference); never on the GPU box.
The reference's modules import five third-party packages that are absent here
(h5features, dtw, spectral, h5py, tensorboardX); they are registered as empty
stub modules in THIS process only, so that abnet3.model / abnet3.loss /
abnet3.utils.cosine_distance / FeaturesGenerator.stack_fbanks can be imported
and executed unmodified on torch-CPU.  Nothing of the reference is copied: the
fixtures are inputs + the outputs the reference produced for them.

Fixture sets:
  G1 tower_*      SiameseNetwork forward, eval + train mode, BN on/off
  G2 train_c1_*   loss scalar + every parameter gradient + params after 3 steps
  G3 loss_edge    pair losses on edge-case rows (label 0, tiny norms, margin)
  G4 train_mid / train_c2   reduced-width C2 with full tensors; true C2 scalars
  G5 cosdist      abnet3.utils.cosine_distance
  G6 stack        FeaturesGenerator.stack_fbanks
  G7 frames       OriginalDataLoader.load_frames_from_pairs (needs the oracle DTW
                  monkey-patched in as get_dtw_alignment; third-party dtw absent)
  G8 multitask_*  SiameseMultitaskNetwork + weighted_loss_multi: forward (eval,
                  train), every gradient, params after 3 steps; and the speaker
                  labels of load_frames_from_pairs(fid2spk=...)
  G5L cosdist_libm  abnet3.utils.cosine_distance with numpy on its plain-libm path
                  (NPY_DISABLE_CPU_FEATURES drops numpy's AVX512/SVML arccos; the
                  generator re-runs itself in a child process with that variable) on
                  matrices large enough for OpenBLAS's regular sgemm kernel: what the
                  oracle and the HIP kernel reproduce bit for bit, plus the per-pair
                  drop decisions (AssertionError) for near-duplicate tokens
  G10 mvn        the four normalisation cases of the reference's own test/test_features.py:37-281
                  (global / per file, with and without a VAD, per channel and over the whole spectrum),
                  re-enacted in memory: the inputs are that test's literals, the VAD file is read and
                  applied by the reference's read_vad_file / filter_vad_* (pure numpy, importable), the
                  expected statistics and outputs are the expressions that test asserts
  G11 gridsearch  the `default_params` of the reference's test/data/buckeye.yaml as a dictionary, and what
                  the reference's own classes make of it when built the way gridsearch.py:145-202 builds them
                  (which keyword arguments they accept, parameter count, state_dict keys, optimizer class)
  G9 frames_loader  FramesDataLoader.load_all_frames / load_batch / batch_iterator
                  (shuffles, batch slicing, max_batches_per_epoch wrap-around) and
                  OriginalDataLoader.add_tcl_to_batch / temporal_coherence_loss

usage: python tools/make_golden.py [--only G1,G3]
"""
import argparse
import os
import sys
import types
import warnings

import numpy as np

REF = '/root/reference'
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, 'tests', 'golden')


def import_reference():
    for name in ['h5features', 'dtw', 'spectral', 'h5py', 'tensorboardX']:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules['dtw'].DTW = None
    sys.modules['spectral'].Spectral = None
    sys.modules['tensorboardX'].SummaryWriter = None
    sys.path.insert(0, REF)
    import scipy
    if not hasattr(scipy, 'arccos'):
        scipy.arccos = np.arccos      # removed from scipy >= 1.12 (utils.py:50,53)
    warnings.simplefilter('ignore')
    import torch  # noqa
    import abnet3.model
    import abnet3.loss
    import abnet3.utils
    import abnet3.features
    import abnet3.dataloader
    return abnet3


def sd_to_np(net, prefix='p.'):
    return {prefix + k: v.detach().cpu().numpy().copy()
            for k, v in net.state_dict().items()}


def make_inputs(B, D, seed):
    import torch
    torch.manual_seed(seed)
    x1 = torch.randn(B, D)
    x2 = torch.randn(B, D)
    np.random.seed(seed)
    y = np.random.choice([1, -1], B)
    return x1, x2, y


def build_net(abnet3, seed, **kw):
    import torch
    torch.manual_seed(seed)
    return abnet3.model.SiameseNetwork(**kw)


OPTIMS = {
    'sgd': lambda torch, p: torch.optim.SGD(p, lr=0.001, momentum=0.9),
    'adadelta': lambda torch, p: torch.optim.Adadelta(p, lr=0.1),
    'adam': lambda torch, p: torch.optim.Adam(p, lr=0.001),
    'adagrad': lambda torch, p: torch.optim.Adagrad(p, lr=0.001),
    'RMSprop': lambda torch, p: torch.optim.RMSprop(p, lr=0.001),
}


def g1_tower(abnet3):
    """G1: tower forward in eval and train mode (model.py:179-196)."""
    import torch
    cases = {
        'sig': dict(activation_layer='sigmoid', batch_norm=False),
        'sig_bn': dict(activation_layer='sigmoid', batch_norm=True),
        'relu_bn': dict(activation_layer='relu', batch_norm=True),
        'tanh': dict(activation_layer='tanh', batch_norm=False),
        'sig_lin': dict(activation_layer='sigmoid', batch_norm=False,
                        last_non_linearity=None),
        'relu_h2': dict(activation_layer='relu', batch_norm=False,
                        num_hidden_layers=2),
        'tanh_softmax': dict(activation_layer='tanh', batch_norm=False,
                             last_non_linearity='softmax'),
        'relu_bn_softmax': dict(activation_layer='relu', batch_norm=True,
                                last_non_linearity='softmax'),
    }
    for name, extra in cases.items():
        kw = dict(input_dim=40, num_hidden_layers=0, hidden_dim=100,
                  output_dim=50, p_dropout=0.0, type_init='xavier_uni')
        kw.update(extra)
        net = build_net(abnet3, 0, **kw)
        x1, x2, y = make_inputs(32, 40, 0)
        out = sd_to_np(net)
        out['x1'], out['x2'], out['y'] = x1.numpy(), x2.numpy(), y
        net.eval()
        with torch.no_grad():
            e1, e2 = net(x1, x2)
        out['eval_e1'], out['eval_e2'] = e1.numpy(), e2.numpy()
        net.train()
        with torch.no_grad():
            e1, e2 = net(x1, x2)
        out['train_e1'], out['train_e2'] = e1.numpy(), e2.numpy()
        out.update(sd_to_np(net, 'after.'))   # BN running stats after 2 calls
        if 'softmax' in name:                 # + one backward through the softmax head
            net = build_net(abnet3, 0, **kw)
            net.train()
            e1, e2 = net(x1, x2)
            lv = abnet3.loss.cosmargin(avg=True)(e1, e2, torch.from_numpy(y))
            lv.backward()
            out['loss'] = np.float64(lv.detach())
            for k, p in net.named_parameters():
                out['grad.' + k] = p.grad.detach().numpy().copy()
        out['kw'] = np.array(repr(kw))
        np.savez_compressed(os.path.join(OUT, 'tower_%s.npz' % name), **out)


def run_steps(abnet3, net, loss_mod, optim_name, batches, nsteps, out, tag):
    """Re-enacts trainer.py:236-242 (optimize_model's five statements)."""
    import torch
    opt = OPTIMS[optim_name](torch, net.parameters())
    net.train()
    losses = []
    for s in range(nsteps):
        x1, x2, y = batches[s % len(batches)]
        e1, e2 = net(x1, x2)
        lv = loss_mod(e1, e2, torch.from_numpy(y))
        opt.zero_grad()
        lv.backward()
        if s == 0:
            for k, p in net.named_parameters():
                out['%s.grad0.%s' % (tag, k)] = p.grad.detach().numpy().copy()
            out['%s.e1_0' % tag] = e1.detach().numpy().copy()
            out['%s.e2_0' % tag] = e2.detach().numpy().copy()
        opt.step()
        losses.append(float(lv.detach()))
    out['%s.losses' % tag] = np.array(losses, dtype=np.float64)
    for k, v in net.state_dict().items():
        out['%s.after.%s' % (tag, k)] = v.detach().numpy().copy()


def g2_train_c1(abnet3):
    """G2: C1 = 40->100->50, B=32. Loss, all grads, params after 3 steps."""
    for bn in (False, True):
        out = {}
        kw = dict(input_dim=40, num_hidden_layers=0, hidden_dim=100,
                  output_dim=50, p_dropout=0.0, type_init='xavier_uni',
                  activation_layer='sigmoid', batch_norm=bn)
        x1, x2, y = make_inputs(32, 40, 0)
        out['x1'], out['x2'], out['y'] = x1.numpy(), x2.numpy(), y
        net0 = build_net(abnet3, 0, **kw)
        out.update(sd_to_np(net0))
        for lname in ('coscos2', 'cosmargin'):
            for avg in (True, False):
                for oname in OPTIMS:
                    if oname not in ('sgd', 'adadelta') and not (
                            lname == 'coscos2' and avg is False):
                        continue
                    net = build_net(abnet3, 0, **kw)
                    loss_mod = getattr(abnet3.loss, lname)(avg=avg)
                    tag = '%s.avg%d.%s' % (lname, int(avg), oname)
                    run_steps(abnet3, net, loss_mod, oname, [(x1, x2, y)], 3,
                              out, tag)
        out['kw'] = np.array(repr(kw))
        np.savez_compressed(
            os.path.join(OUT, 'train_c1_bn%d.npz' % int(bn)), **out)


def g3_loss_edge(abnet3):
    """G3: pair losses + autograd gradients on edge-case rows (loss.py:46-105)."""
    import torch
    torch.manual_seed(3)
    B, D = 32, 50
    e1 = torch.randn(B, D)
    e2 = torch.randn(B, D)
    np.random.seed(3)
    y = np.random.choice([1, -1], B).astype(np.int64)
    y[0] = 0            # any other label leaves raw cos (loss.py:60-63)
    y[1] = 2
    e2[2] = e1[2]       # identical rows, same
    y[2] = 1
    e2[3] = e1[3]       # identical rows, diff
    y[3] = -1
    e2[4] = -e1[4]      # opposite rows
    y[4] = -1
    e1[5] = e1[5] * 1e-8   # tiny norm -> eps clamp on one side
    y[5] = 1
    e1[6] = e1[6] * 1e-8
    e2[6] = e2[6] * 1e-8   # both below eps
    y[6] = -1
    e1[7] = 0.0            # exact zero row
    y[7] = 1
    e1[8] = e1[8] * 1e-7   # around eps=1e-6 (norm ~ 7e-7)
    y[8] = -1
    # cos just above / below the default margin 0.5 for cosmargin
    v = torch.randn(D)
    w = torch.randn(D)
    w = w - (w @ v) / (v @ v) * v
    v = v / v.norm()
    w = w / w.norm()
    for row, c in ((9, 0.5005), (10, 0.4995), (11, 0.5)):
        e1[row] = 2.0 * v
        e2[row] = 3.0 * (c * v + float(np.sqrt(1 - c * c)) * w)
        y[row] = -1
    out = {'e1': e1.numpy().copy(), 'e2': e2.numpy().copy(), 'y': y}
    label_sets = {
        'mixed': y,
        'allsame': np.ones(B, dtype=np.int64),
        'alldiff': -np.ones(B, dtype=np.int64),
        'f64': y.astype(np.float64),
    }
    for lsname, yy in label_sets.items():
        out['y.' + lsname] = yy
        for lname, kwargs in (('coscos2', {}), ('cosmargin', {}),
                              ('cosmargin', {'margin': 0.2})):
            for avg in (True, False):
                a = e1.clone().requires_grad_(True)
                b = e2.clone().requires_grad_(True)
                mod = getattr(abnet3.loss, lname)(avg=avg, **kwargs)
                lv = mod(a, b, torch.from_numpy(yy))
                lv.backward()
                tag = '%s.%s%s.avg%d' % (
                    lsname, lname,
                    '' if not kwargs else '_m%g' % kwargs['margin'], int(avg))
                out[tag + '.loss'] = np.array(float(lv.detach()),
                                              dtype=np.float64)
                out[tag + '.de1'] = a.grad.numpy().copy()
                out[tag + '.de2'] = b.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, 'loss_edge.npz'), **out)


def g4_train_mid(abnet3):
    """G4: reduced-width C2 with full tensors; true C2 with scalars only."""
    import torch
    # reduced width, odd sizes on purpose (tile tails): 40 -> 72 x2 -> 36, B=96
    for bn in (False, True):
        out = {}
        kw = dict(input_dim=40, num_hidden_layers=2, hidden_dim=72,
                  output_dim=36, p_dropout=0.0, type_init='xavier_uni',
                  activation_layer='sigmoid', batch_norm=bn)
        batches = []
        for s in range(3):
            x1, x2, y = make_inputs(96, 40, 10 + s)
            batches.append((x1, x2, y))
            out['x1.%d' % s], out['x2.%d' % s], out['y.%d' % s] = \
                x1.numpy(), x2.numpy(), y
        net0 = build_net(abnet3, 1, **kw)
        out.update(sd_to_np(net0))
        for oname in ('sgd', 'adadelta'):
            net = build_net(abnet3, 1, **kw)
            run_steps(abnet3, net, abnet3.loss.coscos2(avg=False), oname,
                      batches, 5, out, 'coscos2.avg0.' + oname)
        out['kw'] = np.array(repr(kw))
        np.savez_compressed(
            os.path.join(OUT, 'train_mid_bn%d.npz' % int(bn)), **out)

    # true C2: 40 -> 500 x2 -> 100, B=4096; weights regenerated from the seed by
    # torch's own initialisers, so only checksums + scalars + a few rows travel.
    for bn in (False, True):
        out = {}
        kw = dict(input_dim=40, num_hidden_layers=2, hidden_dim=500,
                  output_dim=100, p_dropout=0.0, type_init='xavier_uni',
                  activation_layer='sigmoid', batch_norm=bn)
        net = build_net(abnet3, 2, **kw)
        for k, v in net.state_dict().items():
            out['chk.' + k] = np.array([float(v.double().sum()),
                                        float(v.double().abs().sum())])
        batches = [make_inputs(4096, 40, 20 + s) for s in range(2)]
        opt = torch.optim.Adadelta(net.parameters(), lr=0.1)
        loss_mod = abnet3.loss.coscos2(avg=False)
        net.train()
        losses = []
        for s in range(5):
            x1, x2, y = batches[s % 2]
            e1, e2 = net(x1, x2)
            lv = loss_mod(e1, e2, torch.from_numpy(y))
            opt.zero_grad()
            lv.backward()
            if s == 0:
                out['e1_rows'] = e1.detach().numpy()[:8].copy()
                out['e2_rows'] = e2.detach().numpy()[-8:].copy()
                # every one of the 2 x 4096 embedding rows, as two float64 checksums per row (sum, sum |.|),
                # and each tensor's sum / sum |.| / max |.|
                for name, e in (('e1', e1), ('e2', e2)):
                    ed = e.detach().double()
                    out[name + '_rowsum'] = ed.sum(dim=1).numpy().copy()
                    out[name + '_rowabs'] = ed.abs().sum(dim=1).numpy().copy()
                    out[name + '_chk'] = np.array([float(ed.sum()), float(ed.abs().sum()), float(ed.abs().max())])
                for k, p in net.named_parameters():
                    g = p.grad.detach()
                    out['gchk.' + k] = np.array(
                        [float(g.double().sum()), float(g.double().abs().sum()),
                         float(g.double().abs().max())])
                    out['grow.' + k] = g.numpy().reshape(g.shape[0], -1)[:4].copy()
            opt.step()
            losses.append(float(lv.detach()))
        out['losses'] = np.array(losses)
        for k, v in net.state_dict().items():
            out['after_chk.' + k] = np.array([float(v.double().sum()),
                                              float(v.double().abs().sum())])
        out['kw'] = np.array(repr(kw))
        np.savez_compressed(
            os.path.join(OUT, 'train_c2_bn%d.npz' % int(bn)), **out)


def g5_cosdist(abnet3):
    """G5: abnet3.utils.cosine_distance (utils.py:40-60)."""
    rng = np.random.default_rng(5)
    out = {}
    cases = {}
    a = rng.standard_normal((37, 40)).astype(np.float32)
    b = rng.standard_normal((29, 40)).astype(np.float32)
    cases['f32'] = (a, b)
    cases['f64'] = (a.astype(np.float64), b.astype(np.float64))
    cases['one'] = (a[:1], b[:1])
    az = a.copy()
    bz = b.copy()
    az[3] = 0
    az[10] = 0
    bz[7] = 0
    cases['zero'] = (az, bz)
    bd = a[:29] + 1e-3 * rng.standard_normal((29, 40)).astype(np.float32)
    cases['near'] = (a, bd.astype(np.float32))
    pos = np.abs(rng.standard_normal((50, 40))).astype(np.float32)
    cases['pos'] = (pos[:30], pos[20:])
    for name, (x, y) in cases.items():
        out[name + '.x'], out[name + '.y'] = x, y
        try:
            out[name + '.d'] = abnet3.utils.cosine_distance(x, y)
        except AssertionError:
            out[name + '.d'] = np.array('AssertionError')
    np.savez_compressed(os.path.join(OUT, 'cosdist.npz'), **out)


def g6_stack(abnet3):
    """G6: FeaturesGenerator.stack_fbanks (features.py:135-159)."""
    rng = np.random.default_rng(6)
    fg = abnet3.features.FeaturesGenerator.__new__(
        abnet3.features.FeaturesGenerator)
    out = {}
    for name, (T, D, n) in {'t100_n7': (100, 40, 7), 't100_n3': (100, 40, 3),
                            't5_n7': (5, 40, 7), 't2_n7': (2, 40, 7),
                            't17_n1': (17, 13, 1)}.items():
        f = rng.standard_normal((T, D)).astype(np.float32)
        out[name + '.in'] = f
        out[name + '.n'] = np.array(n)
        out[name + '.out'] = fg.stack_fbanks(f, nframes=n)
    np.savez_compressed(os.path.join(OUT, 'stack.npz'), **out)


def g7_frames(abnet3):
    """G7: OriginalDataLoader.load_frames_from_pairs (dataloader.py:166-261)
    with a fake accessor and the build's oracle DTW standing in for the absent
    third-party dtw.DTW (parity unpinned for the DTW itself)."""
    sys.path.insert(0, REPO)
    from oracle import dtw_oracle
    rng = np.random.default_rng(7)
    feats = {'u%d' % i: rng.standard_normal((n, 40)).astype(np.float32)
             for i, n in enumerate((90, 70, 120, 64))}
    times = {k: (np.arange(len(v)) * 0.01 + 0.0025) for k, v in feats.items()}
    acc = abnet3.utils.Features_Accessor(times, feats)

    def oracle_align(f1, f2):
        d = abnet3.utils.cosine_distance(f1, f2)
        p1, p2 = dtw_oracle.dtw_path(d)
        return list(p1), list(p2)

    abnet3.dataloader.get_dtw_alignment = oracle_align
    out = {}
    for k, v in feats.items():
        out['feat.' + k] = v
    pairs = [('u0', 0.10, 0.42, 'u1', 0.05, 0.31, 'same'),
             ('u2', 0.50, 0.93, 'u3', 0.11, 0.37, 'same'),
             ('u0', 0.33, 0.61, 'u2', 0.70, 1.05, 'diff'),
             ('u1', 0.20, 0.21, 'u3', 0.30, 0.46, 'diff'),   # 1-frame token
             ('u3', 0.40, 0.30, 'u1', 0.10, 0.20, 'same'),   # s>e: skipped
             ('u1', 0.02, 0.29, 'u1', 0.31, 0.64, 'same')]
    out['pairs'] = np.array([' '.join(map(str, p)) for p in pairs])
    # an empty token (no frame time inside [0.20, 0.20]) is legal only without
    # align_different_words (the reference indexes into it otherwise)
    empty = ('u1', 0.20, 0.20, 'u3', 0.30, 0.46, 'diff')
    out['pairs_empty'] = np.array(' '.join(map(str, empty)))
    for align in (False, True):
        dl = abnet3.dataloader.OriginalDataLoader(
            'unused', 'unused', align_different_words=align)
        dl.features = acc
        grouped = abnet3.utils.group_pairs(pairs + ([] if align else [empty]))
        X1, X2, Y = dl.load_frames_from_pairs(grouped)
        out['align%d.X1' % align] = X1
        out['align%d.X2' % align] = X2
        out['align%d.Y' % align] = Y
    np.savez_compressed(os.path.join(OUT, 'frames.npz'), **out)


def g8_multitask(abnet3):
    """G8: SiameseMultitaskNetwork (model.py:211-376) + weighted_loss_multi
    (loss.py:140-182), re-enacting TrainerSiameseMultitask.give_batch_to_network
    (trainer.py:259-279) and trainer.py:236-242."""
    import torch
    cases = {
        'sig': dict(activation_layer='sigmoid', batch_norm=False, num_hidden_layers_shared=1,
                    num_hidden_layers_spk=1, num_hidden_layers_phn=0),
        'relu_bn': dict(activation_layer='relu', batch_norm=True, num_hidden_layers_shared=2,
                        num_hidden_layers_spk=0, num_hidden_layers_phn=1),
        'tanh0': dict(activation_layer='tanh', batch_norm=False, num_hidden_layers_shared=0,
                      num_hidden_layers_spk=0, num_hidden_layers_phn=0),
    }
    for name, extra in cases.items():
        kw = dict(input_dim=40, hidden_dim=64, output_dim=24, p_dropout=0.0,
                  type_init='xavier_uni')
        kw.update(extra)
        x1, x2, y_spk = make_inputs(48, 40, 3)
        np.random.seed(11)
        y_phn = np.random.choice([1, -1], 48)
        out = {'x1': x1.numpy(), 'x2': x2.numpy(), 'y_spk': y_spk, 'y_phn': y_phn,
               'kw': np.array(repr(kw))}
        torch.manual_seed(5)
        net = abnet3.model.SiameseMultitaskNetwork(**kw)
        out.update(sd_to_np(net))
        net.eval()
        with torch.no_grad():
            s1, p1, s2, p2 = net(x1, x2)
        for k, v in (('spk1', s1), ('phn1', p1), ('spk2', s2), ('phn2', p2)):
            out['eval_' + k] = v.numpy()
        for oname, weight in (('sgd', 0.3), ('adadelta', 0.5), ('adam', 1.0)):
            torch.manual_seed(5)
            net = abnet3.model.SiameseMultitaskNetwork(**kw)
            loss_mod = abnet3.loss.weighted_loss_multi(
                loss_spk=abnet3.loss.coscos2(avg=False),
                loss_phn=abnet3.loss.cosmargin(avg=True, margin=0.4), weight=weight)
            opt = OPTIMS[oname](torch, net.parameters())
            net.train()
            tag = '%s.w%g' % (oname, weight)
            losses = []
            for s in range(3):
                emb = net(x1, x2)
                lv = loss_mod(emb[0], emb[1], emb[2], emb[3], torch.from_numpy(y_spk),
                              torch.from_numpy(y_phn))
                opt.zero_grad()
                lv.backward()
                if s == 0:
                    for k, p in net.named_parameters():
                        # dead branches (hidden_layers_spk / _phn) get no gradient at all
                        out['%s.grad0.%s' % (tag, k)] = (
                            p.grad.detach().numpy().copy() if p.grad is not None
                            else np.zeros(0, dtype=np.float32))
                    for k, v in zip(('spk1', 'phn1', 'spk2', 'phn2'), emb):
                        out['%s.%s_0' % (tag, k)] = v.detach().numpy().copy()
                opt.step()
                losses.append(float(lv.detach()))
            out['%s.losses' % tag] = np.array(losses, dtype=np.float64)
            for k, v in net.state_dict().items():
                out['%s.after.%s' % (tag, k)] = v.detach().numpy().copy()
        np.savez_compressed(os.path.join(OUT, 'multitask_%s.npz' % name), **out)

    # speaker labels of the multitask loader (dataloader.py:195-201,235-241): the
    # reference compares the speaker strings with `is`
    sys.path.insert(0, REPO)
    from oracle import dtw_oracle
    rng = np.random.default_rng(8)
    feats = {'u%d' % i: rng.standard_normal((n, 40)).astype(np.float32)
             for i, n in enumerate((80, 60, 75))}
    times = {k: (np.arange(len(v)) * 0.01 + 0.0025) for k, v in feats.items()}
    acc = abnet3.utils.Features_Accessor(times, feats)

    def oracle_align(f1, f2):
        d = abnet3.utils.cosine_distance(f1, f2)
        a, b = dtw_oracle.dtw_path(d)
        return list(a), list(b)

    abnet3.dataloader.get_dtw_alignment = oracle_align
    import tempfile
    with tempfile.NamedTemporaryFile('w', suffix='.spk', delete=False) as fh:
        fh.write('u0 alice\nu1 alice\nu2 b\n')
        spk_path = fh.name
    fid2spk = abnet3.utils.read_spkid_file(spk_path)
    os.unlink(spk_path)
    pairs = [('u0', 0.10, 0.40, 'u1', 0.05, 0.33, 'same'),    # same speaker name, two files
             ('u0', 0.12, 0.30, 'u0', 0.41, 0.66, 'same'),    # same file
             ('u1', 0.20, 0.45, 'u2', 0.10, 0.31, 'diff'),
             ('u2', 0.05, 0.22, 'u2', 0.40, 0.61, 'diff')]
    out = {'pairs': np.array([' '.join(map(str, q)) for q in pairs]),
           'spk_file': np.array('u0 alice\nu1 alice\nu2 b\n')}
    for k, v in feats.items():
        out['feat.' + k] = v
    dl = abnet3.dataloader.OriginalDataLoader('unused', 'unused', align_different_words=False)
    dl.features = acc
    X1, X2, y_spk, y_phn = dl.load_frames_from_pairs(abnet3.utils.group_pairs(pairs),
                                                     fid2spk=fid2spk)
    out['X1'], out['X2'], out['y_spk'], out['y_phn'] = X1, X2, y_spk, y_phn
    np.savez_compressed(os.path.join(OUT, 'multitask_frames.npz'), **out)


NO_AVX512 = 'AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL'


def near_duplicate_pair(p):
    """Token pair p of the drop-decision set: 176 x 40 frames (M*N*K > 10^6, the
    regular sgemm kernel), the second token = the first plus noise of a size that
    puts cos(x_i, y_i) within a few ulp of 1.  Regenerated from the seed by the tests
    (numpy's PCG64 stream); `in_chk` in the fixture guards against a drifted stream."""
    rng = np.random.default_rng(5000 + p)
    n = 176
    x = rng.standard_normal((n, 40)).astype(np.float32)
    if p % 3 == 2:
        x = np.abs(x)                            # filterbank-like: all cosines > 0.5
    eps = [0.0, 3e-4, 6e-4, 8e-4, 1e-3, 1.3e-3, 2e-3, 4e-3][p % 8]
    y = (x + np.float32(eps) * rng.standard_normal((n, 40)).astype(np.float32)).astype(np.float32)
    if p % 16 >= 8:
        y = (y * np.float32(1.0 + 0.37 * (p % 5))).astype(np.float32)   # same direction, other norm
    return x, y


def g5l_cosdist_libm(abnet3):
    """G5L: cosine_distance on numpy's libm path (see the module docstring)."""
    import hashlib
    if os.environ.get('NPY_DISABLE_CPU_FEATURES') != NO_AVX512:
        import subprocess
        env = dict(os.environ, NPY_DISABLE_CPU_FEATURES=NO_AVX512)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), '--only', 'G5L'], env=env)
        return
    rng = np.random.default_rng(55)
    out = {}
    cases = {}
    cases['big'] = (rng.standard_normal((300, 40)).astype(np.float32),
                    rng.standard_normal((280, 40)).astype(np.float32))
    pos = np.abs(rng.standard_normal((530, 40))).astype(np.float32)
    cases['bigpos'] = (pos[:260], pos[260:])
    z = rng.standard_normal((290, 40)).astype(np.float32)
    w = rng.standard_normal((270, 40)).astype(np.float32)
    z[5] = 0; z[77] = 0; w[0] = 0; w[200] = 0
    cases['bigzero'] = (z, w)
    z2, w2 = z.copy(), w.copy()
    w2[13] = -2.0 * z2[13]                          # cos = -1 up to rounding
    w2[14] = 0.5 * z2[14]                           # cos = +1 up to rounding
    cases['bigdup'] = (z2, w2)
    wide = rng.standard_normal((190, 280)).astype(np.float32)   # stacked 7 x 40 features
    cases['wide'] = (wide[:100], wide[100:])
    for name, (x, y) in cases.items():
        out[name + '.x'], out[name + '.y'] = x, y
        try:
            d = abnet3.utils.cosine_distance(x, y)
            d32 = d.astype(np.float32)
            assert (d32.astype(np.float64) == d).all()      # float32 values in a float64 container
            out[name + '.d32'] = d32
        except AssertionError:
            out[name + '.d32'] = np.zeros((0, 0), dtype=np.float32)
            out[name + '.dropped'] = np.array(True)
    P = 48
    dropped, hashes, in_chk = [], [], []
    for p in range(P):
        x, y = near_duplicate_pair(p)
        in_chk.append([float(x.astype(np.float64).sum()), float(y.astype(np.float64).sum())])
        try:
            d = abnet3.utils.cosine_distance(x, y)
            dropped.append(False)
            hashes.append(hashlib.sha256(d.astype(np.float32).tobytes()).hexdigest())
        except AssertionError:
            dropped.append(True)
            hashes.append('')
    out['near.dropped'] = np.array(dropped)
    out['near.sha256'] = np.array(hashes)
    out['near.in_chk'] = np.array(in_chk)
    print('   near-duplicate pairs dropped by the reference: %d of %d' % (sum(dropped), P))
    np.savez_compressed(os.path.join(OUT, 'cosdist_libm.npz'), **out)


def g9_frames_loader(abnet3):
    """G9: FramesDataLoader (dataloader.py:580-739) and the temporal coherence pairs
    of OriginalDataLoader (:314-352), with a fake accessor and the build's oracle DTW
    standing in for the absent third-party dtw.DTW (as in G7)."""
    import random
    sys.path.insert(0, REPO)
    from oracle import dtw_oracle
    rng = np.random.default_rng(9)
    feats = {'u%d' % i: rng.standard_normal((n, 40)).astype(np.float32)
             for i, n in enumerate((90, 70, 120, 64, 100))}
    times = {k: (np.arange(len(v)) * 0.01 + 0.0025) for k, v in feats.items()}

    def oracle_align(f1, f2):
        d = abnet3.utils.cosine_distance(f1, f2)
        p1, p2 = dtw_oracle.dtw_path(d)
        return list(p1), list(p2)

    abnet3.dataloader.get_dtw_alignment = oracle_align
    train = [('u0', 0.10, 0.42, 'u1', 0.05, 0.31, 'same'),
             ('u2', 0.50, 0.93, 'u3', 0.11, 0.37, 'same'),
             ('u0', 0.33, 0.61, 'u2', 0.70, 1.05, 'diff'),
             ('u1', 0.20, 0.21, 'u3', 0.30, 0.46, 'diff'),   # 1-frame token
             ('u3', 0.40, 0.30, 'u1', 0.10, 0.20, 'same'),   # s > e: skipped
             ('u1', 0.02, 0.29, 'u4', 0.31, 0.64, 'same'),
             ('u4', 0.05, 0.44, 'u2', 0.15, 0.50, 'same'),
             ('u4', 0.60, 0.95, 'u0', 0.45, 0.85, 'diff'),
             ('u3', 0.02, 0.30, 'u4', 0.50, 0.80, 'diff')]
    dev = [('u0', 0.50, 0.80, 'u3', 0.20, 0.52, 'same'),
           ('u1', 0.30, 0.55, 'u2', 0.20, 0.41, 'diff'),
           ('u2', 0.02, 0.33, 'u4', 0.20, 0.49, 'same')]
    out = {'train_pairs': np.array([' '.join(map(str, p)) for p in train]),
           'dev_pairs': np.array([' '.join(map(str, p)) for p in dev])}
    for k, v in feats.items():
        out['feat.' + k] = v

    cases = {   # name: (ctor kwargs, numpy seed, sequence of epochs: T(rain) / D(ev))
        'full': (dict(batch_size=50, randomize_dataset=True), 9, 'TTDT'),
        'norand': (dict(batch_size=64, randomize_dataset=False), 10, 'TDT'),
        'sub': (dict(batch_size=40, randomize_dataset=True, max_batches_per_epoch=2), 11, 'TTTDTTT'),
        'tiny': (dict(batch_size=5000, randomize_dataset=True), 12, 'TD'),   # fewer pairs than one batch
    }
    for name, (kw, seed, epochs) in cases.items():
        acc = abnet3.utils.Features_Accessor(dict(times), {k: v.copy() for k, v in feats.items()})
        dl = abnet3.dataloader.FramesDataLoader('unused', 'unused', **kw)
        dl.features = acc
        dl.pairs = {'train': list(train), 'dev': list(dev)}
        np.random.seed(seed)
        X1s, X2s, Ys, sizes, nb = [], [], [], [], []
        for mode in epochs:
            n = 0
            for X1, X2, y in dl.batch_iterator(train_mode=(mode == 'T')):
                X1s.append(X1.numpy()); X2s.append(X2.numpy()); Ys.append(y.numpy())
                sizes.append(len(y))
                n += 1
            nb.append(n)
        out[name + '.kw'] = np.array(repr(kw))
        out[name + '.seed'] = np.array(seed)
        out[name + '.epochs'] = np.array(epochs)
        out[name + '.X1'] = np.vstack(X1s)
        out[name + '.X2'] = np.vstack(X2s)
        out[name + '.y'] = np.concatenate(Ys)
        out[name + '.sizes'] = np.array(sizes)
        out[name + '.batches_per_epoch'] = np.array(nb)
        out[name + '.n_frame_pairs'] = np.array([len(dl.frame_pairs['train']), len(dl.frame_pairs['dev'])])

    # temporal coherence pairs added to a word-pair batch (tcl = 0.3)
    acc = abnet3.utils.Features_Accessor(dict(times), {k: v.copy() for k, v in feats.items()})
    dl = abnet3.dataloader.OriginalDataLoader('unused', 'unused', tcl=0.3)
    dl.features = acc
    dl.train_files = ['u0', 'u2', 'u3', 'u4']       # explicit order (the reference builds it from a set)
    batch = dl.load_frames_from_pairs(abnet3.utils.group_pairs(train[:4]))
    random.seed(4)
    X1, X2, Y = dl.add_tcl_to_batch(batch)
    out['tcl.train_files'] = np.array(dl.train_files)
    out['tcl.n_before'] = np.array(len(batch[2]))
    out['tcl.X1'], out['tcl.X2'], out['tcl.Y'] = X1, X2, Y
    np.savez_compressed(os.path.join(OUT, 'frames_loader.npz'), **out)


def g10_mvn(abnet3):
    """G10: mean / variance normalisation as the reference's test states it (test/test_features.py:37-281).
    h5features / h5py are absent, so FeaturesGenerator.mean_variance_normalisation itself cannot run; what the
    reference's test pins is re-enacted: its literal inputs, the frames a VAD keeps (the reference's own
    read_vad_file + filter_vad_one_file / filter_vad_whole_dataset on its own Features_Accessor), the statistics it
    asserts (np.mean / np.std of the kept frames, axis 0 or None) and the output it asserts
    ((feature - mean) / (std + eps), abnet3/features.py:240,293)."""
    import tempfile
    from abnet3.features import FeaturesGenerator
    from abnet3.utils import read_vad_file, Features_Accessor
    out = {}
    t = lambda n: np.arange(n, dtype=float) * 0.01 + 0.0025
    vad_text = "file,start,stop\nfile1,0.0025,0.5000\nfile1,0.7525,1.000\n"
    with tempfile.NamedTemporaryFile('w', suffix='.vad', delete=False) as fh:
        fh.write(vad_text)
    vad = read_vad_file(fh.name)
    os.unlink(fh.name)
    cases = {
        # test_normalization (:37-85)
        'global': ([np.full((100, 40), 1.0), np.full((150, 40), 2.0)], False, None),
        # test_normalization_per_file (:87-140)
        'per_file': ([np.vstack([np.full((100, 40), 1.), np.full((100, 40), -1.)]),
                      np.vstack([np.full((100, 40), 1.), np.full((100, 40), 2.)])], True, None),
        # test_normalization_with_VAD (:142-204)
        'global_vad': ([np.vstack([np.full((50, 40), 1.0), np.full((50, 40), -1.0)]),
                        np.vstack([np.full((50, 40), 1.0), np.full((50, 40), -1.0)])], False, vad),
        # test_norm_per_file_with_VAD (:206-281)
        'per_file_vad': ([np.vstack([np.full((50, 40), 1.0), np.full((50, 40), -1.0)]),
                          np.vstack([np.full((50, 40), 1.0), np.full((50, 40), -1.0)])], True, vad),
    }
    items = ['file1', 'file2']
    fg = FeaturesGenerator()
    for name, (features, per_file, vad_data) in cases.items():
        times = [t(f.shape[0]) for f in features]
        for i, f in enumerate(features):
            out['%s.x%d' % (name, i)] = f.astype(np.float32)
            out['%s.t%d' % (name, i)] = times[i]
        out[name + '.per_file'] = np.array(int(per_file))
        out[name + '.vad'] = np.array(vad_text if vad_data is not None else '')
        eps = np.finfo(features[0].dtype).eps
        for per_channel in (1, 0):
            axis = 0 if per_channel else None
            tag = '%s.pc%d' % (name, per_channel)
            if per_file:
                for i, (f, tm) in enumerate(zip(features, times)):
                    kept = f
                    if vad_data is not None and items[i] in vad_data:
                        kept = fg.filter_vad_one_file(f, tm, vad_data[items[i]])
                    mean, std = np.mean(kept, axis=axis), np.std(kept, axis=axis)
                    out['%s.kept%d' % (tag, i)] = np.array(kept.shape[0])
                    out['%s.mean%d' % (tag, i)] = np.atleast_1d(mean)
                    out['%s.std%d' % (tag, i)] = np.atleast_1d(std)
                    out['%s.out%d' % (tag, i)] = (f - mean) / (std + eps)
            else:
                acc = Features_Accessor({k: tm for k, tm in zip(items, times)},
                                        {k: f.astype(np.float32) for k, f in zip(items, features)})
                if vad_data is not None:
                    fg.filter_vad_whole_dataset(acc, vad_data)
                kept = np.vstack(list(acc.features.values())).astype(np.float64)
                mean, std = np.mean(kept, axis=axis), np.std(kept, axis=axis)
                out[tag + '.kept'] = np.array(kept.shape[0])
                out[tag + '.mean'] = np.atleast_1d(mean)
                out[tag + '.std'] = np.atleast_1d(std)
                for i, f in enumerate(features):
                    out['%s.out%d' % (tag, i)] = (f - mean) / (std + eps)
    # the literals the reference's test spells out
    assert int(out['global_vad.pc1.kept']) == 75 + 100 and int(out['per_file_vad.pc1.kept0']) == 75
    assert np.allclose(out['per_file.pc0.mean1'], 1.5) and np.allclose(out['per_file.pc0.mean0'], 0.0)
    np.savez_compressed(os.path.join(OUT, 'mvn.npz'), **out)


def g11_gridsearch(abnet3):
    """G11: the caller's side of the boundary.  GridSearch.run_single_experiment (gridsearch.py:145-202) builds every
    object as getattr(abnet3.<module>, cfg['class'])(**cfg['arguments']) after injecting a few arguments; the
    fixture holds the reference's own test configuration (test/data/buckeye.yaml -> default_params, as data) and
    what the reference's classes are when built from it on this machine (sampler excluded: out of scope; the
    reference's trainer needs tensorboardX's SummaryWriter only inside train())."""
    import inspect
    import json
    import yaml
    import abnet3.trainer
    import abnet3.embedder
    with open(os.path.join(REF, 'test', 'data', 'buckeye.yaml')) as fh:
        params = yaml.safe_load(fh)['default_params']
    exp = '/tmp/abnet3_g11'
    info = {}

    def build(module, section, inject):
        cfg = params[section]
        cls = getattr(module, cfg['class'])
        args = dict(cfg['arguments'] or {})
        args.update(inject)
        accepted = set(inspect.signature(cls.__init__).parameters) - {'self'}
        has_var_kw = any(p.kind == p.VAR_KEYWORD for p in inspect.signature(cls.__init__).parameters.values())
        rejected = sorted(k for k in args if k not in accepted and not has_var_kw)
        info[section + '.rejected_kwargs'] = rejected
        return cls(**{k: v for k, v in args.items() if k not in rejected})

    features = build(abnet3.features, 'features', {})
    model = build(abnet3.model, 'model', {'output_path': os.path.join(exp, 'network')})
    loss = build(abnet3.loss, 'loss', {})
    dataloader = build(abnet3.dataloader, 'dataloader',
                       {'pairs_path': os.path.join(exp, 'pairs'), 'features_path': features.output_path})
    trainer = build(abnet3.trainer, 'trainer', {'network': model, 'loss': loss, 'dataloader': dataloader,
                                                'log_dir': os.path.join(exp, 'logs')})
    embedder = build(abnet3.embedder, 'embedder', {'network': model, 'output_path': os.path.join(exp, 'embeddings.h5f'),
                                                   'feature_path': features.output_path,
                                                   'network_path': model.output_path + '.pth'})
    info['model.n_parameters'] = int(sum(p.numel() for p in model.parameters()))
    info['model.state_dict_keys'] = list(model.state_dict().keys())
    info['model.whoami_keys'] = sorted(model.whoami().keys())
    info['loss.class'] = type(loss).__name__
    info['loss.whoami_keys'], info['loss.avg'] = sorted(loss.whoami().keys()), bool(loss.avg)
    info['trainer.optimizer'] = type(trainer.optimizer).__name__
    info['trainer.lr'] = trainer.optimizer.param_groups[0]['lr']
    info['trainer.num_epochs'], info['trainer.patience'] = trainer.num_epochs, trainer.patience
    info['dataloader.batch_size'] = dataloader.batch_size
    info['dataloader.num_max_minibatches'] = dataloader.num_max_minibatches
    info['embedder.batch_size'] = embedder.batch_size
    info['features.run'] = features.run
    with open(os.path.join(OUT, 'gridsearch_buckeye.json'), 'w') as fh:
        json.dump({'default_params': params, 'reference': info}, fh, indent=1, sort_keys=True)


ALL = {'G1': g1_tower, 'G2': g2_train_c1, 'G3': g3_loss_edge,
       'G4': g4_train_mid, 'G5': g5_cosdist, 'G6': g6_stack, 'G7': g7_frames,
       'G8': g8_multitask, 'G5L': g5l_cosdist_libm, 'G9': g9_frames_loader, 'G10': g10_mvn, 'G11': g11_gridsearch}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='')
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    abnet3 = import_reference()
    import torch
    torch.set_num_threads(1)      # deterministic summation order in fixtures
    which = [s for s in args.only.split(',') if s] or list(ALL)
    for g in which:
        print('generating', g, flush=True)
        ALL[g](abnet3)
    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print('golden dir: %.1f KB' % (total / 1024))


if __name__ == '__main__':
    main()
