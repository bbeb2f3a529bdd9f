"""BASELINE.json configs[4] ("C5") on one GPU and the pieces it is made of: the batched feature launches
(abn_fbank_batched, abn_stack_frames_batched), batch plans (the loaders' batches as index lists in HBM,
abn_gather_pairs), planned passes of the trainer (bucketed captured steps with a device-side count of real
pairs), and the whole pipeline on a synthetic ZeroSpeech-shaped corpus with property checks.
Needs an MI355X: run with -m gpu."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.fixture(scope='module')
def small_corpus():
    from tools.c5_corpus import sample_pairs, synth_corpus
    corpus = synth_corpus(n_utts=60, n_types=40, seed=3, device='cuda')
    train_pairs, dev_pairs = sample_pairs(corpus, n_pairs=600, seed=3)
    return corpus, train_pairs, dev_pairs


def test_batched_feature_launches_equal_the_per_utterance_calls(small_corpus):
    """One launch for the corpus = the per-file loop of the reference (features.py:160-203, :299-320): every
    utterance framed, normalised and stacked as if it were alone -- bit for bit."""
    from abnet3_amd.features import FeaturesGenerator
    corpus = small_corpus[0]
    fg = FeaturesGenerator(normalization=False, stack=False)
    waves = corpus.waves[:17] + [corpus.waves[17][:150], corpus.waves[18][:1], corpus.waves[19][:401]]     # short tails too
    table, nfr = fg.fbank_batch(waves, corpus.fs)
    o = 0
    for w, n in zip(waves, nfr):
        one = fg.fbank_from_samples(w, corpus.fs)
        assert one.shape[0] == n
        assert torch.equal(table[o:o + n], one)
        o += int(n)
    assert o == table.shape[0]
    stacked = fg.stack_table(table, nfr, nframes=7)
    o = 0
    for n in nfr:
        assert torch.equal(stacked[o:o + n], fg.stack_fbanks(table[o:o + n].contiguous(), nframes=7))
        o += int(n)
    # deltas never cross an utterance boundary either
    fgd = FeaturesGenerator(normalization=False, stack=False, deltas=True, deltasdeltas=True)
    td, _ = fgd.fbank_batch(waves[:5], corpus.fs)
    o = 0
    for w, n in zip(waves[:5], nfr[:5]):
        assert torch.equal(td[o:o + n], fgd.fbank_from_samples(w, corpus.fs))
        o += int(n)


def test_fbank_framing_of_the_lineage_against_the_oracle():
    """The framing quirks of the Sphinx-III lineage (oracle/features_np.py): a frame's pre-emphasis starts from the
    last sample of the previous frame, the tail frames repeat their samples cyclically -- including an empty
    last frame, a signal shorter than one window, and float input."""
    from abnet3_amd.features import FeaturesGenerator
    from oracle import features_np
    rng = np.random.default_rng(5)
    fg = FeaturesGenerator()
    for n in (16000, 1600, 401, 400, 399, 160, 161, 1, 4799, 4800):
        sig = (3000 * np.sin(np.arange(n) * 0.05) + 500 * rng.standard_normal(n)).astype(np.int16)
        got = fg.fbank_from_samples(sig, 16000).cpu().numpy()
        ref = features_np.fbank(sig, 16000)
        assert got.shape == ref.shape, n
        assert np.abs(got - ref).max() < 5e-5, (n, np.abs(got - ref).max())
    sig = rng.standard_normal(5000).astype(np.float32)
    assert np.abs(fg.fbank_from_samples(sig, 16000).cpu().numpy() - features_np.fbank(sig, 16000)).max() < 5e-5
    # the general (any nfft) kernel frames the same way
    got = fg.fbank_from_samples((1000 * sig).astype(np.int16), 16000, nfft=512).cpu().numpy()
    assert np.abs(got - features_np.fbank((1000 * sig).astype(np.int16), 16000, nfft=512)).max() < 5e-5


def test_gather_pairs_against_indexing():
    from abnet3_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(0)
    for D in (280, 40, 7):
        table = torch.from_numpy(rng.standard_normal((500, D)).astype(np.float32)).cuda()
        i1 = torch.from_numpy(rng.integers(0, 500, 1000)).cuda()
        i2 = torch.from_numpy(rng.integers(0, 500, 1000)).cuda()
        for ydt in (torch.float64, torch.int64):
            y = torch.from_numpy(rng.choice([1, -1], 1000)).cuda().to(ydt)
            for first, n, npad in ((0, 100, 128), (37, 333, 352), (990, 10, 10), (5, 0, 32)):
                x12 = torch.full((2 * npad, D), 7.0, device='cuda')
                yo = torch.full((npad,), 3, dtype=ydt, device='cuda')
                nv = torch.zeros(1, dtype=torch.int32, device='cuda')
                _lib.check(lib.abn_gather_pairs(_lib.ptr(table), table.shape[0], D, _lib.ptr(i1), _lib.ptr(i2), first, n, npad, _lib.ptr(y), 8,
                                                _lib.ptr(x12), _lib.ptr(yo), _lib.ptr(nv), _lib.stream()), 'gather_pairs')
                assert int(nv) == n
                assert torch.equal(x12[:n], table[i1[first:first + n]]) and torch.equal(x12[npad:npad + n], table[i2[first:first + n]])
                assert float(x12[n:npad].abs().sum()) == 0.0 and float(x12[npad + n:].abs().sum()) == 0.0
                assert torch.equal(yo[:n], y[first:first + n]) and float(yo[n:].abs().sum()) == 0.0


def _loader(kind, small_corpus, **kw):
    from tools import c5_pipeline
    corpus, train_pairs, dev_pairs = small_corpus
    stage = {}
    dc, _ = c5_pipeline.build_features(corpus, stage)
    from abnet3_amd.dataloader import FramesDataLoader, OriginalDataLoader
    if kind == 'original':
        dl = OriginalDataLoader('unused', 'unused', num_max_minibatches=kw.pop('num_max_minibatches', 10000), seed=0, batch_size=8, **kw)
    else:
        dl = FramesDataLoader('unused', 'unused', batch_size=kw.pop('batch_size', 512), **kw)
    dl.features = dc
    dl.pairs['train'], dl.pairs['dev'] = list(train_pairs), list(dev_pairs)
    dl.train_files = list({p[0] for p in train_pairs})
    return dl


@pytest.mark.parametrize('kw', [{}, {'align_different_words': True}, {'num_max_minibatches': 20}, {'shuffle_between_epochs': True}])
def test_plan_yields_the_iterators_batches(small_corpus, kw):
    """OriginalDataLoader.plan() = batch_iterator(): the same batches in the same order, bit for bit, the same
    draws from numpy's and random's global generators (what the NEXT epoch selects depends on them), the same
    statistics."""
    import random
    for train_mode in (True, False):
        a, b = _loader('original', small_corpus, **dict(kw)), _loader('original', small_corpus, **dict(kw))
        np.random.seed(11); random.seed(11)
        it = [tuple(t.clone() for t in batch) for batch in a.batch_iterator(train_mode)]
        state_a, rstate_a = np.random.get_state()[1].copy(), random.getstate()
        np.random.seed(11); random.seed(11)
        plan = b.plan(train_mode)
        state_b, rstate_b = np.random.get_state()[1].copy(), random.getstate()
        got = list(plan)
        assert len(got) == len(it) > 0
        for x, y in zip(it, got):
            for u, v in zip(x, y):
                assert u.dtype == v.dtype and torch.equal(u, v)
        assert np.array_equal(state_a, state_b) and rstate_a == rstate_b
        assert dict(a.statistics_training) == dict(b.statistics_training)
        # a second pass draws on from there
        full, rfull = np.random.get_state(), random.getstate()
        it2 = [b_[2].clone() for b_ in a.batch_iterator(train_mode)]
        np.random.set_state(full); random.setstate(rfull)
        got2 = [b_[2] for b_ in b.plan(train_mode)]
        assert len(it2) == len(got2) and all(torch.equal(u, v) for u, v in zip(it2, got2))


def test_frames_plan_yields_the_iterators_batches(small_corpus):
    for train_mode in (True, False):
        a, b = _loader('frames', small_corpus), _loader('frames', small_corpus)
        np.random.seed(5)
        it = [tuple(t.clone() for t in batch) for batch in a.batch_iterator(train_mode)]
        np.random.seed(5)
        got = list(b.plan(train_mode))
        assert len(got) == len(it) > 0
        for x, y in zip(it, got):
            for u, v in zip(x, y):
                assert torch.equal(u, v)


@pytest.mark.parametrize('opt,bn', [('adadelta', False), ('adam', False), ('adadelta', True)])
def test_planned_passes_train_like_the_iterator(small_corpus, opt, bn, tmp_path):
    """Two epochs over the reference's canonical loader (8 word pairs a batch: a different number of frame pairs
    every step), once through planned passes (gather launch + captured step per bucket, padded rows masked by
    the device-side count) and once through the plain iterator with eager steps: the same losses and the same
    parameters up to fp32 summation order (the padding moves tower 2's rows, i.e. the order in which the weight
    gradients are summed).  With BatchNorm the captured step is told the real-row count (abn_tower_desc.n_valid: batch and
    running statistics, the backward's sums and n span the real rows only; the padded rows get no gradient), the untrained
    first pass -- train mode without gradients -- takes the plan's batches one by one: running statistics and
    num_batches_tracked come out like the iterator's too."""
    from abnet3_amd.loss import coscos2
    from abnet3_amd.model import SiameseNetwork
    from abnet3_amd.trainer import TrainerSiamese
    res = []
    for planned in (True, False):
        dl = _loader('original', small_corpus)
        np.random.seed(0)
        torch.manual_seed(0)
        net = SiameseNetwork(input_dim=280, num_hidden_layers=1, hidden_dim=128, output_dim=32, p_dropout=0.0, batch_norm=bn,
                             activation_layer='sigmoid', output_path=str(tmp_path / ('net%d' % planned)))
        tr = TrainerSiamese(network=net, loss=coscos2(avg=False), num_epochs=2, patience=5, optimizer_type=opt, lr=0.1 if opt == 'adadelta' else 1e-3,
                            dataloader=dl, log_dir=str(tmp_path / 'runs'))
        tr.planned_passes = planned
        tr.train()
        if planned:
            assert any(v['graph'] is not None for v in tr._buckets.values())       # captured steps did run
            replayed = sum(1 for v in tr._buckets.values() if v['graph'] is not None)
            assert replayed >= 1
        res.append((list(tr.train_losses), list(tr.dev_losses), {k: v.detach().cpu().numpy().copy() for k, v in net.state_dict().items()}))
    (tl_a, dl_a, p_a), (tl_b, dl_b, p_b) = res
    assert np.allclose(tl_a, tl_b, rtol=2e-5) and np.allclose(dl_a, dl_b, rtol=2e-5), (tl_a, tl_b)
    assert tl_a[-1] < tl_a[0]
    tol = 2e-4 if opt == 'adam' else 2e-5
    for k in p_a:
        if k.endswith('num_batches_tracked'):
            assert p_a[k] == p_b[k] > 0, k
        else:
            assert rel_err(p_a[k], p_b[k]) < tol, (k, rel_err(p_a[k], p_b[k]))


def test_a_batch_without_frames_raises_like_the_reference(small_corpus, tmp_path):
    """Eight word pairs that are all skipped (start > end, abnet3/dataloader.py:179,210): the reference's np.vstack([])
    raises ValueError from the iterator (:247); so do the plan's materialise() and the planned pass."""
    from abnet3_amd.loss import coscos2
    from abnet3_amd.model import SiameseNetwork
    from abnet3_amd.trainer import TrainerSiamese
    dl = _loader('original', small_corpus)
    name = small_corpus[0].names[0]
    dl.pairs['train'] = [(name, 2.0, 1.0, name, 3.0, 2.5, 'diff')] * 8 + dl.pairs['train'][:8]
    np.random.seed(0)
    with pytest.raises(ValueError):
        list(dl.batch_iterator(True))
    np.random.seed(0)
    with pytest.raises(ValueError):
        list(dl.plan(True))
    torch.manual_seed(0)
    net = SiameseNetwork(input_dim=280, num_hidden_layers=0, hidden_dim=64, output_dim=32, p_dropout=0.0, activation_layer='sigmoid',
                         output_path=str(tmp_path / 'net'))
    tr = TrainerSiamese(network=net, loss=coscos2(avg=False), num_epochs=1, optimizer_type='sgd', lr=0.01, dataloader=dl,
                        log_dir=str(tmp_path / 'runs'))
    tr.train_losses, tr.dev_losses = [], []
    with pytest.raises(ValueError):
        tr.optimize_model(do_training=True)


def test_padded_step_masks_the_padding():
    """abn_tower_backward_loss with n_valid: a batch padded with zero rows gives the loss and the gradients of
    the unpadded batch (mean loss: divided by the number of real pairs), and the loss accumulator adds up."""
    from abnet3_amd.loss import coscos2
    from abnet3_amd.model import SiameseNetwork
    torch.manual_seed(1)
    net = SiameseNetwork(input_dim=40, num_hidden_layers=1, hidden_dim=96, output_dim=32, p_dropout=0.0, activation_layer='tanh').cuda()
    net.train()
    rng = np.random.default_rng(0)
    n, npad = 150, 192
    x1, x2 = [torch.from_numpy(rng.standard_normal((n, 40)).astype(np.float32)).cuda() for _ in range(2)]
    y = torch.from_numpy(rng.choice([1.0, -1.0], n)).cuda()
    for avg in (False, True):
        emb, st = net.direct_forward(x1, x2)
        loss_a = net.direct_backward_loss(st, y, 'coscos2', 0.0, avg)
        ga = {k: p.grad.clone() for k, p in net.named_parameters()}
        xp1, xp2 = torch.zeros(npad, 40, device='cuda'), torch.zeros(npad, 40, device='cuda')
        xp1[:n], xp2[:n] = x1, x2
        yp = torch.zeros(npad, dtype=torch.float64, device='cuda')
        yp[:n] = y
        yp[n:] = 1.0                                                   # a label on a padded row must not count either
        nv = torch.tensor([n], dtype=torch.int32, device='cuda')
        acc = torch.full((), 2.5, dtype=torch.float64, device='cuda')
        emb, st = net.direct_forward(xp1, xp2)
        loss_b = net.direct_backward_loss(st, yp, 'coscos2', 0.0, avg, n_valid=nv, loss_accum=acc)
        assert abs(float(loss_a) - float(loss_b)) <= 1e-6 * abs(float(loss_a))
        assert abs(float(acc) - 2.5 - float(loss_b)) < 1e-6
        for k, p in net.named_parameters():
            assert rel_err(p.grad.cpu().numpy(), ga[k].cpu().numpy()) < 1e-5, k


def test_c5_pipeline_on_a_corpus_of_its_shape(tmp_path):
    """fbank -> normalise -> 7-frame stack -> DTW mining -> training through both loaders -> embedding on a
    ZeroSpeech-shaped synthetic corpus (2-10 s utterances, Zipfian word types, 8 word pairs per batch) at a size
    that keeps the test under a minute.  Properties: every same-pair path is a valid DTW path of its two tokens,
    a batch is the gather of its paths, the losses go down, the embeddings are finite."""
    from tools import c5_pipeline
    out, kept, dc, corpus, (train_pairs, dev_pairs) = c5_pipeline.run(n_utts=400, n_pairs=8000, epochs=2, out_dir=str(tmp_path), keep=True)
    assert out['corpus']['feature_dim'] == 280 and out['corpus']['utterances'] == 400
    tr, dl, emb = kept['original']
    # (1) paths: monotone, steps of 0 / 1, from the tokens' first to their last frames
    n_checked = 0
    for key, al in list(dl._align.items())[:400]:
        if al is None:
            continue
        f1, s1, e1, f2, s2, e2, _ = key
        (a0, n1), (b0, n2) = dc.token(f1, s1, e1), dc.token(f2, s2, e2)
        p1, p2 = al[0].cpu().numpy() - a0, al[1].cpu().numpy() - b0
        assert p1[0] == 0 and p2[0] == 0 and p1[-1] == n1 - 1 and p2[-1] == n2 - 1
        d1, d2 = np.diff(p1), np.diff(p2)
        assert ((d1 == 0) | (d1 == 1)).all() and ((d2 == 0) | (d2 == 1)).all() and ((d1 + d2) >= 1).all()
        assert max(n1, n2) <= len(p1) <= n1 + n2 - 1
        n_checked += 1
    assert n_checked > 100
    # (2) a batch = the gather of its pairs' paths (same pairs first, then the diff pairs cut to the shorter token)
    plan = dl.plan(True)
    for bid in plan.order[:20]:
        X1, X2, y = plan.materialise(bid)
        chunk = train_pairs[bid * 8:(bid + 1) * 8]
        rows1, rows2, lab = [], [], []
        for f1, s1, e1, f2, s2, e2, kind in [p for p in chunk if p[6] == 'same'] + [p for p in chunk if p[6] == 'diff']:
            (a0, n1), (b0, n2) = dc.token(f1, s1, e1), dc.token(f2, s2, e2)
            if kind == 'same':
                al = dl._align[(f1, s1, e1, f2, s2, e2, False)]
                if al is None:
                    continue
                rows1.append(al[0]); rows2.append(al[1]); lab += [1.0] * len(al[0])
            else:
                m = min(n1, n2)
                rows1.append(torch.arange(a0, a0 + m, device='cuda')); rows2.append(torch.arange(b0, b0 + m, device='cuda')); lab += [-1.0] * m
        perm = torch.from_numpy(np.random.RandomState(0).permutation(len(lab))).cuda()
        assert torch.equal(X1, dc.table[torch.cat(rows1)[perm]]) and torch.equal(X2, dc.table[torch.cat(rows2)[perm]])
        assert torch.equal(y.cpu(), torch.tensor(lab, dtype=torch.float64)[perm.cpu()])
    # (3) training did something, on both loaders; (4) embeddings
    for kind in ('original', 'frames'):
        st = out['training'][kind]
        assert len(st['train_losses']) == 3 and np.isfinite(st['train_losses']).all() and np.isfinite(st['dev_losses']).all()
        assert st['train_losses'][-1] < st['train_losses'][0], st['train_losses']
        assert st['embeddings_finite'] and st['train_frame_pairs_per_s'] > 0
        assert kept[kind][2].shape == (dc.total, 100)
    assert out['training']['original']['mean_frame_pairs_per_batch'] > 100


def test_planned_passes_fall_back_batch_by_batch(small_corpus, tmp_path, monkeypatch):
    """A network the padded step does not take (exact-fp32 arithmetic): the planned pass steps every batch through the
    iterator's path -- same batches, same order -- and later passes do not ask again."""
    from abnet3_amd.loss import coscos2
    from abnet3_amd.model import SiameseNetwork
    from abnet3_amd.trainer import TrainerSiamese
    res = []
    for planned in (True, False):
        dl = _loader('original', small_corpus)
        np.random.seed(0)
        torch.manual_seed(0)
        net = SiameseNetwork(input_dim=280, num_hidden_layers=0, hidden_dim=64, output_dim=32, p_dropout=0.0, activation_layer='sigmoid',
                             output_path=str(tmp_path / ('n%d' % planned)))
        net.precision = 'fp32'
        tr = TrainerSiamese(network=net, loss=coscos2(avg=True), num_epochs=1, patience=5, optimizer_type='sgd', lr=0.01, dataloader=dl,
                            log_dir=str(tmp_path / 'runs'))
        tr.planned_passes = planned
        tr.train()
        if planned:
            assert tr._plan_refused and not any(v['graph'] for v in tr._buckets.values())
        res.append((list(tr.train_losses), {k: v.detach().cpu().numpy().copy() for k, v in net.state_dict().items()}))
    assert np.allclose(res[0][0], res[1][0], rtol=1e-6)
    for k in res[0][1]:
        assert np.array_equal(res[0][1][k], res[1][1][k]), k


def test_planned_steps_redraw_their_dropout(small_corpus, tmp_path):
    """p_dropout > 0 in planned passes: the per-forward seed is drawn inside the captured step, so every replay
    drops other units (two replays of one batch with a zero learning rate give different losses); an evaluation
    in eval mode does not."""
    from abnet3_amd.loss import coscos2
    from abnet3_amd.model import SiameseNetwork
    from abnet3_amd.trainer import TrainerSiamese
    dl = _loader('original', small_corpus)
    np.random.seed(0)
    torch.manual_seed(0)
    net = SiameseNetwork(input_dim=280, num_hidden_layers=1, hidden_dim=128, output_dim=32, p_dropout=0.3, activation_layer='sigmoid',
                         output_path=str(tmp_path / 'net'))
    tr = TrainerSiamese(network=net, loss=coscos2(avg=False), num_epochs=1, optimizer_type='sgd', lr=0.0, momentum=0.0, dataloader=dl,
                        log_dir=str(tmp_path / 'runs'))
    plan = dl.plan(True)
    bid = plan.order[0]
    net.train()
    tr._bucket_state(tr._bucket(plan.span(bid)[1]), plan)
    losses = []
    for _ in range(4):                      # eager first, then replays of the captured step
        tr._loss_acc.zero_()
        assert tr._planned_step(plan, bid)
        losses.append(float(tr._loss_acc))
    assert any(v['graph'] is not None for v in tr._buckets.values())
    assert len(set(losses)) == 4, losses
    net.eval()
    ev = []
    for _ in range(3):
        tr._loss_acc.zero_()
        tr._planned_eval(plan, bid)
        ev.append(float(tr._loss_acc))
    assert ev[0] == ev[1] == ev[2]
