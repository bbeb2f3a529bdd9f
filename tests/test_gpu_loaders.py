"""FramesDataLoader / temporal-coherence pairs against fixtures produced by the
REFERENCE loaders (tests/golden/frames_loader.npz, tools/make_golden.py g9): frame-pair
list, every shuffle, batch slicing, max_batches_per_epoch wrap-around -- X1 / X2 / y of
every batch, bit for bit.  Needs an MI355X: run with -m gpu."""
import ast
import random

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def parse(line):
    t = str(line).split(' ')
    return (t[0], float(t[1]), float(t[2]), t[3], float(t[4]), float(t[5]), t[6])


def corpus(g):
    feats = {k[5:]: v for k, v in g.items() if k.startswith('feat.')}
    times = {k: np.arange(len(v)) * 0.01 + 0.0025 for k, v in feats.items()}
    return feats, times, [parse(l) for l in g['train_pairs']], [parse(l) for l in g['dev_pairs']]


@pytest.mark.parametrize('case', ['full', 'norand', 'sub', 'tiny'])
def test_frames_dataloader_matches_reference_batch_by_batch(case):
    """G9: the reference's FramesDataLoader (dataloader.py:580-739) iterated for several
    train / dev epochs from one numpy seed; ours must yield the same batches: same
    frame-pair dataset (DTW paths of the 'same' pairs, truncated 'diff' pairs), same
    np.random.shuffle draws at load time and per epoch, same slices, int64 labels."""
    from abnet3_amd.dataloader import FramesDataLoader
    g = load_golden('frames_loader.npz')
    feats, times, train, dev_pairs = corpus(g)
    kw = ast.literal_eval(str(g[case + '.kw']))
    dl = FramesDataLoader('unused', 'unused', **kw)
    dl.set_data(feats, times, train, dev_pairs)
    np.random.seed(int(g[case + '.seed']))
    X1, X2, Y, sizes, nb = [], [], [], [], []
    for mode in str(g[case + '.epochs']):
        n = 0
        for x1, x2, y in dl.batch_iterator(train_mode=(mode == 'T')):
            assert x1.is_cuda and x1.dtype == torch.float32 and y.dtype == torch.int64
            X1.append(x1.cpu().numpy()); X2.append(x2.cpu().numpy()); Y.append(y.cpu().numpy())
            sizes.append(len(y))
            n += 1
        nb.append(n)
    assert [len(dl.frame_pairs['train'][2]), len(dl.frame_pairs['dev'][2])] == list(g[case + '.n_frame_pairs'])
    assert nb == list(g[case + '.batches_per_epoch'])
    assert sizes == list(g[case + '.sizes'])
    assert (np.concatenate(Y) == g[case + '.y']).all()
    assert (np.vstack(X1) == g[case + '.X1']).all()
    assert (np.vstack(X2) == g[case + '.X2']).all()


def test_temporal_coherence_pairs_match_reference():
    """add_tcl_to_batch / temporal_coherence_loss (dataloader.py:314-352): the reference
    appends int(tcl * n / (1 - tcl)) frame pairs drawn with `random` -- one 'same' pair at
    distance 1 and four 'diff' pairs at distances 15..30 per draw -- to a word-pair batch."""
    from abnet3_amd.dataloader import OriginalDataLoader
    from abnet3_amd.utils import group_pairs
    g = load_golden('frames_loader.npz')
    feats, times, train, dev_pairs = corpus(g)
    dl = OriginalDataLoader('unused', 'unused', tcl=0.3)
    dl.set_data(feats, times, train, dev_pairs)
    dl.train_files = [str(f) for f in g['tcl.train_files']]
    batch = dl.frames_from_pairs_device(group_pairs(train[:4]))
    assert len(batch[2]) == int(g['tcl.n_before'])
    random.seed(4)
    X1, X2, Y = dl.add_tcl_to_batch(batch)
    n0 = int(g['tcl.n_before'])
    added = len(Y) - n0
    assert added == round(int(0.3 * n0 / 0.7) / 5) * 5 and added > 0
    assert Y.dtype == torch.float64            # the batch's label dtype (np.concatenate promotes)
    assert (Y.cpu().numpy() == g['tcl.Y']).all()
    assert (X1.cpu().numpy() == g['tcl.X1']).all()
    assert (X2.cpu().numpy() == g['tcl.X2']).all()
    tail = Y[n0:].cpu().numpy().reshape(-1, 5)
    assert (tail == np.array([1, -1, -1, -1, -1])).all()


def test_tcl_through_batch_iterator_shapes_and_labels():
    from abnet3_amd.dataloader import OriginalDataLoader
    g = load_golden('frames_loader.npz')
    feats, times, train, dev_pairs = corpus(g)
    dl = OriginalDataLoader('unused', 'unused', tcl=0.2, batch_size=4)
    dl.set_data(feats, times, train, dev_pairs)
    random.seed(0)
    np.random.seed(0)
    nb = 0
    for x1, x2, y in dl.batch_iterator(train_mode=True):
        assert x1.shape == x2.shape and x1.shape[1] == 40 and len(y) == x1.shape[0]
        assert set(np.unique(y.cpu().numpy())) <= {-1.0, 1.0}
        nb += 1
    assert nb == 3


def test_alignment_is_chunked_by_cell_budget():
    """align_pairs cuts the pair list into DTW calls of bounded size: a tiny
    budget must give the same alignments as one call."""
    from abnet3_amd.dataloader import FramesDataLoader
    g = load_golden('frames_loader.npz')
    feats, times, train, dev_pairs = corpus(g)
    ref = FramesDataLoader('unused', 'unused', batch_size=50)
    ref.set_data(feats, times, train, dev_pairs)
    small = FramesDataLoader('unused', 'unused', batch_size=50)
    small.ALIGN_CELL_BUDGET = 1500            # about one word pair per call
    small.set_data(feats, times, train, dev_pairs)
    np.random.seed(3)
    a = [tuple(t.cpu().numpy() for t in b) for b in ref.batch_iterator(True)]
    np.random.seed(3)
    b = [tuple(t.cpu().numpy() for t in bb) for bb in small.batch_iterator(True)]
    assert len(a) == len(b) > 0
    for u, v in zip(a, b):
        assert all((p == q).all() for p, q in zip(u, v))
