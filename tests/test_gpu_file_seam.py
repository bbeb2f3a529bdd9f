"""The file-level entry points the gridsearch really calls (abnet3/gridsearch.py:204-231) -- features.generate(),
the dataloader's load_data() inside trainer.train(), embedder.embed() -- end to end on files: wav files in, an
h5features file of stacked filterbanks, pairs files, a trained network, an h5features file of embeddings out.  The
`h5features` package is absent from this image; tests/fake_h5features.py stands in for its handful of calls (a pickle
per file), so what is verified is abnet3_amd's own code at that seam -- which arrays it hands over, under which names,
with which time stamps -- not the HDF5 format."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def h5features(monkeypatch):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fake_h5features
    monkeypatch.setitem(sys.modules, 'h5features', fake_h5features)
    return fake_h5features


def test_generate_train_embed_on_files(tmp_path, h5features):
    import torch
    from scipy.io import wavfile
    import abnet3_amd.loss as L
    from abnet3_amd.features import FeaturesGenerator
    from abnet3_amd.model import SiameseNetwork
    from abnet3_amd.dataloader import OriginalDataLoader
    from abnet3_amd.trainer import TrainerSiamese
    from abnet3_amd.embedder import EmbedderSiamese
    from abnet3_amd.utils import write_dataset
    rng = np.random.default_rng(5)
    wavdir = tmp_path / 'wav'
    wavdir.mkdir()
    waves = {}
    for u in range(5):
        n = 16000 * 3 + 137 * u
        waves['utt%d' % u] = (3000 * np.sin(2 * np.pi * (200 + 40 * u) * np.arange(n) / 16000) + 300 * rng.standard_normal(n)).astype(np.int16)
        wavfile.write(str(wavdir / ('utt%d.wav' % u)), 16000, waves['utt%d' % u])
    feat_path = str(tmp_path / 'exp' / 'fb40_stacked7.features')
    fg = FeaturesGenerator(files=str(wavdir), output_path=feat_path, normalization=True, norm_per_file=False,
                           norm_per_channel=True, stack=True, nframes=7, run='once')
    fg.generate()
    with h5features.Reader(feat_path, 'features') as fh:
        data = fh.read()
    assert sorted(data.items()) == sorted(waves)
    # what generate() wrote = the in-memory pipeline on the same audio, item by item, with h5features_compute's times
    names = list(data.items())
    table, _, nfr, times = fg.features_from_waves([waves[k] for k in names], 16000, names)
    offs = np.concatenate(([0], np.cumsum(nfr)))
    for i, k in enumerate(names):
        f = data.dict_features()[k]
        assert f.dtype == np.float32 and f.shape == (int(nfr[i]), 280)
        assert np.array_equal(f, table[offs[i]:offs[i + 1]].cpu().numpy())
        assert np.allclose(data.dict_labels()[k], np.arange(int(nfr[i])) * 0.01 + 0.0025)
    # pairs files as the sampler writes them (abnet3/sampler.py:697-742), read by load_data() inside train()
    toks = [(k, round(0.2 + 0.25 * j, 2), round(0.2 + 0.25 * j + 0.18, 2)) for k in names for j in range(8)]
    def pairs(n):
        out = []
        for _ in range(n):
            a, b, c, d = (toks[i] for i in rng.choice(len(toks), 4, replace=False))
            out += [a + b + ('same',), c + d + ('diff',)]
        return out
    pairs_dir = tmp_path / 'exp' / 'pairs'
    for sub, n in (('train_pairs', 24), ('dev_pairs', 8)):
        (pairs_dir / sub).mkdir(parents=True)
        write_dataset(str(pairs_dir / sub / 'dataset'), pairs(n))
    net = SiameseNetwork(input_dim=280, num_hidden_layers=1, hidden_dim=64, output_dim=24, p_dropout=0.0,
                         activation_layer='sigmoid', output_path=str(tmp_path / 'exp' / 'network'))
    dl = OriginalDataLoader(pairs_path=str(pairs_dir), features_path=feat_path, batch_size=4, num_max_minibatches=100)
    tr = TrainerSiamese(network=net, loss=L.coscos2(avg=False), optimizer_type='adadelta', lr=0.1, num_epochs=2, patience=5,
                        dataloader=dl, log_dir=str(tmp_path / 'exp' / 'logs'))
    tr.train()
    assert len(tr.train_losses) == 3 and np.isfinite(tr.train_losses).all()
    assert os.path.exists(net.output_path + '.pth') and os.path.exists(net.output_path + '.params')
    # the embedder reads the feature file and the saved network, writes the embeddings file
    out_path = str(tmp_path / 'exp' / 'embeddings.h5f')
    emb = EmbedderSiamese(network=net, network_path=net.output_path + '.pth', feature_path=feat_path, output_path=out_path)
    emb.embed()
    with h5features.Reader(out_path, 'features') as fh:
        e = fh.read()
    assert list(e.items()) == names
    ref = emb.embed_features([data.dict_features()[k] for k in names])
    for i, k in enumerate(names):
        assert e.dict_features()[k].shape == (int(nfr[i]), 24)
        assert np.array_equal(e.dict_features()[k], ref[i])
        assert np.array_equal(e.dict_labels()[k], data.dict_labels()[k])
