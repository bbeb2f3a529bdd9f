"""The caller's side of the drop-in boundary: GridSearch.run_single_experiment (abnet3/gridsearch.py:145-202)
builds every object of an experiment as getattr(abnet3.<module>, cfg['class'])(**cfg['arguments']) after injecting
a few arguments.  tests/golden/gridsearch_buckeye.json (tools/make_golden.py G11) holds the reference's own test
configuration (test/data/buckeye.yaml's default_params, `cuda: False` included) and what the reference's classes
are when built from it; here abnet3_amd's modules take the reference's place in the very same statements."""
import copy
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN


def load():
    with open(os.path.join(GOLDEN, 'gridsearch_buckeye.json')) as fh:
        return json.load(fh)


def build_experiment(exp_dir, params, ref, with_trainer):
    """gridsearch.py:145-202 with `abnet3` spelled `abnet3_amd` (the sampler is outside the hot path: SURVEY.md 2)."""
    import abnet3_amd.features, abnet3_amd.model, abnet3_amd.loss, abnet3_amd.dataloader      # noqa
    import abnet3_amd.trainer, abnet3_amd.embedder                                          # noqa
    import abnet3_amd as abnet3
    single_experiment = copy.deepcopy(params)
    single_experiment['pathname_experience'] = str(exp_dir)

    features_prop = single_experiment['features']
    features_class = getattr(abnet3.features, features_prop['class'])
    arguments = features_prop['arguments']
    for k in ref['features.rejected_kwargs']:        # the reference's own constructor raises TypeError on these
        arguments.pop(k)
    if 'output_path' not in arguments:
        arguments['output_path'] = os.path.join(single_experiment['pathname_experience'], 'features')
    features = features_class(**arguments)

    model_prop = single_experiment['model']
    model_class = getattr(abnet3.model, model_prop['class'])
    arguments = model_prop['arguments']
    arguments['output_path'] = os.path.join(single_experiment['pathname_experience'], 'network')
    model = model_class(**arguments)

    loss_prop = single_experiment['loss']
    loss_class = getattr(abnet3.loss, loss_prop['class'])
    loss = loss_class(**loss_prop['arguments'])

    dataloader_prop = single_experiment['dataloader']
    dataloader_class = getattr(abnet3.dataloader, dataloader_prop['class'])
    arguments = dataloader_prop['arguments']
    if 'pairs_path' not in arguments:
        arguments['pairs_path'] = os.path.join(single_experiment['pathname_experience'], 'pairs')      # sampler.directory_output
    arguments['features_path'] = features.output_path
    dataloader = dataloader_class(**arguments)

    trainer = None
    if with_trainer:
        trainer_prop = single_experiment['trainer']
        trainer_class = getattr(abnet3.trainer, trainer_prop['class'])
        arguments = trainer_prop['arguments']
        arguments['network'] = model
        arguments['loss'] = loss
        arguments['dataloader'] = dataloader
        arguments['log_dir'] = os.path.join(single_experiment['pathname_experience'], 'logs')
        with pytest.warns(UserWarning, match='no CPU path'):      # `cuda: False`: there is no CPU path, said out loud
            trainer = trainer_class(**arguments)

    embedder_prop = single_experiment['embedder']
    embedder_class = getattr(abnet3.embedder, embedder_prop['class'])
    arguments = embedder_prop['arguments']
    arguments['network'] = model
    if 'output_path' not in arguments:
        arguments['output_path'] = os.path.join(single_experiment['pathname_experience'], 'embeddings.h5f')
    arguments['feature_path'] = features.output_path
    arguments['network_path'] = model.output_path + '.pth'
    embedder = embedder_class(**arguments)
    return features, model, loss, dataloader, trainer, embedder


def check_against_reference(ref, features, model, loss, dataloader, embedder):
    # every keyword argument the reference's constructors accept, ours accept (build_experiment passed them all)
    for section in ('model', 'loss', 'dataloader', 'trainer', 'embedder'):
        assert ref[section + '.rejected_kwargs'] == []
    assert features.run == ref['features.run'] and features.nframes == 7 and features.n_filters == 40
    assert list(model.state_dict().keys()) == ref['model.state_dict_keys']
    assert sum(p.numel() for p in model.parameters()) == ref['model.n_parameters']
    assert sorted(model.whoami().keys()) == ref['model.whoami_keys']
    assert type(loss).__name__ == ref['loss.class'] and bool(loss.avg) == ref['loss.avg']
    assert sorted(loss.whoami().keys()) == ref['loss.whoami_keys']
    assert dataloader.batch_size == ref['dataloader.batch_size']
    assert dataloader.num_max_minibatches == ref['dataloader.num_max_minibatches']
    assert embedder.batch_size == ref['embedder.batch_size'] and embedder.network is model


def test_objects_build_from_the_reference_configuration(tmp_path):
    """No GPU needed up to the trainer (whose constructor moves the network to the device)."""
    d = load()
    features, model, loss, dataloader, _, embedder = build_experiment(tmp_path, d['default_params'], d['reference'], False)
    check_against_reference(d['reference'], features, model, loss, dataloader, embedder)


@pytest.mark.gpu
def test_experiment_trains_and_embeds(tmp_path):
    """The whole of run_single_experiment's object graph on the device: built from the YAML's dictionary, fed
    in-memory features and word pairs (h5features is absent: set_data stands where load_data reads files), one
    epoch of trainer.train(), then the embedder on the same features."""
    import torch
    d = load()
    params = copy.deepcopy(d['default_params'])
    params['trainer']['arguments']['num_epochs'] = 1
    features, model, loss, dataloader, trainer, embedder = build_experiment(tmp_path, params, d['reference'], True)
    ref = d['reference']
    check_against_reference(ref, features, model, loss, dataloader, embedder)
    assert trainer.optimizer.kind == ref['trainer.optimizer'].lower() and trainer.lr == ref['trainer.lr']
    assert trainer.patience == ref['trainer.patience'] and trainer.num_epochs == 1
    assert isinstance(trainer.loss, type(loss)) and trainer.network is model

    rng = np.random.default_rng(0)
    feats, times, tokens = {}, {}, []
    for u in range(6):
        n = 400
        feats['utt%d' % u] = rng.standard_normal((n, 280)).astype(np.float32)
        times['utt%d' % u] = np.arange(n) * 0.01 + 0.0125
        tokens += [('utt%d' % u, round(0.3 * k + 0.05, 2), round(0.3 * k + 0.05 + rng.uniform(0.1, 0.25), 2)) for k in range(12)]
    def pairs(n):          # the lines of a pairs file (read_dataset): (f1, s1, e1, f2, s2, e2, type)
        out = []
        for _ in range(n):
            a, b, c, e = (tokens[i] for i in rng.choice(len(tokens), 4, replace=False))
            out += [a + b + ('same',), c + e + ('diff',)]
        return out
    dataloader.set_data(feats, times, train_pairs=pairs(40), dev_pairs=pairs(16))
    before = {k: v.clone() for k, v in model.state_dict().items()}
    trainer.train()
    assert len(trainer.train_losses) == 2 and len(trainer.dev_losses) == 2      # pass 0 + one epoch
    assert all(np.isfinite(trainer.train_losses)) and all(np.isfinite(trainer.dev_losses))
    assert any(not torch.equal(before[k].cpu(), v.cpu()) for k, v in model.state_dict().items())
    assert os.path.exists(model.output_path + '.pth')
    if trainer.best_epoch == 0 and trainer.best_dev is not None:
        assert os.path.exists(model.output_path + '.params')
    out = embedder.embed_features([feats['utt0'], feats['utt1'][:17]])
    assert out[0].shape == (400, 100) and out[1].shape == (17, 100) and np.isfinite(out[0]).all()
