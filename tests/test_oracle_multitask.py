"""The numpy oracle's multitask restatement against the golden
vectors the reference produced (tools/make_golden.py G8).  CPU only."""
import ast

import numpy as np
import pytest

from conftest import load_golden, rel_err, check_grads, check_params
from oracle import siamese_np as O

TOL = 1e-5
# sigmoid towers start with cos(e1, e2) within 1e-5 of 1: the loss gradient is a
# difference of nearly equal vectors and the reference's own fp32 gradients carry
# ~1e-4 of rounding (same effect as the train_mid / train_c2 fixtures)
GTOL = {'sig': 1e-4, 'relu_bn': 2e-5, 'tanh0': 2e-5}
RUNS = [('sgd', 0.3, 0.001), ('adadelta', 0.5, 0.1), ('adam', 1.0, 0.001)]


def spec_from(g):
    kw = ast.literal_eval(str(g['kw']))
    return O.MultitaskSpec(kw['input_dim'], kw['num_hidden_layers_shared'],
                           kw['num_hidden_layers_spk'], kw['num_hidden_layers_phn'],
                           kw['hidden_dim'], kw['output_dim'], kw['activation_layer'],
                           kw.get('batch_norm', False)), kw


def params_from(g, prefix='p.'):
    return {k[len(prefix):]: v.copy() for k, v in g.items() if k.startswith(prefix)}


@pytest.mark.parametrize('name', ['sig', 'relu_bn', 'tanh0'])
def test_multitask_forward_eval(name):
    g = load_golden('multitask_%s.npz' % name)
    spec, _ = spec_from(g)
    p = params_from(g)
    s1, p1, _ = O.multitask_forward_once(p, g['x1'], spec, False)
    s2, p2, _ = O.multitask_forward_once(p, g['x2'], spec, False)
    for mine, key in ((s1, 'spk1'), (p1, 'phn1'), (s2, 'spk2'), (p2, 'phn2')):
        assert rel_err(mine, g['eval_' + key]) < TOL, key


@pytest.mark.parametrize('name', ['sig', 'relu_bn', 'tanh0'])
@pytest.mark.parametrize('oname,weight,lr', RUNS)
def test_multitask_grads_and_three_steps(name, oname, weight, lr):
    g = load_golden('multitask_%s.npz' % name)
    spec, _ = spec_from(g)
    p = params_from(g)
    tag = '%s.w%g' % (oname, weight)
    opt = O.Optimizer(oname, lr)
    keys = spec.live_param_keys()
    losses = []
    for s in range(3):
        loss, grads, emb = O.multitask_train_step(
            p, g['x1'], g['x2'], g['y_spk'], g['y_phn'], spec, opt, weight,
            spk=('coscos2', 0.5, False), phn=('cosmargin', 0.4, True))
        losses.append(loss)
        if s == 0:
            for mine, key in zip(emb, ('spk1', 'phn1', 'spk2', 'phn2')):
                assert rel_err(mine, g['%s.%s_0' % (tag, key)]) < TOL, key
            ref = {k: g['%s.grad0.%s' % (tag, k)] for k in keys}
            if weight == 1.0:          # the phoneme head gets an exactly-zero gradient
                for k in keys:
                    if k.startswith('output_layer_phn'):
                        assert not np.any(ref[k]) and not np.any(grads[k]), k
                live = [k for k in keys if not k.startswith('output_layer_phn')]
                check_grads(grads, ref, live, spec.batch_norm, GTOL[name])
            else:
                check_grads(grads, ref, keys, spec.batch_norm, GTOL[name])
            # branches forward never calls: the reference leaves p.grad = None
            for k in spec.dead:
                assert g['%s.grad0.%s.weight' % (tag, k)].size == 0
    assert np.allclose(losses, g[tag + '.losses'], rtol=1e-5, atol=1e-6)
    after = {k: g['%s.after.%s' % (tag, k)] for k in keys}
    # biases start at 0, so after three steps they ARE the (scaled) gradients
    check_params(p, after, keys, spec.batch_norm,
                 max(5 * GTOL[name], 3e-4) if oname == 'adam' else 5 * GTOL[name] if name == 'sig' else 1e-5)
    for k in spec.dead:                # ... and the optimizer never touches them
        for leaf in ('.weight', '.bias'):
            assert np.array_equal(g['%s.after.%s%s' % (tag, k, leaf)], g['p.' + k + leaf])
    if spec.batch_norm and oname != 'adam':
        for k in p:
            if 'running' in k:
                assert rel_err(p[k], g['%s.after.%s' % (tag, k)]) < TOL, k
            if 'num_batches_tracked' in k and not k.startswith(('hidden_layers_spk', 'hidden_layers_phn')):
                assert int(p[k]) == int(g['%s.after.%s' % (tag, k)]) == 6
