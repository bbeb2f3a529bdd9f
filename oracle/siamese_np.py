"""ORACLE (test infrastructure, never shipped as a product path).

Plain-numpy fp32 restatement of the reference's Siamese hot path.  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
Pinned against tests/golden/*.npz, which tools/make_golden.py produced by
importing the reference itself (tower_*, train_c1_*, train_mid_*, train_c2_*,
loss_edge).

Reference lines restated (all relative to /root/reference):
  abnet3/model.py:110-170   layer stack: [Linear, Dropout, (BatchNorm1d), act] x L
  abnet3/model.py:179-196   forward_once / forward (tower run twice, BN per call)
  abnet3/loss.py:46-67      coscos2
  abnet3/loss.py:85-105     cosmargin
  abnet3/trainer.py:68-87   optimizer choice (torch.optim defaults)
  abnet3/trainer.py:236-242 zero_grad / backward / step
  abnet3/model.py:211-376   SiameseMultitaskNetwork (shared trunk, two output heads;
                            hidden_layers_spk / _phn are built but never called)
  abnet3/loss.py:140-182    weighted_loss_multi
  abnet3/trainer.py:259-279 TrainerSiameseMultitask.give_batch_to_network
The backward pass restates what torch autograd executes for those modules.
Dropout takes explicit masks (0 or 1/(1-p)); torch's CPU RNG stream is not
restated, so p>0 is checked with masks shared between oracle and kernel.
"""
import numpy as np

F32 = np.float32
BN_EPS = F32(1e-5)
BN_MOMENTUM = F32(0.1)
COS_EPS = F32(1e-6)


class TowerSpec(object):
    """Static description of one SiameseNetwork (model.py:110-170)."""

    def __init__(self, input_dim, num_hidden_layers, hidden_dim, output_dim,
                 activation_layer, batch_norm=False, last_non_linearity='default'):
        self.dims = [input_dim] + [hidden_dim] * (num_hidden_layers + 1) + [output_dim]
        self.n_layers = len(self.dims) - 1
        self.act = activation_layer
        self.batch_norm = bool(batch_norm)
        if last_non_linearity == 'default':
            self.last_act = activation_layer
        elif last_non_linearity is None:
            self.last_act = 'none'
        else:
            self.last_act = last_non_linearity
        stride = 4 if self.batch_norm else 3   # Linear, Dropout, [BN], act
        self.lin_keys = ['input_emb.0']
        self.lin_keys += ['hidden_layers.%d' % (stride * i)
                          for i in range(num_hidden_layers)]
        self.lin_keys += ['output_layer.0']
        self.bn_keys = ['input_emb.2']
        self.bn_keys += ['hidden_layers.%d' % (stride * i + 2)
                         for i in range(num_hidden_layers)]
        self.bn_keys += ['output_layer.2']

    def layer_act(self, l):
        return self.last_act if l == self.n_layers - 1 else self.act

    def param_keys(self):
        """state_dict parameter keys in nn.Module.parameters() order."""
        keys = []
        for l in range(self.n_layers):
            keys += [self.lin_keys[l] + '.weight', self.lin_keys[l] + '.bias']
            if self.batch_norm:
                keys += [self.bn_keys[l] + '.weight', self.bn_keys[l] + '.bias']
        return keys


def act_fwd(z, kind):
    if kind == 'sigmoid':
        return (F32(1) / (F32(1) + np.exp(-z))).astype(F32)
    if kind == 'relu':
        return np.maximum(z, F32(0)).astype(F32)
    if kind == 'tanh':
        return np.tanh(z).astype(F32)
    if kind == 'none':
        return z
    if kind == 'softmax':      # nn.Softmax() on a 2-D input: over each row (model.py:161-166)
        e = np.exp((z - z.max(axis=1, keepdims=True)).astype(F32)).astype(F32)
        return (e / e.sum(axis=1, keepdims=True, dtype=F32)).astype(F32)
    raise ValueError(kind)


def act_bwd(a, da, kind):
    """Derivative expressed on the activation OUTPUT a (what autograd saves)."""
    if kind == 'sigmoid':
        return (da * a * (F32(1) - a)).astype(F32)
    if kind == 'relu':
        return (da * (a > 0)).astype(F32)
    if kind == 'tanh':
        return (da * (F32(1) - a * a)).astype(F32)
    if kind == 'none':
        return da
    if kind == 'softmax':
        dot = (da.astype(np.float64) * a).sum(axis=1, keepdims=True)
        return (a * (da - dot.astype(F32))).astype(F32)
    raise ValueError(kind)


def tower_forward(params, x, spec, train, update_running=True, masks=None):
    """forward_once (model.py:179-186). Returns (embedding, cache).

    BN train mode: batch mean / biased variance over THIS call's rows; running
    stats updated with momentum 0.1 and the unbiased variance, exactly once per
    call (so twice per Siamese forward)."""
    a = np.ascontiguousarray(x, dtype=F32)
    cache = {'a': [a], 'xhat': [], 'invstd': [], 'masks': masks if train else None}
    for l in range(spec.n_layers):
        W = params[spec.lin_keys[l] + '.weight']
        b = params[spec.lin_keys[l] + '.bias']
        z = (a @ W.T + b).astype(F32)
        if masks is not None and train:      # nn.Dropout between Linear and BN/act
            z = (z * masks[l]).astype(F32)
        if spec.batch_norm:
            k = spec.bn_keys[l]
            g, beta = params[k + '.weight'], params[k + '.bias']
            if train:
                n = z.shape[0]
                mean = z.mean(axis=0, dtype=np.float64).astype(F32)
                var = ((z - mean).astype(np.float64) ** 2).mean(axis=0).astype(F32)
                if update_running:
                    unb = var * F32(n / (n - 1.0)) if n > 1 else var
                    params[k + '.running_mean'] = (
                        (F32(1) - BN_MOMENTUM) * params[k + '.running_mean']
                        + BN_MOMENTUM * mean).astype(F32)
                    params[k + '.running_var'] = (
                        (F32(1) - BN_MOMENTUM) * params[k + '.running_var']
                        + BN_MOMENTUM * unb).astype(F32)
                    params[k + '.num_batches_tracked'] = \
                        params[k + '.num_batches_tracked'] + 1
            else:
                mean = params[k + '.running_mean']
                var = params[k + '.running_var']
            invstd = (F32(1) / np.sqrt(var + BN_EPS)).astype(F32)
            xhat = ((z - mean) * invstd).astype(F32)
            z = (xhat * g + beta).astype(F32)
            cache['xhat'].append(xhat)
            cache['invstd'].append(invstd)
        a = act_fwd(z, spec.layer_act(l))
        cache['a'].append(a)
    return a, cache


def tower_backward(params, cache, dout, spec, grads=None, return_dx=False):
    """Autograd of forward_once in train mode; accumulates into `grads`.
    return_dx: also return the gradient w.r.t. the tower's input (needed when the
    tower is a head sitting on a shared trunk)."""
    if grads is None:
        grads = {}
    da = np.ascontiguousarray(dout, dtype=F32)
    for l in reversed(range(spec.n_layers)):
        a_out = cache['a'][l + 1]
        a_in = cache['a'][l]
        dz = act_bwd(a_out, da, spec.layer_act(l))
        if spec.batch_norm:
            k = spec.bn_keys[l]
            g = params[k + '.weight']
            xhat, invstd = cache['xhat'][l], cache['invstd'][l]
            n = F32(dz.shape[0])
            dgamma = (dz * xhat).sum(axis=0, dtype=np.float64).astype(F32)
            dbeta = dz.sum(axis=0, dtype=np.float64).astype(F32)
            dz = ((g * invstd / n) * (n * dz - dbeta - xhat * dgamma)).astype(F32)
            _acc(grads, k + '.weight', dgamma)
            _acc(grads, k + '.bias', dbeta)
        if cache.get('masks') is not None:
            dz = (dz * cache['masks'][l]).astype(F32)
        W = params[spec.lin_keys[l] + '.weight']
        _acc(grads, spec.lin_keys[l] + '.weight', (dz.T @ a_in).astype(F32))
        _acc(grads, spec.lin_keys[l] + '.bias',
             dz.sum(axis=0, dtype=np.float64).astype(F32))
        if l > 0 or return_dx:
            da = (dz @ W).astype(F32)
    if return_dx:
        return grads, da
    return grads


def _acc(grads, key, val):
    grads[key] = val if key not in grads else (grads[key] + val).astype(F32)


def label_codes(y):
    """torch.eq(y, 1) / torch.eq(y, -1) on any label dtype (loss.py:60-63)."""
    y = np.asarray(y)
    return np.where(y == 1, 1, np.where(y == -1, -1, 0)).astype(np.int8)


def pair_loss(e1, e2, y, kind='coscos2', margin=0.5, avg=True):
    """coscos2 (loss.py:46-67) / cosmargin (loss.py:85-105) with gradients.

    cos = x.y / (max(|x|,eps) * max(|y|,eps)), eps=1e-6 (nn.CosineSimilarity).
    Returns (loss float64, de1 f32, de2 f32, cos f32)."""
    e1 = np.asarray(e1, dtype=F32)
    e2 = np.asarray(e2, dtype=F32)
    assert e1.shape == e2.shape, 'Input not the same size'
    code = label_codes(y)
    B = e1.shape[0]
    dot = (e1.astype(np.float64) * e2).sum(axis=1)
    n1 = np.sqrt((e1.astype(np.float64) ** 2).sum(axis=1))
    n2 = np.sqrt((e2.astype(np.float64) ** 2).sum(axis=1))
    c1 = np.maximum(n1, COS_EPS)
    c2 = np.maximum(n2, COS_EPS)
    cos = dot / (c1 * c2)
    if kind == 'coscos2':
        term = np.where(code == 1, (1 - cos) / 2, np.where(code == -1, cos * cos, cos))
        dcos = np.where(code == 1, -0.5, np.where(code == -1, 2 * cos, 1.0))
    elif kind == 'cosmargin':
        hinge = np.maximum(cos - margin, 0.0)
        term = np.where(code == 1, 1 - cos, np.where(code == -1, hinge, cos))
        # torch.clamp(min=0) passes gradient where input >= min
        dcos = np.where(code == 1, -1.0,
                        np.where(code == -1, (cos - margin >= 0) * 1.0, 1.0))
    else:
        raise ValueError(kind)
    loss = term.sum()
    scale = 1.0 / B if avg else 1.0
    loss = loss * scale
    dcos = dcos * scale
    # ATen clamps the norms in place under NoGradGuard, so autograd still sees
    # d|x|/dx = x/|x| (0 at |x| = 0) while the VALUE is max(|x|, eps):
    #   d cos / d x = y/(c1 c2) - cos/c1 * x/|x|
    # which is the textbook y/(|x||y|) - cos x/|x|^2 whenever |x| >= eps.
    inv = 1.0 / (c1 * c2)
    k1 = np.where(n1 > 0, cos / (c1 * np.where(n1 > 0, n1, 1.0)), 0.0)
    k2 = np.where(n2 > 0, cos / (c2 * np.where(n2 > 0, n2, 1.0)), 0.0)
    de1 = dcos[:, None] * (e2 * inv[:, None] - e1 * k1[:, None])
    de2 = dcos[:, None] * (e1 * inv[:, None] - e2 * k2[:, None])
    return float(loss), de1.astype(F32), de2.astype(F32), cos.astype(F32)


# ----------------------------------------------------------------------------
# torch.optim restatements (trainer.py:68-87; torch defaults for the rest)
# ----------------------------------------------------------------------------
class Optimizer(object):
    def __init__(self, kind, lr, momentum=0.9):
        self.kind, self.lr, self.momentum = kind, F32(lr), F32(momentum)
        self.state = {}
        self.t = 0

    def step(self, params, grads, keys):
        self.t += 1
        for k in keys:
            p, g = params[k], grads[k].astype(F32)
            st = self.state.setdefault(k, {})
            if self.kind == 'sgd':
                if 'buf' not in st:
                    st['buf'] = g.copy()
                else:
                    st['buf'] = (self.momentum * st['buf'] + g).astype(F32)
                p = p - self.lr * st['buf']
            elif self.kind == 'adadelta':
                rho, eps = F32(0.9), F32(1e-6)
                sq = st.get('sq', np.zeros_like(p))
                ad = st.get('ad', np.zeros_like(p))
                sq = (rho * sq + (F32(1) - rho) * g * g).astype(F32)
                std = np.sqrt(sq + eps)
                delta = (np.sqrt(ad + eps) / std * g).astype(F32)
                ad = (rho * ad + (F32(1) - rho) * delta * delta).astype(F32)
                st['sq'], st['ad'] = sq, ad
                p = p - self.lr * delta
            elif self.kind == 'adam':
                b1, b2, eps = 0.9, 0.999, 1e-8
                m = st.get('m', np.zeros_like(p))
                v = st.get('v', np.zeros_like(p))
                m = (F32(b1) * m + F32(1 - b1) * g).astype(F32)
                v = (F32(b2) * v + F32(1 - b2) * g * g).astype(F32)
                st['m'], st['v'] = m, v
                bc1 = 1 - b1 ** self.t
                bc2 = 1 - b2 ** self.t
                denom = (np.sqrt(v) / F32(np.sqrt(bc2)) + F32(eps)).astype(F32)
                p = p - F32(float(self.lr) / bc1) * (m / denom)
            elif self.kind == 'adagrad':
                s = st.get('sum', np.zeros_like(p))
                s = (s + g * g).astype(F32)
                st['sum'] = s
                p = p - self.lr * (g / (np.sqrt(s) + F32(1e-10)))
            elif self.kind == 'RMSprop':
                alpha, eps = F32(0.99), F32(1e-8)
                sq = st.get('sq', np.zeros_like(p))
                sq = (alpha * sq + (F32(1) - alpha) * g * g).astype(F32)
                st['sq'] = sq
                p = p - self.lr * (g / (np.sqrt(sq) + eps))
            else:
                raise ValueError(self.kind)
            params[k] = p.astype(F32)


def siamese_forward(params, x1, x2, spec, train):
    """SiameseNetwork.forward (model.py:188-196): two tower calls, shared weights."""
    e1, c1 = tower_forward(params, x1, spec, train)
    e2, c2 = tower_forward(params, x2, spec, train)
    return e1, e2, (c1, c2)


def train_step(params, x1, x2, y, spec, opt, kind='coscos2', margin=0.5,
               avg=True, do_training=True):
    """One batch of TrainerSiamese.optimize_model (trainer.py:236-242)."""
    e1, e2, (c1, c2) = siamese_forward(params, x1, x2, spec, train=True)
    loss, de1, de2, _ = pair_loss(e1, e2, y, kind, margin, avg)
    grads = {}
    if do_training:
        tower_backward(params, c1, de1, spec, grads)
        tower_backward(params, c2, de2, spec, grads)
        opt.step(params, grads, spec.param_keys())
    return loss, grads, (e1, e2)


# ---- multitask (model.py:211-376, loss.py:140-182, trainer.py:259-279) ------------

class _Stack(object):
    """A run of [Linear, Dropout, (BN), act] blocks with explicit state_dict keys:
    the interface tower_forward / tower_backward need."""

    def __init__(self, dims, lin_keys, bn_keys, act, batch_norm):
        self.dims = dims
        self.n_layers = len(dims) - 1
        self.act = self.last_act = act
        self.batch_norm = bool(batch_norm)
        self.lin_keys, self.bn_keys = lin_keys, bn_keys

    def layer_act(self, l):
        return self.act


class MultitaskSpec(object):
    """SiameseMultitaskNetwork: input_emb + hidden_layers_shared feed BOTH
    output_layer_spk and output_layer_phn (forward_once, model.py:337-345).
    hidden_layers_spk / hidden_layers_phn hold parameters that forward never
    uses: they receive no gradient and no optimizer update."""

    def __init__(self, input_dim, num_hidden_layers_shared, num_hidden_layers_spk,
                 num_hidden_layers_phn, hidden_dim, output_dim, activation_layer,
                 batch_norm=False):
        stride = 4 if batch_norm else 3
        H = hidden_dim
        self.batch_norm = bool(batch_norm)
        self.trunk = _Stack(
            [input_dim] + [H] * (num_hidden_layers_shared + 1),
            ['input_emb.0'] + ['hidden_layers_shared.%d' % (stride * i)
                               for i in range(num_hidden_layers_shared)],
            ['input_emb.2'] + ['hidden_layers_shared.%d' % (stride * i + 2)
                               for i in range(num_hidden_layers_shared)],
            activation_layer, batch_norm)
        self.head_spk = _Stack([H, output_dim], ['output_layer_spk.0'], ['output_layer_spk.2'],
                               activation_layer, batch_norm)
        self.head_phn = _Stack([H, output_dim], ['output_layer_phn.0'], ['output_layer_phn.2'],
                               activation_layer, batch_norm)
        self.dead = (['hidden_layers_spk.%d' % (stride * i) for i in range(num_hidden_layers_spk)]
                     + ['hidden_layers_phn.%d' % (stride * i) for i in range(num_hidden_layers_phn)])

    def live_param_keys(self):
        keys = []
        for st in (self.trunk, self.head_spk, self.head_phn):
            for l in range(st.n_layers):
                keys += [st.lin_keys[l] + '.weight', st.lin_keys[l] + '.bias']
                if st.batch_norm:
                    keys += [st.bn_keys[l] + '.weight', st.bn_keys[l] + '.bias']
        return keys


def multitask_forward_once(params, x, spec, train):
    h, ct = tower_forward(params, x, spec.trunk, train)
    es, cs = tower_forward(params, h, spec.head_spk, train)
    ep, cp = tower_forward(params, h, spec.head_phn, train)
    return es, ep, (ct, cs, cp)


def multitask_train_step(params, x1, x2, y_spk, y_phn, spec, opt, weight,
                         spk=('coscos2', 0.5, True), phn=('coscos2', 0.5, True),
                         do_training=True):
    """One batch of TrainerSiameseMultitask: loss = w*L_spk + (1-w)*L_phn
    (loss.py:178-181); spk / phn = (kind, margin, avg) of the two base losses."""
    s1, p1, c1 = multitask_forward_once(params, x1, spec, True)
    s2, p2, c2 = multitask_forward_once(params, x2, spec, True)
    ls, ds1, ds2, _ = pair_loss(s1, s2, y_spk, *spk)
    lp, dp1, dp2, _ = pair_loss(p1, p2, y_phn, *phn)
    w = F32(weight)
    loss = F32(w * F32(ls) + (F32(1.0) - w) * F32(lp))
    grads = {}
    if do_training:
        for (ct, cs, cp), ds, dp in ((c1, ds1, dp1), (c2, ds2, dp2)):
            _, dh_s = tower_backward(params, cs, (w * ds).astype(F32), spec.head_spk, grads, True)
            _, dh_p = tower_backward(params, cp, ((F32(1.0) - w) * dp).astype(F32), spec.head_phn,
                                     grads, True)
            tower_backward(params, ct, (dh_s + dh_p).astype(F32), spec.trunk, grads)
        opt.step(params, grads, spec.live_param_keys())
    return loss, grads, (s1, p1, s2, p2)
