"""Pair losses on MI355X: the class surface of abnet3/loss.py, HIP inside.

Mirrors (file:line relative to the reference checkout)
  LossBuilder   abnet3/loss.py:15-34
  coscos2       abnet3/loss.py:37-67
  cosmargin     abnet3/loss.py:70-105
  weighted_loss_multi  abnet3/loss.py:140-182
forward(input1, input2, y) returns a 0-dim tensor with .backward(); the
arithmetic (cosine similarity with eps=1e-6, per-label transform, sum, /N) and
its gradient run fused in one kernel (abn_pair_loss).
"""
import torch
import torch.nn as nn

from . import _lib


_UNIT = {}        # device -> 0-dim tensor holding 1.0, never written again


def unit_grad(device):
    """The gradient seed d loss / d loss = 1 as a cached device tensor.  A trainer
    that starts backward with it (torch.autograd.backward(loss, unit_grad(dev)))
    saves autograd's ones_like fill, and the pair loss recognises it by address and
    hands its stored gradient back without the multiply by 1."""
    key = torch.device(device)
    if key.type == 'cuda' and key.index is None:
        key = torch.device('cuda', torch.cuda.current_device())
    t = _UNIT.get(key)
    if t is None:
        t = _UNIT[key] = torch.ones((), dtype=torch.float32, device=key)
    return t


_WS = {}          # (device index, stream) -> zero-initialised scratch (partial sums + ticket counter)
# a failed launch may leave a ticket counter non-zero: the cached buffers are dropped (and zeroed afresh on the next call)
_lib._ON_ERROR.append(_WS.clear)


def _scratch(nbytes, device):
    """abn_pair_loss's scratch: its ticket counter has to be zero before the first call and is
    left zero by every call, so one buffer per (device, stream) is zeroed once and reused."""
    if torch.cuda.is_current_stream_capturing():
        # inside a hipGraph capture: a buffer of the graph's own pool, zeroed by a captured memset
        return torch.zeros(max(int(nbytes), 4096), dtype=torch.uint8, device=device)
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _WS[key] = torch.zeros(max(int(nbytes), 4096), dtype=torch.uint8, device=device)
    return ws


def _pair_loss_raw(e1, e2, y, kind, margin, avg, need_grad, act=None, masks=None):
    """abn_pair_loss: (0-dim loss, [2, B, D] gradient of the loss w.r.t. (e1, e2) or None).
    With `act` (the activation that produced e1 / e2) the gradient is taken w.r.t. the
    pre-activations instead (abn_pair_loss_dz; `masks` = the output layer's dropout
    multipliers for the two towers, or None)."""
    lib = _lib.load()
    _lib.require_device(e1, e2, y)
    if e1.dtype != torch.float32 or e2.dtype != torch.float32:
        raise TypeError('abnet3_amd: embeddings must be float32')
    if y.dtype not in _lib.Y_DTYPE:
        raise TypeError('abnet3_amd: unsupported label dtype %s' % y.dtype)
    B, D = e1.shape
    if y.numel() != B:
        raise ValueError('abnet3_amd: %d labels for %d pairs' % (y.numel(), B))
    e1, e2, y = e1.contiguous(), e2.contiguous(), y.contiguous()
    loss = torch.empty((), dtype=torch.float32, device=e1.device)
    de = torch.empty(2, B, D, dtype=torch.float32, device=e1.device) if need_grad else None
    ws = _scratch(lib.abn_pair_loss_ws_bytes(B), e1.device)
    if act is not None:
        m1, m2 = masks if masks is not None else (None, None)
        _lib.check(lib.abn_pair_loss_dz(
            _lib.ptr(e1), _lib.ptr(e2), _lib.ptr(y), _lib.Y_DTYPE[y.dtype], B, D,
            _lib.LOSS[kind], float(margin), int(bool(avg)), _lib.ACT[act], _lib.ptr(m1), _lib.ptr(m2),
            _lib.ptr(loss), _lib.ptr(de[0]), _lib.ptr(de[1]), _lib.ptr(ws), _lib.stream()),
            'abn_pair_loss_dz')
        return loss, de
    _lib.check(lib.abn_pair_loss(
        _lib.ptr(e1), _lib.ptr(e2), _lib.ptr(y), _lib.Y_DTYPE[y.dtype], B, D,
        _lib.LOSS[kind], float(margin), int(bool(avg)), _lib.ptr(loss),
        _lib.ptr(de[0]) if need_grad else None,
        _lib.ptr(de[1]) if need_grad else None, _lib.ptr(ws), _lib.stream()),
        'abn_pair_loss')
    return loss, de


class _PairLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, e1, e2, y, kind, margin, avg):
        need_grad = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        loss, ctx.de = _pair_loss_raw(e1, e2, y, kind, margin, avg, need_grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        de = ctx.de
        if de is None:
            return None, None, None, None, None, None
        # d loss / d e was produced with the forward; chain the incoming scalar
        # (unless it is the cached unit seed, identified by its address)
        unit = _UNIT.get(g.device)
        if unit is None or g.data_ptr() != unit.data_ptr():
            de = de * g
        return de[0], de[1], None, None, None, None


class LossBuilder(nn.Module):
    """Generic Loss class for ABnet3 (abnet3/loss.py:15-34)."""

    def __init__(self, *args, **kwargs):
        super(LossBuilder, self).__init__(*args, **kwargs)

    def forward(self, *args, **kwargs):
        raise NotImplementedError('Unimplemented forward for class:',
                                  self.__class__.__name__)

    def whoami(self, *args, **kwargs):
        return {'params': self.__dict__, 'class_name': self.__class__.__name__}


def _pair_loss(input1, input2, y, kind, margin, avg):
    assert input1.size() == input2.size(), 'Input not the same size'
    return _PairLossFunction.apply(input1.contiguous(), input2.contiguous(),
                                   y.contiguous(), kind, margin, avg)


class coscos2(LossBuilder):
    """coscos2 Loss function (abnet3/loss.py:37-67)."""

    def __init__(self, avg=True, *args, **kwargs):
        super(coscos2, self).__init__(*args, **kwargs)
        self.avg = avg

    def forward(self, input1, input2, y):
        return _pair_loss(input1, input2, y, 'coscos2', 0.0, self.avg)

    def value_and_grad(self, input1, input2, y):
        """(loss, [2, B, D] gradient w.r.t. (input1, input2)) without autograd: what
        forward(...).backward() yields, for a trainer that drives the kernels itself."""
        assert input1.size() == input2.size(), 'Input not the same size'
        return _pair_loss_raw(input1, input2, y, 'coscos2', 0.0, self.avg, True)

    def value_and_dz(self, input1, input2, y, act, masks=None):
        """(loss, [2, B, D] gradient w.r.t. the PRE-activations of the layer that produced the
        inputs through `act`): the loss gradient and the output layer's activation derivative
        in one launch."""
        assert input1.size() == input2.size(), 'Input not the same size'
        return _pair_loss_raw(input1, input2, y, 'coscos2', 0.0, self.avg, True, act=act, masks=masks)


class cosmargin(LossBuilder):
    """cosmargin Loss function (abnet3/loss.py:70-105); margin in [0, 1]."""

    def __init__(self, avg=True, margin=0.5, *args, **kwargs):
        super(cosmargin, self).__init__(*args, **kwargs)
        self.margin = margin
        self.avg = avg
        assert (margin >= 0 and margin <= 1)

    def forward(self, input1, input2, y, avg=True):
        # like the reference, the `avg` ARGUMENT is ignored (loss.py:103)
        return _pair_loss(input1, input2, y, 'cosmargin', self.margin, self.avg)

    def value_and_grad(self, input1, input2, y):
        assert input1.size() == input2.size(), 'Input not the same size'
        return _pair_loss_raw(input1, input2, y, 'cosmargin', self.margin, self.avg, True)

    def value_and_dz(self, input1, input2, y, act, masks=None):
        assert input1.size() == input2.size(), 'Input not the same size'
        return _pair_loss_raw(input1, input2, y, 'cosmargin', self.margin, self.avg, True, act=act, masks=masks)


class weighted_loss_multi(LossBuilder):
    """Weighted loss for multi-task training based on two pair losses
    (abnet3/loss.py:140-182): weight*loss_spk + (1 - weight)*loss_phn.

    Parameters
    ----------
    loss_phn, loss_spk : abnet3_amd.loss functions (coscos2 / cosmargin)
    weight : float
        variable between 0 and 1, to weight one or the other task.
    """

    def __init__(self, avg=True, loss_phn=None, loss_spk=None,
                 weight=0.5, *args, **kwargs):
        super(weighted_loss_multi, self).__init__(*args, **kwargs)
        assert type(weight) is float
        assert (weight >= 0 and weight <= 1)
        self.weight = weight
        self.avg = avg
        self.loss_phn = loss_phn
        self.loss_spk = loss_spk

    def forward(self, emb_spk1, emb_phn1, emb_spk2, emb_phn2,
                y_spk, y_phn):
        output_spk = self.loss_spk(emb_spk1, emb_spk2, y_spk)
        output_phn = self.loss_phn(emb_phn1, emb_phn2, y_phn)
        output = self.weight * output_spk + (1.0 - self.weight) * output_phn
        return output

