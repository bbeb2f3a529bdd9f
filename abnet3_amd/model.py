"""SiameseNetwork on MI355X: the class surface of abnet3/model.py, HIP inside.

Mirrors (file:line relative to the reference checkout)
  NetworkBuilder   abnet3/model.py:30-79
  SiameseNetwork   abnet3/model.py:82-208   (same constructor kwargs, asserts,
                   state_dict key names, init order and RNG consumption, so one
                   torch.manual_seed gives the reference's initial weights and
                   reference .pth files load unchanged)
The nn.Linear / nn.BatchNorm1d modules are kept as parameter HOLDERS only; the
arithmetic of forward_once / forward and of their autograd runs in
libabnet3_hip.so (abn_tower_forward / abn_tower_backward).  There is no CPU
path: calling the network with CPU tensors raises.
"""
import os

import torch
import torch.nn as nn

from . import _lib

activation_functions = {'relu': nn.ReLU,
                        'sigmoid': nn.Sigmoid,
                        'tanh': nn.Tanh,
                        'softmax': nn.Softmax,
                        }

init_functions = {'xavier_uni': nn.init.xavier_uniform_,
                  'xavier_normal': nn.init.xavier_normal_,
                  'orthogonal': nn.init.orthogonal_}

_ALIGN = 64     # floats: every tensor of the flat buffers starts 256-B aligned


class NetworkBuilder(nn.Module):
    """Generic Neural Network Model class (abnet3/model.py:30-79)."""

    def __init__(self, *args, **kwargs):
        super(NetworkBuilder, self).__init__()

    def forward_once(self, *args, **kwargs):
        raise NotImplementedError('Unimplemented forward_once for class:',
                                  self.__class__.__name__)

    def forward(self, *args, **kwargs):
        raise NotImplementedError('Unimplemented forward for class:',
                                  self.__class__.__name__)

    def whoami(self, *args, **kwargs):
        raise NotImplementedError('Unimplemented whoami for class:',
                                  self.__class__.__name__)

    def save_network(self, *args, **kwargs):
        raise NotImplementedError('Unimplemented save_network for class:',
                                  self.__class__.__name__)

    def load_network(self, *args, **kwargs):
        raise NotImplementedError('Unimplemented load_network for class:',
                                  self.__class__.__name__)

    def init_weight_method(self, *args, **kwargs):
        raise NotImplementedError('Unimplemented init_weight_method' +
                                  'for class:',
                                  self.__class__.__name__)

    def plot_network(self, *args, **kwargs):
        raise NotImplementedError('Unimplemented plot_network for class:',
                                  self.__class__.__name__)


class _Segment(object):
    """A run of [Linear, Dropout, (BatchNorm1d), act] blocks executed by one
    abn_tower_forward / abn_tower_backward launch sequence: the whole
    SiameseNetwork, or the trunk / one head of a SiameseMultitaskNetwork."""

    def __init__(self, net, blocks, first_block, act, last_act, batch_norm):
        self.net = net
        self.blocks = blocks                  # [(Linear, BatchNorm1d | None)]
        self.first_block = first_block        # index into the network's dropout masks
        self.act, self.last_act = act, last_act
        self.batch_norm = bool(batch_norm)
        self.input_dim = blocks[0][0].in_features
        self.output_dim = blocks[-1][0].out_features
        self.params = []
        for lin, bn in blocks:
            self.params += [lin.weight, lin.bias]
            if bn is not None:
                self.params += [bn.weight, bn.bias]

    def bn_modules(self):
        return [bn for _, bn in self.blocks if bn is not None]

    def offsets(self):
        """Float offsets of self.params in the network's flat buffers (the live
        parameters are the segments' parameters, concatenated in segment order)."""
        net = self.net
        first = 0
        for seg in net._segment_list():
            if seg is self:
                break
            first += len(seg.params)
        return net._offsets[first:first + len(self.params)]

    def masks_of(self, all_masks):
        if isinstance(all_masks, _DropoutInKernel):
            return all_masks
        if all_masks is None:
            return None
        return all_masks[self.first_block:self.first_block + len(self.blocks)]

    def _template(self):
        """The descriptor with everything but gradient and mask pointers filled in,
        rebuilt only when the parameters were re-homed (a step calls this three
        times; filling ~100 ctypes fields each time was a quarter of the eager
        step's host cost)."""
        net = self.net
        net.flat_parameters()
        sync = getattr(net, 'bn_sync', None) if self.batch_norm else None
        key = (net._flat.data_ptr(), net._generation, net.precision, id(sync))
        cached = getattr(self, '_desc_cache', None)
        if cached is not None and cached[0] == key:
            return cached[1]
        d = _lib.TowerDesc()
        d.n_layers = len(self.blocks)
        d.act = _lib.ACT[self.act]
        if self.last_act not in _lib.ACT:
            raise NotImplementedError('abnet3_amd: last_non_linearity=%r is not on '
                                      'the accelerated path' % (self.last_act,))
        d.last_act = _lib.ACT[self.last_act]
        d.batch_norm = int(self.batch_norm)
        d.precision = _lib.PRECISION[net.precision]
        if sync is not None and sync.world >= 1:      # cross-replica BatchNorm statistics (parallel.BatchNormSync)
            d.bn_sync_world = sync.world
            d.bn_sync_fn = sync.fn
            d.bn_sync_ctx = getattr(sync, 'ctx', None)
        d.dims[0] = self.input_dim
        grad_slots = []                      # (field, layer, float offset in the flat buffers)
        offs = self.offsets()                # in the order of self.params
        gi = 0
        for l, (lin, bn) in enumerate(self.blocks):
            d.dims[l + 1] = lin.out_features
            d.W[l] = lin.weight.data_ptr()
            d.b[l] = lin.bias.data_ptr()
            grad_slots += [('dW', l, offs[gi]), ('db', l, offs[gi + 1])]
            gi += 2
            if bn is not None:
                d.bn_w[l] = bn.weight.data_ptr()
                d.bn_b[l] = bn.bias.data_ptr()
                d.bn_rm[l] = bn.running_mean.data_ptr()
                d.bn_rv[l] = bn.running_var.data_ptr()
                if bn.num_batches_tracked is not None and bn.num_batches_tracked.is_cuda:
                    d.bn_nbt[l] = bn.num_batches_tracked.data_ptr()      # (the training forward counts its calls itself)
                grad_slots += [('dbn_w', l, offs[gi]), ('dbn_b', l, offs[gi + 1])]
                gi += 2
        # a persistent image of the weights as operand fragments (abn_tower_desc.wpack): the forward skips its
        # pack launch while the image is known to match the parameters (_weights_key)
        self._wpack, self._wpack_key = None, None
        if net._flat.is_cuda and os.environ.get('ABN_WPACK') != '0':      # (the variable: A/B runs)
            n = _lib.load().abn_tower_wpack_floats(_lib.C.byref(d))
            if n > 0:
                self._wpack = torch.zeros(n, dtype=torch.float32, device=net._flat.device)
                d.wpack = self._wpack.data_ptr()
        # the resident BatchNorm tower's sync buffer (abn_tower_desc.sync_ws): zero once, the library's from then on
        self._sync_ws = None
        if self.batch_norm and net._flat.is_cuda and os.environ.get('ABN_BN_SYNC_WS') != '0' and not getattr(self, '_sync_disabled', False):
            self._sync_ws = torch.zeros(_lib.load().abn_tower_sync_ws_bytes() // 4, dtype=torch.int32, device=net._flat.device)
            d.sync_ws = self._sync_ws.data_ptr()
        self._desc_cache = (key, d, grad_slots)
        return d

    # -- the resident BatchNorm tower's failure word (csrc/tower_bn_persist.h): a hand-over that is not met within its bounded
    # spin -- part of the grid was not on the chip: another process, stream or resident kernel held CUs -- sets word 16 of
    # the sync buffer, the launches drain with NaN outputs and the optimizer's launch drops the step (parameters untouched).
    # The word is sticky by design; whoever owns the buffer looks at it, clears it and takes the tower off the resident path.
    def sync_fail_word(self):
        """1-element int32 view of the failure word, or None (no sync buffer lent)."""
        ws = getattr(self, '_sync_ws', None)
        return None if ws is None else ws[16:17]

    def recover_sync(self):
        """Clears the sync buffer and keeps this tower on the layer launches (ABN_PATH_BN_LAYERS) from now on."""
        ws = getattr(self, '_sync_ws', None)
        if ws is None:
            return
        ws.zero_()
        self._sync_disabled = True
        cached = getattr(self, '_desc_cache', None)
        if cached is not None:
            cached[1].sync_ws = None

    def _weights_key(self):
        net = self.net
        # (writes through p.data / the flat buffer do not bump p._version: the flat buffer's own counter and
        # the network's `_weights_epoch`, bumped by whoever writes behind torch, cover them)
        return (net._flat.data_ptr(), net._generation, net.precision, net._flat._version,
                getattr(net, '_weights_epoch', 0), tuple(p._version for p in self.params))

    def wpack_state(self):
        """(valid flag for the next call's descriptor, key of the weights as they stand)"""
        if getattr(self, '_wpack', None) is None:
            return 0, None
        if torch.cuda.is_current_stream_capturing():     # a captured call is replayed with other weights: always rebuild
            self._wpack_key = None
            return 0, None
        key = self._weights_key()
        return int(self._wpack_key == key), key

    def wpack_matches(self, key):
        """the image was (re)built from, or kept in step with, the weights `key` describes"""
        if getattr(self, '_wpack', None) is not None:
            self._wpack_key = key

    def descriptor(self, with_grads, grad_buf=None, masks=None, d_out_is_dz=False, defer_reduce=False, forward_only=False):
        """abn_tower_desc for one call.  grad_buf: the flat gradient buffer of this
        backward pass (gradients land at the parameters' offsets in it)."""
        tmpl = self._template()
        valid, self._key_at_descriptor = self.wpack_state()
        if not with_grads and masks is None:
            tmpl.wpack_valid = valid
            tmpl.forward_only = int(forward_only)
            return tmpl                        # read-only for the library
        d = _lib.TowerDesc.from_buffer_copy(tmpl)
        d.wpack_valid = valid
        d.forward_only = int(forward_only)
        d.d_out_is_dz = int(d_out_is_dz)
        d.defer_reduce = int(defer_reduce)
        if isinstance(masks, _DropSeed):
            d.drop_seed = masks.seed.data_ptr()
            d.drop_p = masks.p
        elif masks is not None:
            for l, m in enumerate(masks):
                d.drop_mask[l] = m.data_ptr()
        if with_grads:
            base = grad_buf.data_ptr()
            for field, l, off in self._desc_cache[2]:
                getattr(d, field)[l] = base + 4 * off
        return d


class _GradPass(object):
    """The flat gradient buffer of ONE backward pass, shared by the segments of a
    network: a fresh buffer per forward (caching allocator: no memset, no sync).
    autograd takes the per-parameter views as p.grad without copying, so after one
    backward the whole gradient sits in one buffer laid out like the flat
    parameters.  A segment that runs backward a second time (retain_graph) gets a
    private buffer, and autograd adds it into the first as for any other module."""

    def __init__(self, net):
        self.net = net
        self.buf = None
        self.used = set()

    def views(self, seg):
        """(flat gradient buffer, its per-parameter views for `seg`)."""
        net = self.net
        net.flat_parameters()
        if self.buf is None or id(seg) in self.used:
            buf = torch.empty_like(net._flat)
            if self.buf is None:
                self.buf = buf
                net._last_grad_flat = buf
        else:
            buf = self.buf
        self.used.add(id(seg))
        specs = getattr(seg, '_view_specs', None)
        if specs is None or specs[0] != net._flat.data_ptr():
            specs = seg._view_specs = (net._flat.data_ptr(),
                                       [(tuple(p.shape), tuple(p.stride()), off)
                                        for p, off in zip(seg.params, seg.offsets())])
        return buf, [buf.as_strided(shape, stride, off) for shape, stride, off in specs[1]]


class _DropoutInKernel(object):
    """What _draw_dropout_masks hands down when no mask tensors were asked for: nn.Dropout(p) to be drawn
    inside the kernels (abn_tower_desc.drop_seed) wherever the library can, as tensors elsewhere."""
    __slots__ = ('p', 'net')

    def __init__(self, p, net):
        self.p, self.net = p, net


class _DropSeed(object):
    """The in-kernel dropout of ONE forward of one segment: a device uint64 drawn by torch's generator (so that
    a captured step redraws it at every replay) and p.  The backward gets the same object."""
    __slots__ = ('seed', 'p')

    def __init__(self, p, device):
        self.seed = torch.empty(1, dtype=torch.int64, device=device).random_()
        self.p = float(p)


class _Saved(object):
    """What a segment's forward leaves for its backward."""
    __slots__ = ('x1', 'x2', 'ws', 'masks', 'n_calls', 'train', 'rows', 'bn_synced', 'n_valid', 'source')


def _segment_forward(seg, all_masks, n_calls, x1, x2, forward_only=False, n_valid=None, source=None):
    """Raw forward of one segment (no autograd): the launch sequence of
    abn_tower_forward.  Returns ([rows, out] embeddings as a view of the workspace,
    _Saved).  n_valid (device int32 tensor): a padded batch through a BatchNorm tower in training -- only the
    first n_valid rows of every call are real (abn_tower_desc.n_valid).  source (_lib.StepSource): the first layer's launch
    stages its rows from the pass's plan instead of from x1 / x2 (abn_tower_desc.source; HipLibraryError where the library
    does not take the call that way: the caller asks abn_tower_path first)."""
    lib = _lib.load()
    net = seg.net
    # the reference takes strided inputs (a column slice of stacked features):
    # make them dense first, then insist on the device
    x1 = x1.contiguous()
    x2 = x2.contiguous() if x2 is not None else None
    _lib.require_device(x1, x2)
    if x1.dtype != torch.float32 or (x2 is not None and x2.dtype != torch.float32):
        raise TypeError('abnet3_amd: features must be float32 (the reference '
                        'casts them, abnet3/utils.py:228-235)')
    if x1.dim() != 2 or x1.shape[1] != seg.input_dim:
        raise ValueError('abnet3_amd: expected input of shape [n, %d], got %s'
                         % (seg.input_dim, tuple(x1.shape)))
    if x2 is not None and x2.shape != x1.shape:
        raise ValueError('abnet3_amd: the two inputs must have the same shape')
    train = bool(net.training)
    rows = x1.shape[0] * (2 if x2 is not None else 1)
    masks = seg.masks_of(all_masks) if train else None
    if isinstance(masks, _DropoutInKernel):
        # drawn inside the kernels where they are the operand-plane ones, as tensors for the per-layer path.
        # The probe carries the call's own forward_only: a BatchNorm tower in train mode under torch.no_grad()
        # (TrainerBuilder.train's first pass, abnet3/trainer.py:137) runs on the per-layer kernels
        probe = seg.descriptor(with_grads=False, forward_only=forward_only)
        if lib.abn_tower_uses_planes(_lib.C.byref(probe), rows, _lib.ptr(x1), _lib.ptr(x2), _lib.ptr(x1), 1) == 1:
            masks = _DropSeed(masks.p, x1.device)
        else:
            masks = seg.masks_of(masks.net._draw_mask_tensors(rows, x1.device))
    desc = seg.descriptor(with_grads=False, masks=masks, forward_only=forward_only)
    if n_valid is not None and train and seg.batch_norm:
        desc = _lib.TowerDesc.from_buffer_copy(desc)
        desc.n_valid = n_valid.data_ptr()
    else:
        n_valid = None                   # (rows that do not see each other: nothing to tell the library)
    synced = False
    if train and seg.batch_norm and desc.bn_sync_fn:
        # Cross-replica statistics exist on the one-launch-per-layer BatchNorm kernels, and those are the kernels of
        # a forward that keeps what a backward needs.  A train-mode pass under torch.no_grad() (TrainerBuilder.train's
        # first pass, abnet3/trainer.py:137) asks for them too: the replicas' running statistics must move together.
        # Where the library would not take that path (fewer than 256 rows: OriginalDataLoader's short ragged batches;
        # odd widths) the call falls back to per-replica statistics, said once -- not an error.
        # The choice is the GROUP's: the ranks hold batches of different sizes (OriginalDataLoader's ragged word-pair batches),
        # and a rank that left the exchange on its own would leave the others waiting in it -- or add its sums to another
        # layer's.  One small all-reduce (MIN) per training forward settles it; a step being captured into a graph replays the
        # answer of the eager step before it (its shapes are the captured ones on every rank).
        desc = _lib.TowerDesc.from_buffer_copy(desc)
        desc.forward_only = 0
        mine = lib.abn_tower_path(_lib.C.byref(desc), _lib.ptr(x1), _lib.ptr(x2), rows, n_calls, 1, _lib.ptr(x1), 0, None) in (_lib.PATH_BN_LAYERS, _lib.PATH_BN_TOWER)
        if torch.cuda.is_current_stream_capturing():
            agreed = bool(getattr(net, '_bn_sync_agreed', {}).get(id(seg), False)) and mine      # (this segment's own answer)
        else:
            from . import parallel
            agreed = parallel.all_agree(mine, getattr(net.bn_sync, 'group', None)) if getattr(net.bn_sync, 'collective', False) else mine
            if not isinstance(getattr(net, '_bn_sync_agreed', None), dict):
                net._bn_sync_agreed = {}
            net._bn_sync_agreed[id(seg)] = agreed
        if agreed:
            synced = True
        else:
            desc.forward_only = int(forward_only)
            desc.bn_sync_fn, desc.bn_sync_world = None, 0
            if not getattr(net, '_warned_bn_sync_fallback', False):
                net._warned_bn_sync_fallback = True
                import warnings
                warnings.warn('abnet3_amd: sync_batch_norm: a batch of %d rows (here; the ranks decide together) does not run on the '
                              'per-layer BatchNorm launches that carry the cross-replica statistics on every rank; such steps use '
                              'per-replica statistics' % rows)
    if source is not None:
        desc = _lib.TowerDesc.from_buffer_copy(desc)
        desc.source = _lib.C.addressof(source)
    ws_floats = lib.abn_tower_ws_floats(_lib.C.byref(desc), rows, n_calls)
    if ws_floats < 0:
        _lib.check(-1, 'abn_tower_ws_floats')
    ws = torch.empty(max(ws_floats, 1), dtype=torch.float32, device=x1.device)
    if synced:
        net.bn_sync.buffers = [ws]
    _lib.check(lib.abn_tower_forward(_lib.C.byref(desc), _lib.ptr(x1), _lib.ptr(x2),
                                     rows, n_calls, int(train), _lib.ptr(ws),
                                     _lib.stream()), 'abn_tower_forward')
    _lib.note_path(desc, x1, x2, rows, n_calls, train, ws, backward=False)
    if desc.wpack and not desc.wpack_valid and seg._key_at_descriptor is not None:
        # the operand-plane kernels rebuilt the persistent weight image (the per-layer path never touches it)
        if lib.abn_tower_uses_planes(_lib.C.byref(desc), rows, _lib.ptr(x1), _lib.ptr(x2), _lib.ptr(ws), int(train)) == 1:
            seg.wpack_matches(seg._key_at_descriptor)
    if train and seg.batch_norm:        # (the counters the library was not given: it adds n_calls to the others itself)
        left = [bn.num_batches_tracked for l, bn in enumerate(seg.bn_modules()) if not desc.bn_nbt[l]]
        if left:
            torch._foreach_add_(left, n_calls)
    off = lib.abn_tower_out_offset(_lib.C.byref(desc), rows, n_calls)
    out = ws[off:off + rows * seg.output_dim].view(rows, seg.output_dim)
    sv = _Saved()
    sv.x1, sv.x2, sv.ws, sv.masks, sv.n_calls, sv.train, sv.rows = x1, x2, ws, masks, n_calls, train, rows
    sv.bn_synced = synced
    sv.n_valid = n_valid
    sv.source = source
    return out, sv


def _lend_forward_workspace(desc, sv, rows):
    """A deferred backward on the layer-per-launch kernels lends the forward's workspace to abn_tower_reduce_step
    (abn_tower_desc.fwd_ws): the weight gradients then wait for the optimizer's launch and are computed THERE, over all rows
    and with the update rule, in one launch (csrc/tower_wgrad_step.h).  Only where the call really takes that path."""
    lib = _lib.load()
    path = lib.abn_tower_path(_lib.C.byref(desc), _lib.ptr(sv.x1), _lib.ptr(sv.x2), rows, sv.n_calls, 1, _lib.ptr(sv.ws), 1, None)
    if path == _lib.PATH_WIDE:
        desc.fwd_ws = sv.ws.data_ptr()
        desc.fwd_calls = sv.n_calls


def _segment_backward(seg, sv, d_out, grad_pass, need_dx, d_out_is_dz=False, defer_reduce=False):
    """Raw backward of one segment: abn_tower_backward into the pass's flat gradient
    buffer.  Returns (per-parameter gradient views, dx or None).  With defer_reduce the
    weight gradients stay unreduced in the scratch and the third return value is what
    abn_tower_reduce_step needs to finish: (descriptor, rows, scratch, scratch floats, buffer)."""
    lib = _lib.load()
    if seg.batch_norm and not sv.train:
        raise NotImplementedError(
            'abnet3_amd: backward through an eval-mode BatchNorm forward is '
            'not on the accelerated path (the reference trains in train mode, '
            'abnet3/trainer.py:234)')
    d_out = d_out.contiguous()
    _lib.require_device(d_out)
    rows = d_out.shape[0]
    grad_buf, grads = grad_pass.views(seg)
    desc = seg.descriptor(with_grads=True, grad_buf=grad_buf, masks=sv.masks, d_out_is_dz=d_out_is_dz,
                          defer_reduce=defer_reduce)
    if defer_reduce and not need_dx:
        _lend_forward_workspace(desc, sv, rows)
    scratch_floats = lib.abn_tower_bwd_scratch_floats(_lib.C.byref(desc), rows)
    scratch = torch.empty(max(scratch_floats, 1), dtype=torch.float32, device=d_out.device)
    dx = torch.empty(rows, seg.input_dim, dtype=torch.float32,
                     device=d_out.device) if need_dx else None
    if desc.bn_sync_fn and not sv.bn_synced:      # the forward fell back to per-replica statistics: so does its backward
        desc.bn_sync_fn, desc.bn_sync_world = None, 0
    if desc.bn_sync_fn:
        seg.net.bn_sync.buffers = [sv.ws, scratch]
    if sv.n_valid is not None:                    # (a padded batch through BatchNorm: the forward's real-row count)
        desc.n_valid = sv.n_valid.data_ptr()
    _lib.check(lib.abn_tower_backward(
        _lib.C.byref(desc), _lib.ptr(sv.x1), _lib.ptr(sv.x2), _lib.ptr(d_out), rows,
        sv.n_calls, _lib.ptr(sv.ws), _lib.ptr(scratch), scratch_floats,
        _lib.ptr(dx), _lib.stream()), 'abn_tower_backward')
    _lib.note_path(desc, sv.x1, sv.x2, rows, sv.n_calls, True, sv.ws, backward=True)
    if defer_reduce:
        return grads, dx, (desc, rows, scratch, scratch_floats, grad_buf, seg)
    return grads, dx


class _TowerFunction(torch.autograd.Function):
    """One launch sequence for n_calls forward_once calls of one segment."""

    @staticmethod
    def forward(ctx, seg, grad_pass, all_masks, n_calls, split, infer, x1, x2, *params):
        # parameters are views of the flat buffer (dense by construction); only the
        # device is checked here
        for p_ in params:
            if not p_.is_cuda:
                _lib.require_device(p_)
        # nothing requires a gradient (torch.no_grad() -- needs_input_grad does not see it, and grad mode is
        # always off in here: the caller passes it -- or frozen parameters): inference, the library keeps
        # nothing for a backward
        out, sv = _segment_forward(seg, all_masks, n_calls, x1, x2,
                                   forward_only=infer or not any(ctx.needs_input_grad))
        ctx.seg, ctx.grad_pass, ctx.sv = seg, grad_pass, sv
        ctx.have_x2 = x2 is not None
        ctx.split = split
        if split:        # the two towers' embeddings as two outputs (no slice nodes)
            half = sv.rows // 2
            return out[:half], out[half:]
        return out

    @staticmethod
    def backward(ctx, *d_outs):
        seg = ctx.seg
        if ctx.split:
            d1, d2 = d_outs
            if d1 is None or d2 is None:
                z = d1 if d1 is not None else d2
                d1 = d1 if d1 is not None else torch.zeros_like(z)
                d2 = d2 if d2 is not None else torch.zeros_like(z)
            # the pair loss hands back two halves of ONE buffer: take it whole
            if (d1.is_contiguous() and d2.is_contiguous() and d1._base is not None
                    and d1._base is d2._base and d1._base.is_contiguous()
                    and d1._base.numel() == 2 * d1.numel()
                    and d1.data_ptr() == d1._base.data_ptr()
                    and d2.data_ptr() == d1.data_ptr() + d1.numel() * 4):
                d_out = d1._base.view(2 * d1.shape[0], d1.shape[1])
            else:
                d_out = torch.cat([d1, d2])
        else:
            d_out = d_outs[0]
        need_dx = ctx.needs_input_grad[6] or (ctx.have_x2 and ctx.needs_input_grad[7])
        grads, dx = _segment_backward(seg, ctx.sv, d_out, ctx.grad_pass, need_dx)
        dx1 = dx2 = None
        if need_dx:
            rows = d_out.shape[0]
            if ctx.have_x2:
                dx1, dx2 = dx[:rows // 2], dx[rows // 2:]
            else:
                dx1 = dx
        return (None, None, None, None, None, None, dx1, dx2) + tuple(grads)


class _SoftmaxRows(torch.autograd.Function):
    """nn.Softmax() over the rows of a [n, width] matrix (the 'softmax' choice of
    last_non_linearity, abnet3/model.py:161-166): abn_softmax_rows and its autograd."""

    @staticmethod
    def forward(ctx, z):
        lib = _lib.load()
        z = z.contiguous()
        _lib.require_device(z)
        out = torch.empty_like(z)
        _lib.check(lib.abn_softmax_rows(_lib.ptr(z), z.shape[0], z.shape[1], _lib.ptr(out),
                                        _lib.stream()), 'abn_softmax_rows')
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, da):
        lib = _lib.load()
        (a,) = ctx.saved_tensors
        da = da.contiguous()
        dz = torch.empty_like(da)
        _lib.check(lib.abn_softmax_rows_backward(_lib.ptr(a), _lib.ptr(da), a.shape[0], a.shape[1],
                                                 _lib.ptr(dz), _lib.stream()),
                   'abn_softmax_rows_backward')
        return dz


def _blocks_of(*sequentials):
    """[(Linear, BatchNorm1d | None)] of nn.Sequentials in forward order."""
    out = []
    for seq in sequentials:
        lin = None
        for m in seq:
            if isinstance(m, nn.Linear):
                if lin is not None:
                    out.append((lin, None))
                lin = m
            elif isinstance(m, nn.BatchNorm1d):
                out.append((lin, m))
                lin = None
        if lin is not None:
            out.append((lin, None))
    return out


class _HipNetwork(NetworkBuilder):
    """What the HIP-backed networks share: the flat parameter / gradient buffers
    (one optimizer launch and one RCCL all-reduce per step instead of one per
    tensor), dropout masks and the segment launcher.  Subclasses provide
    _segments() and the reference's constructor / forward surface."""

    def _init_hip_state(self):
        self._flat = None
        self._last_grad_flat = None
        self._offsets = None
        self._segs = None
        self._mask_override = None      # tests: fixed dropout masks, one per live block
        self._generation = getattr(self, '_generation', 0)
        self._live_cache = None
        # Arithmetic of the tower GEMMs (abn_tower_desc.precision):
        #   'f16x2' (default)   every fp32 operand scaled by a power of two (per operand row / 32-row weight block)
        #                       and split into two fp16 terms (22 significant bits), three fp16 MFMA products per
        #                       operand pair, fp32 accumulation: as close to a float64 evaluation as the
        #                       reference's own fp32 (measured), passes every golden test at the 1e-5 bar;
        #   'bf16x3'            every fp32 operand split into three bf16 terms (24 bits, no scales), six bf16 MFMA
        #                       products per operand pair: same grade, ~20 % slower;
        #   'fp32'              exact-fp32 MFMA (v_mfma_f32_32x32x2_f32): one fp32 fma chain per
        #                       output, bit-for-bit what a CPU computes in that order;
        #   'bf16'              operands rounded to bf16 once: ~3 digits, outside the parity bar.
        self.precision = os.environ.get('ABNET3_PRECISION', 'f16x2')
        assert self.precision in _lib.PRECISION, self.precision

    def _segments(self):
        raise NotImplementedError

    def _segment_list(self):
        if self._segs is None:
            self._segs = self._segments()
        return self._segs

    def _blocks(self):
        return [b for seg in self._segment_list() for b in seg.blocks]

    def _bn_modules(self):
        return [bn for _, bn in self._blocks() if bn is not None]

    def live_parameters(self):
        """The parameters forward() uses (all of them, except the never-called
        branches of a SiameseMultitaskNetwork), in nn.Module.parameters() order:
        what the flat buffers hold and the optimizer updates."""
        return [p for seg in self._segment_list() for p in seg.params]

    def flatten_parameters(self):
        """Re-homes every live parameter into one flat fp32 buffer (gradients come
        back in a buffer of the same layout).  Called lazily; redone automatically
        when .cuda()/.to() replaced the storages."""
        params = self.live_parameters()
        dev = params[0].device
        offs, o = [], 0
        for p in params:
            offs.append(o)
            o += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        flat = torch.zeros(o, dtype=torch.float32, device=dev)
        for p, off in zip(params, offs):
            flat[off:off + p.numel()].copy_(p.data.reshape(-1))
            p.data = flat[off:off + p.numel()].view(p.shape)
        self._flat = flat
        self._offsets = offs
        self._live_cache = params
        return flat

    def _apply(self, fn, *args, **kwargs):
        # .cuda() / .to() / .float() replace the parameter storages: the flat buffer
        # and every cached descriptor are stale afterwards
        out = super(_HipNetwork, self)._apply(fn, *args, **kwargs)
        # a no-op move (.cuda() on a network that already lives there, as the embedders
        # do between epochs) leaves every parameter a view of the flat buffer: keep it,
        # the optimizer state and captured graphs hang on its address
        if getattr(self, '_flat', None) is not None and self._live_cache is not None \
                and self._flat.device == self._live_cache[0].device and self._is_flat(full=True):
            return out
        self._flat = None
        self._generation = getattr(self, '_generation', 0) + 1
        return out

    def _is_flat(self, full=False):
        """Are the live parameters views of self._flat?  _apply() invalidates the
        buffer itself; the per-call check looks at the first and the last parameter
        only (someone re-assigning p.data by hand), `full` at all of them."""
        if self._flat is None:
            return False
        base = self._flat.data_ptr()
        params = self._live_cache
        if full:
            return all(p.data_ptr() == base + 4 * off for p, off in zip(params, self._offsets))
        return (params[0].data_ptr() == base + 4 * self._offsets[0] and
                params[-1].data_ptr() == base + 4 * self._offsets[-1])

    def flat_parameters(self):
        if not self._is_flat():
            self.flatten_parameters()
        return self._flat

    def grads_in_flat_buffer(self):
        """True when every live p.grad IS its view of one flat gradient buffer."""
        buf = getattr(self, '_last_grad_flat', None)
        if buf is None or not self._is_flat():
            return False
        base = buf.data_ptr()
        return all(p.grad is not None and p.grad.data_ptr() == base + 4 * off
                   for p, off in zip(self.live_parameters(), self._offsets))

    def flat_grad(self):
        """The gradient of all live parameters as one flat fp32 buffer laid out
        like flat_parameters() (the all-reduce bucket).  Zero-copy after a single
        backward; otherwise p.grad tensors are packed into a new buffer."""
        if self.grads_in_flat_buffer():
            return self._last_grad_flat
        self.flat_parameters()
        buf = torch.zeros_like(self._flat)
        for p, off in zip(self.live_parameters(), self._offsets):
            if p.grad is not None:
                buf[off:off + p.numel()].copy_(p.grad.reshape(-1))
                p.grad = buf[off:off + p.numel()].view(p.shape)
        self._last_grad_flat = buf
        return buf

    def _draw_dropout_masks(self, rows, device):
        """nn.Dropout(p) of every block in train mode (model.py:137,148,157): one
        [rows, width] multiplier per layer, 0 with probability p else 1/(1-p),
        drawn with torch's device generator (the reference's CPU stream cannot be
        reproduced on a GPU; the arithmetic that consumes the mask is what the
        parity tests pin).  None when p == 0."""
        if self._mask_override is not None:
            return self._mask_override
        p = float(self.p_dropout)
        if p <= 0.0:
            return None
        if os.environ.get('ABN_DROPOUT_IN_KERNEL') != '0':      # (the variable: A/B runs)
            return _DropoutInKernel(p, self)
        return self._draw_mask_tensors(rows, device)

    def _draw_mask_tensors(self, rows, device):
        p = float(self.p_dropout)
        # all layers' masks in ONE buffer: two launches per forward (bernoulli, scale)
        # instead of three per layer; every mask starts 256-byte aligned
        widths = [lin.out_features for lin, _ in self._blocks()]
        offs, o = [], 0
        for w in widths:
            offs.append(o)
            o += (rows * w + _ALIGN - 1) // _ALIGN * _ALIGN
        flat = torch.empty(o, dtype=torch.float32, device=device)
        if p < 1.0:
            flat.bernoulli_(1.0 - p).mul_(1.0 / (1.0 - p))
        else:
            flat.zero_()
        return [flat[off:off + rows * w].view(rows, w) for off, w in zip(offs, widths)]

    def resident_tower_failed(self):
        """Did a resident BatchNorm tower launch of this network give up since the last look?  ONE device-to-host read (it
        synchronises with the stream): callers look where they synchronise anyway -- the end of a pass -- or every few
        hundred steps.  A tower that did is cleared and kept on the layer launches from here on (a warning says so once);
        the steps it dropped changed no parameter (the optimizer's launch skips them).  Returns the number of towers hit."""
        words = [(seg, seg.sync_fail_word()) for seg in self._segment_list()]
        words = [(seg, w) for seg, w in words if w is not None and not getattr(seg, '_sync_disabled', False)]
        if not words:
            return 0
        got = torch.cat([w for _, w in words]).cpu()
        hit = 0
        for (seg, _), v in zip(words, got.tolist()):
            if v != 0:
                seg.recover_sync()
                hit += 1
        if hit:
            import warnings
            warnings.warn('abnet3_amd: a resident BatchNorm tower launch gave up on a hand-over between workgroups (the '
                          'GPU is shared with another process, stream or resident kernel?).  The steps since were dropped '
                          '-- no parameter changed, their losses are not numbers --; the sync buffer is cleared and this '
                          'network trains on one launch per layer from here on (ABN_BN_PERSIST=0 does so from the start).',
                          RuntimeWarning, stacklevel=2)
        return hit

    def weights_changed_behind_torch(self):
        """Called by whoever rewrites the parameters without torch noticing (abn_optimizer_step, a replayed
        hipGraph, a broadcast into the flat buffer, writes through p.data): the persistent weight images
        are stale."""
        self._weights_epoch = getattr(self, '_weights_epoch', 0) + 1
        for seg in (getattr(self, '_segs', None) or ()):
            seg._wpack_key = None

    _weights_rewritten = weights_changed_behind_torch

    def _run(self, seg, grad_pass, masks, x1, x2, n_calls, split):
        if x1.shape[0] == 0:
            # nn.Linear on zero rows gives zero rows (and nothing to launch)
            _lib.require_device(x1)
            if x1.dim() != 2 or x1.shape[1] != seg.input_dim:
                raise ValueError('abnet3_amd: expected input of shape [n, %d], got %s'
                                 % (seg.input_dim, tuple(x1.shape)))
            empty = x1.new_zeros((0, seg.output_dim))
            return (empty, empty.clone()) if split else empty
        return _TowerFunction.apply(seg, grad_pass, masks, n_calls, split, not torch.is_grad_enabled(), x1, x2,
                                    *seg.params)

    # HIP plumbing kept in __dict__ next to the reference's attributes: device
    # buffers, ctypes descriptors and caches -- not part of the description of the
    # network and not picklable
    _HIP_STATE = ('_flat', '_last_grad_flat', '_offsets', '_segs', '_mask_override',
                  '_generation', '_live_cache', '_weights_epoch', '_pending_reduce', '_fused_loss_refused',
                  '_pending_lower', 'bn_sync', '_bn_sync_agreed')

    def whoami(self):
        """Output description for the neural network and all parameters
        (abnet3/model.py:198-202: {'params': self.__dict__, 'class_name'}), minus the
        HIP plumbing, so that TrainerBuilder.save_whoami can pickle it as the
        reference does (abnet3/trainer.py:106-108)."""
        params = {k: v for k, v in self.__dict__.items() if k not in self._HIP_STATE}
        return {'params': params, 'class_name': self.__class__.__name__}

    def load_network(self, network_path=None):
        self.load_state_dict(torch.load(network_path))


class SiameseNetwork(_HipNetwork):
    """Siamese neural network Architecture (abnet3/model.py:82-208).

    Parameters: see the reference docstring; identical names and defaults.
    """

    def __init__(self, input_dim=None, num_hidden_layers=None, hidden_dim=None,
                 output_dim=None, p_dropout=0.1, batch_norm=False,
                 type_init='xavier_uni', activation_layer=None,
                 output_path=None, last_non_linearity="default"):
        super(SiameseNetwork, self).__init__()
        assert activation_layer in ('relu', 'sigmoid', 'tanh')
        assert type_init in ('xavier_uni', 'xavier_normal', 'orthogonal')
        assert type(input_dim) == int, 'input dim should be int'
        assert type(hidden_dim) == int, 'hidden dim should be int'
        assert type(num_hidden_layers) == int, 'num hidden lay should be int'
        assert type(output_dim) == int, 'output dim should be int'
        assert num_hidden_layers + 2 <= _lib.MAX_LAYERS, 'too many layers'

        self.input_dim = input_dim
        self.num_hidden_layers = num_hidden_layers
        self.hidden_dim = hidden_dim
        self.output_dim = output_dim
        self.activation_layer = activation_layer
        self.batch_norm = batch_norm
        self.type_init = type_init
        self.last_non_linearity = last_non_linearity
        self.p_dropout = p_dropout

        activation = activation_functions[activation_layer]

        # same module order as the reference => same state_dict keys and the
        # same RNG consumption during construction
        input_layer = [nn.Linear(input_dim, hidden_dim), nn.Dropout(p=p_dropout)]
        if self.batch_norm:
            input_layer.append(nn.BatchNorm1d(hidden_dim))
        input_layer.append(activation())
        self.input_emb = nn.Sequential(*input_layer)

        hidden = []
        for idx in range(self.num_hidden_layers):
            hidden.append(nn.Linear(hidden_dim, hidden_dim))
            hidden.append(nn.Dropout(p=p_dropout))
            if self.batch_norm:
                hidden.append(nn.BatchNorm1d(hidden_dim))
            hidden.append(activation())
        self.hidden_layers = nn.Sequential(*hidden)

        output_layer = [nn.Linear(hidden_dim, output_dim), nn.Dropout(p=p_dropout)]
        if self.batch_norm:
            output_layer.append(nn.BatchNorm1d(output_dim))
        if self.last_non_linearity == "default":
            output_layer.append(activation())
            self._last_act = activation_layer
        elif self.last_non_linearity is None:
            self._last_act = 'none'
        else:
            output_layer.append(activation_functions[self.last_non_linearity]())
            self._last_act = self.last_non_linearity
        self.output_layer = nn.Sequential(*output_layer)
        self.output_path = output_path
        self.apply(self.init_weight_method)
        self._init_hip_state()

    def init_weight_method(self, layer):
        if isinstance(layer, nn.Linear):
            init_func = init_functions[self.type_init]
            init_func(layer.weight.data,
                      gain=nn.init.calculate_gain(self.activation_layer))
            layer.bias.data.fill_(0.0)
            self._weights_rewritten()

    # -- HIP plumbing ------------------------------------------------------
    def _segments(self):
        # a row-wise softmax cannot ride in a GEMM epilogue: the tower ends linear
        # and abn_softmax_rows follows
        last = 'none' if self._last_act == 'softmax' else self._last_act
        return [_Segment(self, _blocks_of(self.input_emb, self.hidden_layers, self.output_layer),
                         0, self.activation_layer, last, self.batch_norm)]

    def _run_tower(self, x1, x2, n_calls, split):
        seg = self._segment_list()[0]
        rows = x1.shape[0] * (2 if x2 is not None else 1)
        masks = self._draw_dropout_masks(rows, x1.device) if self.training else None
        out = self._run(seg, _GradPass(self), masks, x1, x2, n_calls, split)
        if self._last_act == 'softmax':
            out = tuple(_SoftmaxRows.apply(o) for o in out) if split else _SoftmaxRows.apply(out)
        return out

    # -- autograd-free training path (TrainerSiamese.train_step) -------------------
    def direct_ok(self):
        return self._last_act != 'softmax'

    def direct_forward(self, x1, x2, forward_only=False, n_valid=None, source=None):
        """forward(x1, x2) in the current mode without building an autograd graph:
        ([2B, out] embeddings of both towers, state for direct_backward).  Dispatching
        one backward through torch's autograd engine costs ~150 us of host time per
        step (worker-thread hand-off, graph bookkeeping) -- more than the GPU needs
        for a reference-sized batch -- and the trainer knows the graph anyway."""
        seg = self._segment_list()[0]
        rows = 2 * x1.shape[0]
        masks = self._draw_dropout_masks(rows, x1.device) if self.training else None
        out, sv = _segment_forward(seg, masks, 2, x1, x2, forward_only=forward_only, n_valid=n_valid, source=source)
        return out, (seg, sv, _GradPass(self))

    def takes_padded_batch_norm(self, x12, npad):
        """Does a TRAINING forward of this (BatchNorm) network on the padded batch x12 = [tower 1: npad rows | tower 2: npad rows]
        run on the launches that take a real-row count (abn_tower_desc.n_valid: the BatchNorm layer launches)?"""
        seg = self._segment_list()[0]
        desc = seg.descriptor(with_grads=False)
        lib = _lib.load()
        return lib.abn_tower_path(_lib.C.byref(desc), _lib.ptr(x12[:npad]), _lib.ptr(x12[npad:]), 2 * npad, 2, 1, _lib.ptr(x12), 0,
                                  None) in (_lib.PATH_BN_LAYERS, _lib.PATH_BN_TOWER)

    def direct_dz_info(self, state):
        """What the pair loss needs to hand back d loss / d z of the output layer itself
        (abn_pair_loss_dz): (activation name, (mask tower 1, mask tower 2) | None), or None when
        the tower's backward must start from d loss / d e (BatchNorm in front of the activation)."""
        seg, sv, _ = state
        if seg.batch_norm or os.environ.get('ABN_LOSS_DZ') == '0':      # (the variable: A/B measurements)
            return None
        masks = None
        if isinstance(sv.masks, _DropSeed):
            return None                          # the output layer's mask exists inside the tower kernels only
        if sv.masks is not None:
            m = sv.masks[-1]
            half = m.shape[0] // 2
            masks = (m[:half], m[half:])
        return seg.last_act, masks

    def can_defer_reduce(self, state):
        """True when direct_backward may leave the split-K reduction to the optimizer's
        launch (abn_tower_reduce_step): one segment owns every live parameter."""
        seg = state[0]
        return (os.environ.get('ABN_FUSED_STEP') != '0'      # (the variable: A/B runs)
                and len(seg.params) == len(self.live_parameters()))

    def direct_backward(self, state, d_out, d_out_is_dz=False, defer_reduce=False):
        """Backward of direct_forward: gradients of every parameter land in a fresh
        flat buffer and are installed as p.grad (views), exactly what autograd's
        backward leaves behind.  With defer_reduce the p.grad views are installed but
        hold nothing yet: FlatOptimizer.step() must follow, it sums the split-K slabs,
        fills the gradients and updates the parameters in one launch."""
        seg, sv, grad_pass = state
        self._pending_reduce = None
        if defer_reduce:
            grads, _, self._pending_reduce = _segment_backward(seg, sv, d_out, grad_pass, False, d_out_is_dz, True)
        else:
            grads, _ = _segment_backward(seg, sv, d_out, grad_pass, False, d_out_is_dz)
        for p, g in zip(seg.params, grads):
            p.grad = g

    def direct_backward_loss(self, state, y, loss_kind, margin, avg, defer_reduce=False, n_valid=None, loss_accum=None, loss_ws=None,
                             wgrad_split=None):
        """loss(emb1, emb2, y) and its backward in the backward's own launches (abn_tower_backward_loss:
        the data-gradient chain computes the pair loss and d loss / d z of the output layer in its first
        phase; a BatchNorm tower's backward in the launch that sums the output layer's dy and dy xhat).  Returns the 0-dim loss,
        or None when the library does not take this tower that way (exact-fp32 arithmetic, odd widths, cross-replica
        BatchNorm statistics): the caller then uses value_and_dz + direct_backward.
        n_valid (device int32 tensor): a padded batch, only the first n_valid pairs are real; loss_accum (device
        float64 tensor): the loss is also added to it (include/abnet3_hip.h)."""
        seg, sv, grad_pass = state
        rows = sv.rows
        if (sv.n_calls != 2 or os.environ.get('ABN_LOSS_IN_BACKWARD') == '0'      # (the variable: A/B runs)
                or self._fused_loss_refused == (rows, self.precision)):
            return None
        from .loss import _scratch
        lib = _lib.load()
        y = y.contiguous()
        _lib.require_device(y)
        source = getattr(sv, 'source', None)      # (the labels are the plan's then: y = all of them, the step's offset is the library's to find)
        if y.dtype not in _lib.Y_DTYPE or (y.numel() * 2 != rows and source is None):
            return None
        grad_buf, grads = grad_pass.views(seg)
        desc = seg.descriptor(with_grads=True, grad_buf=grad_buf, masks=sv.masks, d_out_is_dz=True, defer_reduce=defer_reduce)
        if source is not None:
            desc.source = _lib.C.addressof(source)
        if wgrad_split is not None:              # data-parallel overlap: this call stops after the upper layers' gradients
            desc.wgrad_part, desc.wgrad_split = 1, int(wgrad_split)
        elif defer_reduce:
            _lend_forward_workspace(desc, sv, rows)
        if sv.n_valid is not None:               # (a padded batch through BatchNorm: the forward's real-row count)
            desc.n_valid = sv.n_valid.data_ptr()
            if n_valid is None:
                n_valid = sv.n_valid
        scratch_floats = lib.abn_tower_bwd_scratch_floats(_lib.C.byref(desc), rows)
        scratch = torch.empty(max(scratch_floats, 1), dtype=torch.float32, device=y.device)
        loss = torch.empty((), dtype=torch.float32, device=y.device)
        # the loss scratch (a ticket counter every call leaves zero + per-workgroup partial sums): the caller's own
        # buffer -- a captured step keeps one outside the graph instead of a memset node per replay -- or the
        # per-stream one
        need = lib.abn_tower_backward_loss_ws_bytes(rows)
        lws = loss_ws if loss_ws is not None and loss_ws.numel() >= need else _scratch(need, y.device)
        rc = lib.abn_tower_backward_loss(
            _lib.C.byref(desc), _lib.ptr(sv.x1), _lib.ptr(sv.x2), _lib.ptr(y), _lib.Y_DTYPE[y.dtype], _lib.LOSS[loss_kind],
            float(margin), int(bool(avg)), rows, _lib.ptr(sv.ws), _lib.ptr(scratch), scratch_floats, _lib.ptr(loss),
            _lib.ptr(lws), _lib.ptr(n_valid), _lib.ptr(loss_accum), _lib.stream())
        if rc == _lib.E_UNSUPPORTED:
            self._fused_loss_refused = (rows, self.precision)
            grad_pass.buf = None                 # nothing was written: the separate calls start this pass afresh
            grad_pass.used.discard(id(seg))
            return None
        _lib.check(rc, 'abn_tower_backward_loss')
        _lib.note_path(desc, sv.x1, sv.x2, rows, sv.n_calls, True, sv.ws, backward=True)
        self._pending_reduce = (desc, rows, scratch, scratch_floats, grad_buf, seg) if defer_reduce else None
        self._pending_lower = (desc, rows, scratch, scratch_floats, sv) if wgrad_split is not None else None
        for p, g in zip(seg.params, grads):
            p.grad = g
        return loss

    _fused_loss_refused = None

    def direct_backward_lower(self):
        """The second half of a direct_backward_loss(..., wgrad_split=s): the weight gradients of the layers below s
        (abn_tower_desc.wgrad_part = 2, same workspace and scratch).  Between the two calls the caller starts the
        all-reduce of the upper layers' gradients (TrainerSiamese.train_step under torch.distributed)."""
        desc, rows, scratch, scratch_floats, sv = self._pending_lower
        self._pending_lower = None
        desc.wgrad_part = 2
        _lib.check(_lib.load().abn_tower_backward(
            _lib.C.byref(desc), _lib.ptr(sv.x1), _lib.ptr(sv.x2), _lib.ptr(None), rows,
            sv.n_calls, _lib.ptr(sv.ws), _lib.ptr(scratch), scratch_floats, _lib.ptr(None), _lib.stream()), 'abn_tower_backward')

    def grad_split_offset(self, state, layer):
        """Float offset inside the flat gradient buffer where the gradients of the tower's layers >= `layer` begin
        (the buffer holds the parameters in layer order: the upper layers are its tail)."""
        seg = state[0]
        offs = seg.offsets()
        per_layer = len(seg.params) // len(seg.blocks)
        return int(offs[layer * per_layer])

    def take_pending_reduce(self):
        """The unfinished reduction a direct_backward(defer_reduce=True) left (or None); clears it."""
        pending = getattr(self, '_pending_reduce', None)
        self._pending_reduce = None
        return pending

    # -- reference surface ---------------------------------------------------
    def forward_once(self, x):
        """Simple forward pass for one instance x (abnet3/model.py:179-186)."""
        return self._run_tower(x, None, 1, False)

    def forward(self, input1, input2):
        """Forward pass through the same network (abnet3/model.py:188-196): both
        towers in one launch sequence, BatchNorm statistics per tower call."""
        return self._run_tower(input1, input2, 2, True)

    def forward_pair_rows(self, x12):
        """forward(x12[:B], x12[B:]) for a batch that already sits in one
        [2B, D] buffer (what a device-side batch builder produces): saves the
        concatenation copy inside the call."""
        if x12.shape[0] % 2:
            raise ValueError('abnet3_amd: forward_pair_rows needs an even number of rows')
        return self._run_tower(x12, None, 2, True)

    def save_network(self, epoch=''):
        torch.save(self.state_dict(), self.output_path + str(epoch) + '.pth')

    def load_network(self, network_path=None):
        self.load_state_dict(torch.load(network_path))


class SiameseMultitaskNetwork(_HipNetwork):
    """Siamese network for multi-task speaker and speech representation
    (abnet3/model.py:211-376): input_emb and hidden_layers_shared feed BOTH
    output_layer_spk and output_layer_phn.  hidden_layers_spk / hidden_layers_phn
    are built and initialised as in the reference (same state_dict keys, same RNG
    consumption) but, as in the reference's forward_once (model.py:337-345),
    never called: they get no gradient and no optimizer update.

    Three launch sequences per forward: the trunk over both towers' rows, then
    each one-layer head over the trunk's output; autograd adds the two heads'
    gradients into the trunk's."""

    def __init__(self, input_dim=None, num_hidden_layers_shared=None,
                 num_hidden_layers_spk=None,
                 num_hidden_layers_phn=None,
                 hidden_dim=None,
                 output_dim=None, p_dropout=0.1, batch_norm=False,
                 type_init='xavier_uni', activation_layer=None,
                 output_path=None):
        super(SiameseMultitaskNetwork, self).__init__()
        assert activation_layer in ('relu', 'sigmoid', 'tanh')
        assert type_init in ('xavier_uni', 'xavier_normal', 'orthogonal')
        assert type(input_dim) == int, 'input dim should be int'
        assert type(hidden_dim) == int, 'hidden dim should be int'
        assert type(num_hidden_layers_shared) == int
        assert type(num_hidden_layers_spk) == int
        assert type(num_hidden_layers_phn) == int
        assert type(output_dim) == int, 'output dim should be int'
        assert num_hidden_layers_shared + 1 <= _lib.MAX_LAYERS, 'too many layers'

        self.input_dim = input_dim
        self.num_hidden_layers_shared = num_hidden_layers_shared
        self.num_hidden_layers_spk = num_hidden_layers_spk
        self.num_hidden_layers_phn = num_hidden_layers_phn
        self.hidden_dim = hidden_dim
        self.output_dim = output_dim
        self.activation_layer = activation_layer
        self.batch_norm = batch_norm
        self.type_init = type_init
        self.p_dropout = p_dropout

        activation = activation_functions[activation_layer]

        def block(n_in, n_out):
            layers = [nn.Linear(n_in, n_out), nn.Dropout(p=p_dropout)]
            if self.batch_norm:
                layers.append(nn.BatchNorm1d(n_out))
            layers.append(activation())
            return layers

        def stack(n):
            layers = []
            for idx in range(n):
                layers += block(hidden_dim, hidden_dim)
            return layers

        # construction order of the reference: input, shared, spk, phn, then the
        # two output layers (every nn.Linear draws its default init from the RNG)
        self.input_emb = nn.Sequential(*block(input_dim, hidden_dim))
        shared = stack(self.num_hidden_layers_shared)
        spk = stack(self.num_hidden_layers_spk)
        phn = stack(self.num_hidden_layers_phn)
        self.hidden_layers_shared = nn.Sequential(*shared)
        self.hidden_layers_spk = nn.Sequential(*spk)
        self.hidden_layers_phn = nn.Sequential(*phn)
        self.output_layer_spk = nn.Sequential(*block(hidden_dim, output_dim))
        self.output_layer_phn = nn.Sequential(*block(hidden_dim, output_dim))

        self.output_path = output_path
        self.apply(self.init_weight_method)
        self._init_hip_state()

    def init_weight_method(self, layer):
        if isinstance(layer, nn.Linear):
            init_func = init_functions[self.type_init]
            init_func(layer.weight.data,
                      gain=nn.init.calculate_gain(self.activation_layer))
            layer.bias.data.fill_(0.0)
            self._weights_rewritten()

    def _segments(self):
        act, bn = self.activation_layer, self.batch_norm
        trunk = _blocks_of(self.input_emb, self.hidden_layers_shared)
        return [_Segment(self, trunk, 0, act, act, bn),
                _Segment(self, _blocks_of(self.output_layer_spk), len(trunk), act, act, bn),
                _Segment(self, _blocks_of(self.output_layer_phn), len(trunk) + 1, act, act, bn)]

    def _forward(self, x1, x2):
        trunk, head_spk, head_phn = self._segment_list()
        n_calls = 1 if x2 is None else 2
        rows = x1.shape[0] * n_calls
        gp = _GradPass(self)
        masks = self._draw_dropout_masks(rows, x1.device) if self.training else None
        h = self._run(trunk, gp, masks, x1, x2, n_calls, False)
        split = x2 is not None
        return (self._run(head_spk, gp, masks, h, None, n_calls, split),
                self._run(head_phn, gp, masks, h, None, n_calls, split))

    def forward_once(self, x):
        """(output_spk, output_phn) for one instance x (abnet3/model.py:337-345)."""
        return self._forward(x, None)

    def forward(self, input1, input2):
        """(spk1, phn1, spk2, phn2) (abnet3/model.py:347-356): both towers in one
        launch sequence per segment, BatchNorm statistics per tower call."""
        (spk1, spk2), (phn1, phn2) = self._forward(input1, input2)
        return spk1, phn1, spk2, phn2

    def save_network(self, epoch=''):
        torch.save(self.state_dict(), self.output_path + epoch + '.pth')

