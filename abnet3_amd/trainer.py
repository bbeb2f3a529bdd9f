"""Trainers on MI355X: the class surface of abnet3/trainer.py.

Mirrors (file:line relative to the reference checkout)
  TrainerBuilder   abnet3/trainer.py:32-201  (same kwargs, defaults, early
                   stopping, best-model saving, whoami)
  TrainerSiamese   abnet3/trainer.py:203-256 (give_batch_to_network,
                   optimize_model)
  TrainerSiameseMultitask  abnet3/trainer.py:259-279
Differences, all on purpose:
  * the optimizer is one fused HIP kernel over the network's flat parameter
    buffer with torch.optim's update rules (FlatOptimizer);
  * per-batch losses are accumulated on the device in fp64 and read back once
    per epoch (the reference syncs every batch at trainer.py:242,248);
  * under torch.distributed (one process per GPU) the flat gradient bucket is
    all-reduced over RCCL between backward and step, and batches are sharded
    round-robin over ranks;
  * tensorboardX is optional (absent in this image): scalars then only go to
    train_losses / dev_losses.
"""
import copy
import gc
import os
import pickle
import time
import warnings
from pathlib import Path

import torch
import torch.optim as optim

import weakref

import numpy as np

from . import _lib, parallel
from .loss import unit_grad
from .model import NetworkBuilder, SiameseMultitaskNetwork

try:                                        # pragma: no cover
    from tensorboardX import SummaryWriter
except Exception:                           # noqa: the package is optional here
    class SummaryWriter(object):
        def __init__(self, log_dir=None):
            self.log_dir = log_dir

        def add_scalar(self, *args, **kwargs):
            pass



# Trainers that hold planned-pass state (scratch with a "left zero" ticket, captured graphs holding its address).  A library
# error may leave the ticket dirty: the state of every live trainer is dropped -- the scratch, the graphs, their pool and
# the pass's running loss, so that nothing of an aborted pass can be mistaken for a result -- and the error propagates out of
# the pass (nothing in this module catches HipLibraryError); the next pass allocates and captures afresh.
_PLANNED_TRAINERS = weakref.WeakSet()


def _drop_planned_state():
    for t in list(_PLANNED_TRAINERS):
        for name in ('_loss_ws', '_buckets', '_bucket_pool', '_loss_acc'):
            t.__dict__.pop(name, None)


_lib._ON_ERROR.append(_drop_planned_state)

class FlatOptimizer(object):
    """torch.optim.{SGD,Adadelta,Adam,Adagrad,RMSprop} semantics (torch's default
    hyper-parameters, abnet3/trainer.py:68-87) as ONE kernel launch over the
    network's flat parameter / gradient buffers (abn_optimizer_step)."""

    HP = {  # kind: (hp0, hp1, eps)
        'sgd': (None, 0.0, 0.0),
        'adadelta': (0.9, 0.0, 1e-6),
        'adam': (0.9, 0.999, 1e-8),
        'adagrad': (0.0, 0.0, 1e-10),
        'RMSprop': (0.99, 0.0, 1e-8),
    }

    def __init__(self, network, kind, lr, momentum=0.9):
        assert kind in self.HP
        self.network = network
        self.kind = kind
        self.lr = float(lr)
        self.momentum = float(momentum if momentum is not None else 0.0)
        self.step_count = 0
        self.grad_scale = 1.0
        self._flat = None
        self._s1 = self._s2 = None

    def zero_grad(self, set_to_none=True):
        for p in self.network.parameters():
            p.grad = None

    def _state(self):
        if not self.network._is_flat(full=True):
            self.network.flatten_parameters()
        flat = self.network.flat_parameters()
        if self._flat is None or self._flat.data_ptr() != flat.data_ptr():
            if self.step_count > 0:
                raise RuntimeError('abnet3_amd: the network was moved after the '
                                   'optimizer took its first step')
            self._flat = flat
            self._s1 = torch.zeros_like(flat)
            self._s2 = torch.zeros_like(flat)
        return flat

    def step(self):
        lib = _lib.load()
        flat = self._state()
        take = getattr(self.network, 'take_pending_reduce', None)
        pending = take() if take is not None else None
        grad = self.network.flat_grad()
        _lib.require_device(flat, grad)
        self.step_count += 1
        hp0, hp1, eps = self.HP[self.kind]
        if self.kind == 'sgd':
            hp0 = self.momentum
        if pending is not None:
            # the backward left its split-K slabs unreduced: reduction + update in one launch
            desc, rows, scratch, scratch_floats, grad_buf, seg = pending
            if grad_buf.data_ptr() != grad.data_ptr():
                raise RuntimeError('abnet3_amd: gradients were replaced between a deferred backward and step()')
            _lib.check(lib.abn_tower_reduce_step(
                _lib.C.byref(desc), rows, _lib.ptr(scratch), scratch_floats, _lib.OPT[self.kind], _lib.ptr(flat),
                _lib.ptr(grad), _lib.ptr(self._s1), _lib.ptr(self._s2), flat.numel(), self.lr, hp0, hp1, eps,
                self.step_count, float(self.grad_scale), _lib.stream()), 'abn_tower_reduce_step')
            self.network.weights_changed_behind_torch()
            return
        if hasattr(self.network, 'weights_changed_behind_torch'):
            self.network.weights_changed_behind_torch()
        _lib.check(lib.abn_optimizer_step(
            _lib.OPT[self.kind], _lib.ptr(flat), _lib.ptr(grad), _lib.ptr(self._s1),
            _lib.ptr(self._s2), flat.numel(), self.lr, hp0, hp1, eps, self.step_count,
            float(self.grad_scale), _lib.stream()), 'abn_optimizer_step')

    def state_dict(self):
        return {'kind': self.kind, 'lr': self.lr, 'momentum': self.momentum,
                'step': self.step_count, 's1': self._s1, 's2': self._s2}


def _capture_mode():
    """hipGraph capture mode: with a process group alive its watchdog thread polls the events of finished collectives
    (hipEventQuery) whenever it likes -- under the default 'global' mode such a call from ANOTHER thread invalidates a capture
    in progress (the one-rank RCCL run died in ProcessGroupNCCL's watchdog); 'thread_local' confines the check to the
    capturing thread's own calls."""
    import torch.distributed as dist
    return 'thread_local' if dist.is_available() and dist.is_initialized() else 'global'


class TrainerBuilder:
    """Generic Trainer class for ABnet3 (abnet3/trainer.py:32-201)."""

    def __init__(self, network=None, loss=None,
                 num_epochs=200, patience=20,
                 optimizer_type='sgd', lr=0.001, momentum=0.9, cuda=True,
                 seed=0, dataloader=None, log_dir=None,
                 feature_generator=None,
                 checkpoints=False, sync_batch_norm=False):
        self.network = network
        self.loss = loss
        self.num_epochs = num_epochs
        self.patience = patience
        self.lr = lr
        self.momentum = momentum
        self.best_epoch = 0
        self.seed = seed
        self.cuda = cuda
        self.statistics_training = {}
        self.dataloader = dataloader
        self.feature_generator = feature_generator
        self.checkpoints = checkpoints
        self.rank, self.world_size = parallel.world()
        self.dp = parallel.active()         # the collectives of a data-parallel step execute (parallel.active)

        if not cuda:
            warnings.warn('abnet3_amd has no CPU path: cuda=False is ignored and '
                          'the trainer runs on the MI355X')
        self.cuda = True
        self.loss.cuda()
        self.network.cuda()

        if log_dir is None:
            self.log_dir = Path('./runs/%s' % time.strftime('%m-%d-%Hh%M-%S'))
        else:
            self.log_dir = Path(log_dir) / ('%s' % time.strftime('%m-%d-%Hh%M-%S'))

        assert optimizer_type in ('sgd', 'adadelta', 'adam', 'adagrad',
                                  'RMSprop', 'LBFGS')
        if optimizer_type == 'LBFGS':
            # like the reference: constructed, but .step() needs a closure the
            # reference loop never passes (trainer.py:240)
            self.optimizer = optim.LBFGS(self.network.parameters(), lr=self.lr)
        else:
            self.optimizer = FlatOptimizer(self.network, optimizer_type, self.lr,
                                           self.momentum)
        if self.dp:
            parallel.broadcast_parameters(self.network.flat_parameters())
            if hasattr(self.network, 'weights_changed_behind_torch'):
                self.network.weights_changed_behind_torch()      # (a rank that already ran a forward holds a stale image)
            # one RNG state on all ranks to start from; what an epoch visits is
            # broadcast from rank 0 anyway (parallel.py, the loaders' batch_iterator)
            parallel.seed_all(self.seed)
            # the gradient bucket's exchange as ONE launch per rank over peer-mapped mailboxes (parallel.OneShotAllReduce) instead
            # of torch.distributed's ring: behind a switch -- it has never run over xGMI (tests: two processes on one GPU)
            if os.environ.get('ABN_ONESHOT_ALLREDUCE') == '1':
                try:
                    self.oneshot = parallel.OneShotAllReduce(self.network.flat_parameters().numel())
                except (RuntimeError, OSError, AttributeError) as e:      # (every rank together: OneShotAllReduce agrees before it raises)
                    warnings.warn('abnet3_amd: ABN_ONESHOT_ALLREDUCE: falling back to torch.distributed.all_reduce (%s)' % (e,))
            # BatchNorm statistics over ALL replicas' rows (off: every replica normalises with its own, like torch's
            # DistributedDataParallel without SyncBatchNorm): R replicas then step like one process on the whole batch
            if sync_batch_norm and getattr(self.network, 'batch_norm', False):
                self.network.bn_sync = parallel.bn_sync()
        self.sync_batch_norm = bool(sync_batch_norm)

    def params(self):
        """None.  The reference's params() copies the trainer's attributes into a dictionary and then
        drops it (no return statement, abnet3/trainer.py:89-92): every .params file it ever wrote holds
        'params': None, and so do ours."""
        return None

    def whoami(self):
        """What save_whoami() pickles: the class name and the whoami() of every part
        (abnet3/trainer.py:94-104)."""
        parts = {'network': self.network, 'loss': self.loss, 'dataloader': self.dataloader}
        if self.feature_generator is not None:
            parts['feature_generator'] = self.feature_generator
        description = {name: part.whoami() for name, part in parts.items()}
        description['params'] = self.params()
        description['class_name'] = type(self).__name__
        return description

    def save_whoami(self):
        """<output_path>.params beside the network, as the reference leaves it
        (abnet3/trainer.py:106-108); the networks' whoami() leaves their HIP
        plumbing out, everything else is the reference's dictionary."""
        with open(self.network.output_path + '.params', 'wb') as fh:
            pickle.dump(self.whoami(), fh)

    def optimize_model(self, do_training=True):
        raise NotImplementedError('Unimplemented optimize_model for class:',
                                  self.__class__.__name__)

    def _keep_best(self, epoch):
        """Rank 0 writes the new best network (and a numbered checkpoint when asked), then its description."""
        if self.rank != 0:
            return
        print('Saving best model so far, epoch {}... '.format(epoch + 1), end='', flush=True)
        if self.checkpoints:
            self.network.save_network(epoch=epoch + 1)
        self.network.save_network()
        self.save_whoami()
        print('Done.')

    def train(self):
        """The reference's training schedule (abnet3/trainer.py:117-173), seen from outside:
        * the untrained network is saved and evaluated once (pass 0: optimize_model(False), in eval mode);
        * then up to num_epochs passes of optimize_model(True); after each, the train and dev losses go to
          two tensorboard curves (<log_dir>/train_loss, <log_dir>/dev_loss, step = epoch, pass 0 at step 0);
        * a pass whose dev loss beats the best so far saves the network (+ .params, + a numbered checkpoint
          with `checkpoints`) and resets the patience counter; any other pass counts against `patience`, and
          the (patience + 1)-th such pass in a row ends the run.
        Leaves best_dev / best_epoch / patience_dev / train_losses / dev_losses behind, as the reference does.
        Under torch.distributed every rank runs the same schedule (the losses are all-reduced in
        optimize_model); only rank 0 writes files and prints."""
        master = self.rank == 0
        self.patience_dev, self.best_dev = 0, None
        self.train_losses, self.dev_losses = [], []
        self.num_batches_train = self.num_batches_dev = 0
        curves = {'train': SummaryWriter(log_dir=str(self.log_dir / 'train_loss')),
                  'dev': SummaryWriter(log_dir=str(self.log_dir / 'dev_loss'))}

        def plot(step):
            curves['train'].add_scalar('loss', self.train_losses[-1], step)
            curves['dev'].add_scalar('loss', self.dev_losses[-1], step)

        # pass 0
        self.network.eval()
        if master:
            self.network.save_network()
        self.optimize_model(do_training=False)
        plot(0)
        if self.checkpoints and master:
            self.network.save_network(epoch=0)
        self.statistics_training = dict.fromkeys(self.statistics_training, 0)

        for epoch in range(self.num_epochs):
            dev_loss = self.optimize_model(do_training=True)
            plot(epoch + 1)
            if self.best_dev is None or dev_loss < self.best_dev:
                self.best_dev, self.best_epoch, self.patience_dev = dev_loss, epoch, 0
                self._keep_best(epoch)
                continue
            self.patience_dev += 1
            if self.patience_dev > self.patience:
                if master:
                    print('No improvements after {} iterations, stopping now'.format(self.patience))
                    print('Finished Training')
                break
        if master:
            print('Saving best checkpoint network')

    def plot_summary_statistics(self):
        print(" ***** Statistics for the training step ***** ")
        for key in self.statistics_training.keys():
            stats = self.statistics_training[key]
            print(" Number of {} pairs seen: {} \t\t".format(key, stats))

    def pretty_print_losses(self, train_loss, dev_loss):
        if self.rank == 0:
            print("  training loss:\t\t{:.6f}".format(train_loss))
            print("  dev loss:\t\t\t{:.6f}".format(dev_loss))


class TrainerSiamese(TrainerBuilder):
    """Siamese Trainer class for ABnet3 (abnet3/trainer.py:203-256)."""

    def __init__(self, *args, **kwargs):
        super(TrainerSiamese, self).__init__(*args, **kwargs)
        assert isinstance(self.network, NetworkBuilder)

    def give_batch_to_network(self, batch):
        """Feeds one (X1, X2, y) batch to the network and returns the loss to
        optimize (abnet3/trainer.py:211-224)."""
        X_batch1, X_batch2, y_batch = batch
        X_batch1 = X_batch1.cuda(non_blocking=True)
        X_batch2 = X_batch2.cuda(non_blocking=True)
        y_batch = y_batch.cuda(non_blocking=True)
        emb_batch1, emb_batch2 = self.network(X_batch1, X_batch2)
        return self.loss(emb_batch1, emb_batch2, y_batch)

    def _direct_ok(self):
        """The autograd-free step applies to the plain Siamese case: our network and
        pair losses, the fused optimizer, the unmodified give_batch_to_network."""
        if not getattr(self, 'direct_steps', True):
            return False
        ok = getattr(self, '_direct_cache', None)
        if ok is None:
            from .loss import coscos2, cosmargin
            from .model import SiameseNetwork
            ok = self._direct_cache = (
                type(self.network) is SiameseNetwork and self.network.direct_ok()
                and type(self.loss) in (coscos2, cosmargin)
                and isinstance(self.optimizer, FlatOptimizer)
                and type(self).give_batch_to_network is TrainerSiamese.give_batch_to_network)
        return ok

    @staticmethod
    def _backward(loss_value):
        """loss.backward() (abnet3/trainer.py:239) seeded with the cached unit
        gradient instead of a fresh ones_like: two small launches less per step."""
        if loss_value.dim() == 0 and loss_value.dtype == torch.float32 and loss_value.is_cuda:
            torch.autograd.backward(loss_value, grad_tensors=unit_grad(loss_value.device))
        else:
            loss_value.backward()

    overlap_allreduce = True          # data-parallel steps: two gradient buckets, the first all-reduce under the rest of the backward
    oneshot = None                    # parallel.OneShotAllReduce: the bucket's exchange as one launch over peer-mapped mailboxes (ABN_ONESHOT_ALLREDUCE=1)

    def _overlap_split(self, state):
        """The layer the data-parallel backward is cut at (None: one call, one all-reduce): towers of >= 3 layers whose
        parameters all live in ONE segment of the flat buffer, no BatchNorm; the upper bucket = the top half of the layers."""
        net = self.network
        seg = state[0]
        if not self.overlap_allreduce or self.oneshot is not None or getattr(net, 'batch_norm', False) or len(seg.blocks) < 3:
            return None
        one = getattr(self, '_one_segment', None)           # (the parameter walk once, not per step)
        if one is None or one[0] is not seg:
            one = self._one_segment = (seg, len(seg.params) == 2 * len(seg.blocks) and len(list(net.parameters())) == len(seg.params))
        return (len(seg.blocks) + 1) // 2 if one[1] else None

    def _loss_is_mean(self):
        """Data-parallel gradient exchange: a mean loss averages over ranks, a
        summed loss sums."""
        return bool(getattr(self.loss, 'avg', False))

    def train_step(self, batch, do_training=True):
        """The five statements of the reference's inner loop
        (abnet3/trainer.py:236-240) plus the data-parallel gradient exchange.
        Returns the (device) loss of the batch; never synchronises."""
        if do_training and batch[0].shape[0] > 0 and self._direct_ok():
            # the five statements with the trainer driving the kernels itself: forward,
            # fused loss + its gradient, backward -- no autograd graph, no engine
            X_batch1, X_batch2, y_batch = batch
            X_batch1 = X_batch1.cuda(non_blocking=True)
            X_batch2 = X_batch2.cuda(non_blocking=True)
            y_batch = y_batch.cuda(non_blocking=True)
            emb, state = self.network.direct_forward(X_batch1, X_batch2)
            n = X_batch1.shape[0]
            info = self.network.direct_dz_info(state)
            self.optimizer.zero_grad()
            # single process: nothing happens between backward and step, so the split-K
            # reduction of the weight gradients rides in the optimizer's launch
            defer = not self.dp and self.network.can_defer_reduce(state)
            # data-parallel: the backward in two calls, the all-reduce of the upper layers' gradients in flight while the
            # lower layers' are computed (two buckets of the flat gradient buffer; abn_tower_desc.wgrad_part)
            split = self._overlap_split(state) if self.dp else None
            # the pair loss inside the backward's first launch, where the library offers it
            loss_value = self.network.direct_backward_loss(
                state, y_batch, type(self.loss).__name__, getattr(self.loss, 'margin', 0.0), self.loss.avg, defer_reduce=defer,
                wgrad_split=split)
            if loss_value is not None and split is not None:
                flat = self.network.flat_grad()
                cut = self.network.grad_split_offset(state, split)
                work = [torch.distributed.all_reduce(flat[cut:], op=torch.distributed.ReduceOp.SUM, async_op=True)]
                self.network.direct_backward_lower()
                work.append(torch.distributed.all_reduce(flat[:cut], op=torch.distributed.ReduceOp.SUM, async_op=True))
                for w in work:
                    w.wait()
                self.optimizer.grad_scale = 1.0 / self.world_size if self._loss_is_mean() else 1.0
                self.optimizer.step()
                return loss_value.detach()
            if loss_value is not None:
                pass
            elif info is not None:    # loss gradient and the output layer's act' (+ dropout) in ONE launch
                loss_value, dz = self.loss.value_and_dz(emb[:n], emb[n:], y_batch, info[0], info[1])
                self.network.direct_backward(state, dz.view(2 * n, -1), d_out_is_dz=True, defer_reduce=defer)
            else:
                loss_value, de = self.loss.value_and_grad(emb[:n], emb[n:], y_batch)
                self.network.direct_backward(state, de.view(2 * n, -1), defer_reduce=defer)
            if self.dp:
                if split is not None:
                    # the library did not take the loss into the backward on THIS rank (row count, alignment): the
                    # other ranks may have -- the same two collectives in the same order, whatever path ran here
                    flat = self.network.flat_grad()
                    cut = self.network.grad_split_offset(state, split)
                    torch.distributed.all_reduce(flat[cut:], op=torch.distributed.ReduceOp.SUM)
                    torch.distributed.all_reduce(flat[:cut], op=torch.distributed.ReduceOp.SUM)
                    self.optimizer.grad_scale = 1.0 / self.world_size if self._loss_is_mean() else 1.0
                else:
                    self.optimizer.grad_scale = parallel.all_reduce_gradients(
                        self.network.flat_grad(), self._loss_is_mean(), self.oneshot)
            self.optimizer.step()
        elif do_training:
            loss_value = self.give_batch_to_network(batch)
            self.optimizer.zero_grad()
            self._backward(loss_value)
            if self.dp:
                self.optimizer.grad_scale = parallel.all_reduce_gradients(
                    self.network.flat_grad(), self._loss_is_mean(), self.oneshot)
            self.optimizer.step()
        else:
            with torch.no_grad():
                loss_value = self.give_batch_to_network(batch)
            self.optimizer.zero_grad()
        return loss_value.detach()

    def make_graphed_step(self, example_batch, warmup=3):
        """Captures one whole train step (forward of both towers, loss,
        backward, optimizer) for batches shaped like `example_batch` into a
        hipGraph and returns step(batch) -> device loss.  A C2 step is ~15
        kernel launches of 5-50 us each: replaying one graph removes the
        per-launch host cost (Python, autograd, ctypes) from the loop.  Under
        torch.distributed the gradient all-reduce and the optimizer stay
        outside the graph (fwd + bwd are captured).  Adam's bias correction is
        host-computed per step, so its optimizer launch also stays outside."""
        bs = getattr(self.network, 'bn_sync', None)
        if bs is not None and type(bs).__name__ != 'RcclBatchNormSync':
            # (on RCCL the exchange is a stream-ordered ncclAllReduce issued by the library itself -- parallel.RcclBatchNormSync --
            # and is captured like any launch; the Python callback runs torch.distributed between two launches)
            raise NotImplementedError('abnet3_amd: a step with cross-replica BatchNorm statistics through the Python callback '
                                      '(parallel.BatchNormSync) calls back to the host between launches and cannot be captured '
                                      'into a graph')
        static, fwd_loss = self._graph_inputs(example_batch)
        opt = self.optimizer
        capture_opt = (not self.dp and isinstance(opt, FlatOptimizer)
                       and opt.kind != 'adam')
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):      # also moves the optimizer past step 1
                warmup_loss = self.train_step(tuple(static), True)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        steps_before = getattr(opt, 'step_count', 0)
        # No cyclic garbage collection while the stream is capturing: a collector run
        # started by the allocations below may finalise an OLD trainer's CUDAGraph (its
        # step closure and the trainer reference each other), whose destructor frees
        # device memory -- not permitted during capture, and fatal inside a destructor.
        # torch.cuda.graph() collects once before it begins capturing.
        gc_was_enabled = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.graph(graph, capture_error_mode=_capture_mode()):
                loss_value = fwd_loss()
                opt.zero_grad()
                self._backward(loss_value)
                if capture_opt:
                    opt.step()
        finally:
            if gc_was_enabled:
                gc.enable()
        if capture_opt:
            opt.step_count = steps_before        # capturing recorded the launch, it did not run it
        static_loss = loss_value.detach()
        # The replayed backward writes into the buffers THIS capture allocated.  An
        # eager step in between (another batch shape, train_step_auto) re-points
        # p.grad / _last_grad_flat at its own buffer; every replay therefore puts the
        # captured gradient tensors back before anything reads flat_grad().
        net = self.network
        live = list(net.live_parameters()) if hasattr(net, 'live_parameters') else list(net.parameters())
        captured_grads = [p.grad for p in live]
        captured_flat = getattr(net, '_last_grad_flat', None)

        blob = getattr(self, '_static_blob', None)
        self._static_blob = None

        def step(batch):
            if isinstance(batch, torch.Tensor):          # a pack_batch() blob
                blob.copy_(batch, non_blocking=True)
            else:
                for dst, src in zip(static, batch):
                    dst.copy_(src, non_blocking=True)
            graph.replay()
            if hasattr(net, 'weights_changed_behind_torch'):
                net.weights_changed_behind_torch()
            for p, g in zip(live, captured_grads):
                p.grad = g
            if captured_flat is not None:
                net._last_grad_flat = captured_flat
            if capture_opt:
                opt.step_count += 1
            else:
                if self.dp:
                    opt.grad_scale = parallel.all_reduce_gradients(
                        self.network.flat_grad(), self._loss_is_mean(), self.oneshot)
                opt.step()
            return static_loss
        step.graph = graph
        step.warmup_loss = warmup_loss
        step.flat_ptr = self.network.flat_parameters().data_ptr()
        return step

    # shapes a captured step exists for / how often a shape was seen
    GRAPH_AFTER = 3          # capture once a batch shape has come up this often
    MAX_GRAPHS = 4

    def train_step_auto(self, batch):
        """train_step for an epoch loop.  With self.graph_steps = True, batches whose
        shape keeps recurring (the FramesDataLoader's fixed `batch_size` frame pairs)
        are served by a captured hipGraph from their GRAPH_AFTER-th occurrence on;
        every other batch takes the eager path.  Off by default: since train_step
        drives the kernels without autograd the eager step is within 20 % of the
        replayed one even at 100 frame pairs (and ahead of it from ~500-wide towers
        on, where the three input copies of a replay cost more than its launches)."""
        if (not getattr(self, 'graph_steps', False) or not isinstance(self.optimizer, FlatOptimizer)
                or getattr(self.network, 'bn_sync', None) is not None):
            return self.train_step(batch, True)
        if not hasattr(self, '_graphs'):
            self._graphs, self._shape_seen = {}, {}
        key = tuple((tuple(t.shape), str(t.dtype)) for t in batch)
        step = self._graphs.get(key)
        if step is not None:
            if step.flat_ptr == self.network.flat_parameters().data_ptr() and self.network.training:
                return step(batch)
            del self._graphs[key]                # parameters were re-homed: the graph is stale
        n = self._shape_seen[key] = self._shape_seen.get(key, 0) + 1
        if n < self.GRAPH_AFTER or len(self._graphs) >= self.MAX_GRAPHS or batch[0].shape[0] == 0:
            return self.train_step(batch, True)
        step = self.make_graphed_step(batch, warmup=1)   # its one warm-up step IS this batch's step
        self._graphs[key] = step
        return step.warmup_loss

    # -- planned passes: loaders that hand out a BatchPlan (dataloader.py) ----------------------------------
    # The reference's canonical loader yields a different number of frame pairs every step (8 word pairs =
    # a few hundred 280-d rows, abnet3/dataloader.py:248-255): the step is tens of microseconds of GPU work
    # behind ~150 us of per-batch host work.  With a plan, a step is ONE gather launch (abn_gather_pairs,
    # straight into the static buffers of a captured step, zero rows up to the bucket's size and the number
    # of real pairs in a device word) and ONE graph replay; a bucket = the batch sizes that round up to the
    # same multiple of BUCKET pairs.  Padded rows contribute no loss and no gradient (abn_tower_backward_loss,
    # n_valid).  planned_passes = False switches back to the plain iterator.
    BUCKET = 32              # pairs: a tower call is padded to whole 32-row workgroups anyway
    MIN_BUCKET = 32          # (the layer-per-launch kernels of csrc/tower_wide.h take any number of 32-row workgroups)
    MAX_BUCKET_GRAPHS = 96

    def _planned(self, train_mode):
        """The loader's plan of one pass, or None when this trainer / network / loader takes the iterator."""
        dl = self.dataloader
        if not getattr(self, 'planned_passes', True) or not hasattr(dl, 'plan'):
            return None
        if not self._direct_ok():
            return None
        if getattr(self.network, 'batch_norm', False) and getattr(self.network, 'bn_sync', None) is not None:
            return None                      # (cross-replica statistics: the replicas' real-row counts differ)
        if getattr(self, '_plan_refused', False):
            return None
        return dl.plan(train_mode)

    def _bucket(self, n):
        return max(self.MIN_BUCKET, (n + self.BUCKET - 1) // self.BUCKET * self.BUCKET)

    def _bucket_state(self, npad, plan):
        st = getattr(self, '_buckets', None)
        flat_ptr = self.network.flat_parameters().data_ptr()
        if st is not None and self._bucket_flat_ptr != flat_ptr:
            st = None                        # the parameters were re-homed (.cuda() / .to()): every captured step is stale
        if st is None:
            self._bucket_flat_ptr = flat_ptr
            st = self._buckets = {}
            self._bucket_pool = torch.cuda.graph_pool_handle()
            self._loss_acc = torch.zeros((), dtype=torch.float64, device=plan.table.device)
            self._loss_ws = torch.zeros(1 << 16, dtype=torch.uint8, device=plan.table.device)     # (ticket + partial sums of the in-backward loss)
            # a failed or aborted launch may leave the ticket counter non-zero, and every later planned step or replay
            # would then elect the wrong "last" workgroup: on a library error the scratch and the graphs that captured
            # its address are dropped (the next pass allocates and captures afresh)
            _PLANNED_TRAINERS.add(self)      # (ONE module-level hook over the live trainers: a gridsearch builds many)
        key = (npad, plan.table.shape[1], plan.labels.dtype)
        b = st.get(key)
        if b is None:
            dev, D = plan.table.device, plan.table.shape[1]
            b = st[key] = dict(x12=torch.zeros(2 * npad, D, dtype=torch.float32, device=dev),
                               y=torch.zeros(npad, dtype=plan.labels.dtype, device=dev),
                               nv=torch.zeros(1, dtype=torch.int32, device=dev), graph=None, npad=npad)
        return b

    def _bucket_body(self, b):
        """The five statements of the reference's loop (abnet3/trainer.py:236-240) on a bucket's static buffers -- or, for a
        bucket that reads its batches from the pass's plan (b['source']: abn_step_source), on the plan itself."""
        net, opt, npad = self.network, self.optimizer, b['npad']
        x12 = b['x12']
        src = b.get('source')
        # (BatchNorm: the batch statistics span the real rows only -- abn_tower_desc.n_valid, the device word the gather wrote)
        emb, state = net.direct_forward(x12[:npad], x12[npad:], n_valid=b['nv'] if getattr(net, 'batch_norm', False) else None,
                                        source=src['src'] if src is not None else None)
        opt.zero_grad()
        defer = not self.dp and net.can_defer_reduce(state)
        loss_value = net.direct_backward_loss(state, src['labels'] if src is not None else b['y'], type(self.loss).__name__,
                                              getattr(self.loss, 'margin', 0.0), self.loss.avg, defer_reduce=defer, n_valid=b['nv'],
                                              loss_accum=self._loss_acc, loss_ws=self._loss_ws)
        return loss_value

    # -- steps that read their batch from the plan (abn_step_source): no gather launch, nothing to tell a replay ------------
    def _step_source(self, plan):
        """The pass's plan as the library's step source, with this pass's (first pair, pairs) table uploaded and the step
        counter at zero -- or None where a step's last launch is not the optimizer's (data-parallel, Adam) or the switch
        says no (ABN_STEP_SOURCE=0).  The device arrays persist between passes: captured steps hold their addresses."""
        opt = self.optimizer
        if (self.dp or os.environ.get('ABN_STEP_SOURCE') == '0' or not isinstance(opt, FlatOptimizer) or opt.kind == 'adam'
                or getattr(self.network, 'batch_norm', False) or not plan.order):
            return None
        key = (plan.table.data_ptr(), plan.idx1.data_ptr(), plan.idx2.data_ptr(), plan.labels.data_ptr(), plan.table.shape[0])
        st = getattr(self, '_src', None)
        if st is None or st['key'] != key or st['steps'].shape[0] < len(plan.order):
            if st is not None:
                self._forget_captured_steps()
            dev = plan.table.device
            steps = torch.zeros(max(len(plan.order), 1024), 2, dtype=torch.int64, device=dev)
            ctr = torch.zeros(1, dtype=torch.int32, device=dev)
            src = _lib.StepSource()
            src.table, src.table_rows = plan.table.data_ptr(), plan.table.shape[0]
            src.idx1, src.idx2, src.labels = plan.idx1.data_ptr(), plan.idx2.data_ptr(), plan.labels.data_ptr()
            src.steps, src.step_ctr = steps.data_ptr(), ctr.data_ptr()
            st = self._src = dict(key=key, steps=steps, ctr=ctr, src=src, labels=plan.labels, plan_arrays=(plan.table, plan.idx1, plan.idx2))
        spans = np.array([plan.span(b) for b in plan.order], dtype=np.int64).reshape(-1, 2)
        st['steps'][:len(spans)].copy_(torch.from_numpy(spans), non_blocking=False)
        st['ctr'].zero_()
        return st

    def _bucket_finish(self):
        opt = self.optimizer
        if self.dp:
            opt.grad_scale = parallel.all_reduce_gradients(self.network.flat_grad(), self._loss_is_mean(), self.oneshot)
        opt.step()

    def _planned_step(self, plan, bid, source=None):
        """One training step on batch `bid` of the plan; returns 0 (nothing stepped) when the library refuses
        the padded form for this network / batch size: the caller then takes the iterator's step on the batch; 1: stepped;
        2: stepped from the plan itself (`source`: the step's own last launch has advanced the pass's step counter)."""
        lib = _lib.load()
        first, n = plan.span(bid)
        if n == 0:        # no arrays: ValueError like the reference's np.vstack([]); zero frames: the iterator's (empty) step
            self._loss_acc.add_(self.train_step(plan.materialise(bid), True))
            return 1
        b = self._bucket_state(self._bucket(n), plan)
        if 'source' not in b:
            # a bucket reads from the plan when its size runs on the layer-per-launch kernels (asked once per bucket)
            b['source'] = None
            if source is not None:
                probe = self.network._segment_list()[0].descriptor(with_grads=False)
                x12 = b['x12']
                if lib.abn_tower_path(_lib.C.byref(probe), _lib.ptr(x12[:b['npad']]), _lib.ptr(x12[b['npad']:]), 2 * b['npad'], 2, 1,
                                      _lib.ptr(x12), 0, None) == _lib.PATH_WIDE and plan.table.shape[1] % 4 == 0:
                    b['source'] = source
        if b['source'] is not None and b['source'] is not source:
            return 0                                  # (a bucket captured against another pass's arrays: cannot happen after _forget_captured_steps)
        sourced = b['source'] is not None
        if not sourced:
            _lib.check(lib.abn_gather_pairs(_lib.ptr(plan.table), plan.table.shape[0], plan.table.shape[1], _lib.ptr(plan.idx1), _lib.ptr(plan.idx2),
                                            first, n, b['npad'], _lib.ptr(plan.labels), plan.labels.element_size(),
                                            _lib.ptr(b['x12']), _lib.ptr(b['y']), _lib.ptr(b['nv']), _lib.stream()),
                       'abn_gather_pairs')
        done = 2 if sourced else 1
        opt = self.optimizer
        in_graph_opt = not self.dp and opt.kind != 'adam'      # (Adam's bias correction is host arithmetic per step)
        if b['graph'] is not None:
            g, grads, flat, pending = b['graph']
            g.replay()
            net = self.network
            net.weights_changed_behind_torch()
            for p_, g_ in zip(net.live_parameters(), grads):
                p_.grad = g_
            net._last_grad_flat = flat
            if in_graph_opt:
                opt.step_count += 1
            else:
                net._pending_reduce = pending
                self._bucket_finish()
            return done
        # first batch of this bucket: the step itself runs eagerly (and warms everything up), then the same
        # launch sequence is captured for the batches to come (capturing executes nothing)
        if getattr(self.network, 'batch_norm', False) and not self.network.takes_padded_batch_norm(b['x12'], b['npad']):
            return 0                          # (odd widths: the per-layer kernels, no real-row count inside their statistics)
        if self._bucket_body(b) is None:
            return 0
        self._bucket_finish()
        if len([1 for v in self._buckets.values() if v['graph'] is not None]) >= self.MAX_BUCKET_GRAPHS:
            return done
        steps_before = opt.step_count
        graph = torch.cuda.CUDAGraph()
        gc_was_enabled = gc.isenabled()
        gc.disable()                          # (see make_graphed_step)
        try:
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, pool=self._bucket_pool, capture_error_mode=_capture_mode()):
                self._bucket_body(b)
                pending = getattr(self.network, '_pending_reduce', None)
                if in_graph_opt:
                    opt.step()
        finally:
            if gc_was_enabled:
                gc.enable()
        opt.step_count = steps_before
        net = self.network
        b['graph'] = (graph, [p_.grad for p_ in net.live_parameters()], getattr(net, '_last_grad_flat', None), pending)
        return done

    def _planned_eval(self, plan, bid):
        """loss(network(batch)) of batch `bid` without gradients, in the network's current mode, added to the
        accumulator: one gather launch + one replay of the bucket's captured [forward, pair loss]."""
        lib = _lib.load()
        first, n = plan.span(bid)
        if n == 0:
            with torch.no_grad():
                self._loss_acc.add_(self.give_batch_to_network(plan.materialise(bid)))
            return
        b = self._bucket_state(self._bucket(n), plan)
        _lib.check(lib.abn_gather_pairs(_lib.ptr(plan.table), plan.table.shape[0], plan.table.shape[1], _lib.ptr(plan.idx1), _lib.ptr(plan.idx2),
                                        first, n, b['npad'], _lib.ptr(plan.labels), plan.labels.element_size(),
                                        _lib.ptr(b['x12']), _lib.ptr(b['y']), _lib.ptr(b['nv']), _lib.stream()),
                   'abn_gather_pairs')
        key = 'eval_train' if self.network.training else 'eval'
        if b.get(key) is not None:
            b[key].replay()
            return

        def body():
            net, npad = self.network, b['npad']
            emb, _ = net.direct_forward(b['x12'][:npad], b['x12'][npad:], forward_only=True)
            loss_value = torch.empty((), dtype=torch.float32, device=emb.device)
            D = emb.shape[1]
            _lib.check(lib.abn_pair_loss_padded(
                _lib.ptr(emb), _lib.ptr(emb[npad:]), _lib.ptr(b['y']), _lib.Y_DTYPE[b['y'].dtype], npad, D,
                _lib.LOSS[type(self.loss).__name__], float(getattr(self.loss, 'margin', 0.0)), int(bool(self.loss.avg)),
                _lib.ptr(b['nv']), _lib.ptr(loss_value), _lib.ptr(self._loss_acc), None, None, _lib.ptr(self._loss_ws),
                _lib.stream()), 'abn_pair_loss_padded')
        with torch.no_grad():
            body()
            if sum(1 for v in self._buckets.values() for k in ('graph', 'eval', 'eval_train') if v.get(k) is not None) >= 3 * self.MAX_BUCKET_GRAPHS:
                return
            graph = torch.cuda.CUDAGraph()
            gc_was_enabled = gc.isenabled()
            gc.disable()
            try:
                torch.cuda.synchronize()
                with torch.cuda.graph(graph, pool=self._bucket_pool, capture_error_mode=_capture_mode()):
                    body()
            finally:
                if gc_was_enabled:
                    gc.enable()
        b[key] = graph

    def _run_planned_eval(self, plan, loss_sum):
        """A pass without gradients over a plan (the dev pass, or the untrained first training pass)."""
        if not plan.order:
            return 0
        self._bucket_state(self._bucket(plan.span(plan.order[0])[1]), plan)
        self._loss_acc.zero_()
        for bid in plan.order:
            self._planned_eval(plan, bid)
        loss_sum.add_(self._loss_acc)
        return len(plan.order)

    def _run_planned(self, plan, do_training, loss_sum):
        """The training pass over a plan; adds the batches' losses to `loss_sum` (device float64) and returns
        the number of batches."""
        if not do_training:
            if getattr(self.network, 'batch_norm', False) and self.network.training:
                # a pass without gradients in TRAIN mode (the untrained first pass, abnet3/trainer.py:137): batch statistics, and
                # the running ones move -- the iterator's step on each of the plan's batches (only the real rows exist there)
                if plan.order:
                    self._bucket_state(self._bucket(max(1, plan.span(plan.order[0])[1])), plan)      # (creates the accumulator)
                    self._loss_acc.zero_()
                    for bid in plan.order:
                        self._loss_acc.add_(self.train_step(plan.materialise(bid), False))
                    loss_sum.add_(self._loss_acc)
                return len(plan.order)
            return self._run_planned_eval(plan, loss_sum)
        if not plan.order:
            return 0
        source = self._step_source(plan)          # (first: it may drop the captured steps of another plan)
        self._bucket_state(self._bucket(plan.span(plan.order[0])[1]), plan)      # (creates the accumulator)
        self._loss_acc.zero_()
        for k, bid in enumerate(plan.order):
            done = 0 if getattr(self, '_plan_refused', False) else self._planned_step(plan, bid, source)
            if not done:
                # the library does not take this network / batch in the padded form (exact-fp32 arithmetic, odd widths,
                # a tiny batch with the layer-per-launch kernels switched off): the iterator's step on the same batch;
                # refused on the very first batch = refused for good (the next passes do not ask again)
                if k == 0:
                    self._plan_refused = True
                self._loss_acc.add_(self.train_step(plan.materialise(bid), True))
            if source is not None and done != 2:
                source['ctr'].add_(1)             # (a step that did not read the plan: the pass's step counter moves all the same)
        loss_sum.add_(self._loss_acc)
        return len(plan.order)

    @staticmethod
    def _packed_layout(B, D, y):
        nx = 2 * B * D * 4
        off_y = (nx + 255) // 256 * 256
        return nx, off_y, off_y + y.numel() * y.element_size()

    def pack_batch(self, batch):
        """(X1, X2, y) as ONE device byte blob laid out like the captured step's
        static inputs ([X1; X2] float32, then y, 256-byte aligned): a batch builder
        that assembles batches in this form feeds a replayed step with a single
        device copy instead of three."""
        x1, x2, y = [t.cuda() for t in batch]
        B, D = x1.shape
        nx, off_y, total = self._packed_layout(B, D, y)
        blob = torch.zeros(total, dtype=torch.uint8, device=x1.device)
        blob[:nx].view(torch.float32).view(2 * B, D).copy_(torch.cat([x1.float(), x2.float()]))
        blob[off_y:total].view(y.dtype).copy_(y.reshape(-1))
        return blob

    def _graph_inputs(self, example_batch):
        """Static device tensors a captured step reads its batch from, and the
        forward+loss closure over them.  Both towers' rows sit in ONE [2B, D]
        buffer (saves the concatenation inside the network call), and that buffer
        and the labels share one allocation (see pack_batch)."""
        B, D = example_batch[0].shape
        y0 = example_batch[2].cuda()
        nx, off_y, total = self._packed_layout(B, D, y0)
        blob = self.pack_batch(example_batch)
        x12 = blob[:nx].view(torch.float32).view(2 * B, D)
        x1, x2 = x12[:B], x12[B:]
        y = blob[off_y:total].view(y0.dtype)

        def fwd_loss():
            e1, e2 = self.network.forward_pair_rows(x12)
            return self.loss(e1, e2, y)
        self._static_blob = blob
        return [x1, x2, y], fwd_loss

    def _batches(self, train_mode):
        it = self.dataloader.batch_iterator(train_mode=train_mode)
        if self.world_size > 1 and not getattr(self.dataloader, 'shards_itself', False):
            # a foreign loader: every rank iterates everything and keeps its share.  The
            # train pass needs the same number of steps on every rank (the all-reduce is
            # a collective); the dev pass keeps its tail (sums / counts are reduced later)
            it = parallel.shard_batches(it, self.rank, self.world_size, drop_tail=train_mode)
        return it

    RESIDENT_CHECK_EVERY = 256      # eager steps between two looks at the resident BatchNorm towers' failure words

    def _forget_captured_steps(self):
        """Captured steps hold the addresses a tower's descriptor had when they were captured (its sync buffer among them)."""
        for name in ('_graphs', '_shape_seen', '_buckets', '_bucket_pool', '_loss_acc', '_loss_ws', '_src'):
            self.__dict__.pop(name, None)

    def optimize_model(self, do_training=True):
        """Optimization model step for the Siamese network
        (abnet3/trainer.py:226-256)."""
        dev_ = self.network.flat_parameters().device
        train_loss = torch.zeros((), dtype=torch.float64, device=dev_)
        dev_loss = torch.zeros((), dtype=torch.float64, device=dev_)
        num_batches_train = 0
        num_batches_dev = 0
        timed = getattr(self, 'time_passes', False)      # (measurement: one more synchronisation per pass)
        if timed:
            torch.cuda.synchronize()
            t_pass = time.perf_counter()
        self.network.train()
        plan = self._planned(True)
        # A resident BatchNorm tower launch that gives up (the GPU shared with another process: csrc/tower_bn_persist.h)
        # drops its step -- the optimizer's launch leaves the parameters alone -- and reports a loss that is not a number.
        # The eager loop therefore sums the losses in chunks of RESIDENT_CHECK_EVERY steps and looks at the towers' failure
        # words between chunks (one device-to-host read): a chunk with a dropped step is left out of the pass's mean, the
        # tower is cleared and trains on the layer launches from then on (SiameseNetwork.resident_tower_failed warns once).
        watch = do_training and hasattr(self.network, 'resident_tower_failed') and any(
            seg.sync_fail_word() is not None for seg in self.network._segment_list())
        if plan is not None:
            num_batches_train = self._run_planned(plan, do_training, train_loss)
            if watch and self.network.resident_tower_failed():
                self._forget_captured_steps()        # (they hold the sync buffer's address; this pass's training loss is NaN)
        else:
            chunk_loss, chunk_n = (torch.zeros_like(train_loss), 0) if watch else (train_loss, 0)
            for minibatch in self._batches(True):
                # fp64 accumulator += fp32 loss in ONE launch (add_ promotes the operand)
                if do_training:
                    chunk_loss.add_(self.train_step_auto(minibatch))
                else:
                    chunk_loss.add_(self.train_step(minibatch, False))
                chunk_n += 1
                if watch and chunk_n == self.RESIDENT_CHECK_EVERY:
                    if self.network.resident_tower_failed():
                        self._forget_captured_steps()
                    else:
                        train_loss.add_(chunk_loss)
                        num_batches_train += chunk_n
                    chunk_loss.zero_()
                    chunk_n = 0
            if watch:
                if self.network.resident_tower_failed():
                    self._forget_captured_steps()
                else:
                    train_loss.add_(chunk_loss)
                    num_batches_train += chunk_n
            else:
                num_batches_train = chunk_n

        if timed:
            torch.cuda.synchronize()
            t_mid = time.perf_counter()
        self.network.eval()
        with torch.no_grad():
            plan = self._planned(False)
            if plan is not None:
                num_batches_dev = self._run_planned_eval(plan, dev_loss)
            else:
                for minibatch in self._batches(False):
                    num_batches_dev += 1
                    dev_loss.add_(self.give_batch_to_network(minibatch))

        if timed:
            torch.cuda.synchronize()
            self.pass_seconds = getattr(self, 'pass_seconds', []) + [(t_mid - t_pass, time.perf_counter() - t_mid)]
        sums = torch.stack([train_loss, dev_loss])
        counts = torch.tensor([num_batches_train, num_batches_dev], dtype=torch.float64,
                              device=dev_)
        if self.dp:
            torch.distributed.all_reduce(sums)
            torch.distributed.all_reduce(counts)
        train_loss, dev_loss = [float(v) for v in sums.cpu()]      # one sync per epoch
        if self.oneshot is not None and not parallel.all_agree(not self.oneshot.failed()):
            # a rank gave up waiting for a peer inside abn_allreduce_oneshot: that step's gradients were NaN on EVERY rank
            # (the rank that gives up hands NaN to its peers), so are the parameters since: every rank stops here
            raise RuntimeError('abnet3_amd: abn_allreduce_oneshot gave up on a peer during this pass (a rank died or stalled '
                               'for more than a minute); the parameters are not numbers any more on any rank')
        num_batches_train, num_batches_dev = [max(float(v), 1.0) for v in counts.cpu()]

        self.train_losses.append(train_loss / num_batches_train)
        self.dev_losses.append(dev_loss / num_batches_dev)
        self.pretty_print_losses(train_loss / num_batches_train,
                                 dev_loss / num_batches_dev)
        return dev_loss


class TrainerSiameseMultitask(TrainerSiamese):
    """Siamese Trainer class for ABnet3 for multi task phn and spk
    (abnet3/trainer.py:259-279): batches are (X1, X2, y_spk, y_phn)."""

    def __init__(self, *args, **kwargs):
        super(TrainerSiameseMultitask, self).__init__(*args, **kwargs)
        assert type(self.network) == SiameseMultitaskNetwork

    def give_batch_to_network(self, batch):
        X_batch1, X_batch2, y_spk_batch, y_phn_batch = batch
        X_batch1 = X_batch1.cuda(non_blocking=True)
        X_batch2 = X_batch2.cuda(non_blocking=True)
        y_spk_batch = y_spk_batch.cuda(non_blocking=True)
        y_phn_batch = y_phn_batch.cuda(non_blocking=True)
        emb_spk1, emb_phn1, emb_spk2, emb_phn2 = self.network(X_batch1, X_batch2)
        return self.loss(emb_spk1, emb_phn1, emb_spk2, emb_phn2,
                         y_spk_batch, y_phn_batch)

    def _loss_is_mean(self):
        spk = bool(getattr(self.loss.loss_spk, 'avg', False))
        phn = bool(getattr(self.loss.loss_phn, 'avg', False))
        if spk != phn:
            raise NotImplementedError(
                'abnet3_amd: data-parallel multitask training needs both base losses '
                'to agree on avg (one summed and one averaged loss cannot share one '
                'all-reduce scale)')
        return spk

    def _graph_inputs(self, example_batch):
        self._static_blob = None
        static = [t.cuda().clone() for t in example_batch]
        return static, lambda: self.give_batch_to_network(tuple(static))

