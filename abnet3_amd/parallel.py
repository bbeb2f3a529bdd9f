"""Data-parallel plumbing: one process per GPU, RCCL over xGMI.

The reference has no multi-process path at all.  Pair
minibatches are independent, so data-parallel training needs exactly one
exchange per step: a SUM all-reduce of the flat fp32 gradient bucket between
backward and optimizer step (abnet3/trainer.py:239 -> :240).  The bucket is ONE
contiguous buffer (SiameseNetwork.flat_grad()), i.e. one small-message
collective (2.29 MB for C2) instead of eight.

Producer side ("each rank runs its own DTW mining"): the loaders
shard themselves (`shards_itself`).  Every decision that involves a random draw
-- which word-pair batches an epoch visits and in which order, the shuffle of
the frame-pair list -- is taken on rank 0 and broadcast, so all ranks cut ONE
order into disjoint pieces whatever their local RNG state is; each rank then
aligns (DTW) and gathers only its own batches.  A frame-pair dataset that is
aligned once (FramesDataLoader) splits the alignment work over the ranks and
exchanges the resulting index lists (all_gather_varlen).

torch.distributed's "nccl" backend IS RCCL on ROCm; "gloo" is used by the CPU
tests of this file's logic (and by the two-ranks-on-one-GPU parity test).
"""
import os
import random

import numpy as np
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def active():
    """Is this a data-parallel run -- do the collectives below execute?  Yes with more than one rank; and with a
    process group of ONE rank when ABN_DP_SINGLE_RANK=1: every exchange then runs on the real backend (RCCL on the
    single GPU of a test box) and is the identity, so the run must equal a run without a group bit for bit
    (tests/test_gpu_dp.py::test_one_rank_on_rccl)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get('ABN_DP_SINGLE_RANK') == '1'


def init_from_env(backend=None):
    """Initialises torch.distributed from RANK / WORLD_SIZE / MASTER_* if the
    process was launched by torch.distributed.run; returns (rank, world, local)."""
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if (ws > 1 or os.environ.get('ABN_DP_SINGLE_RANK') == '1') and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
        elif torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend=backend, rank=rank, world_size=ws)
    return rank, ws, local


def _comm_device():
    """Where small control messages live: RCCL moves device memory only."""
    if dist.get_backend() == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def seed_all(seed):
    """Same `random` / numpy global RNG state on every rank (the loaders draw from
    both; the reference never seeds them, abnet3/trainer.py:36-87 only stores `seed`)."""
    random.seed(seed)
    np.random.seed(seed)


def all_reduce_gradients(flat_grad, loss_is_mean, oneshot=None):
    """SUM-reduces the flat gradient bucket in place over all ranks (oneshot: a OneShotAllReduce to do it with).

    Returns the factor the optimizer must apply to the reduced gradient so that
    R ranks x B pairs equal one process with R*B pairs: 1 for a summed loss
    (avg=False, the canonical configuration), 1/R for a mean loss (avg=True)."""
    _, ws = world()
    if not active():
        return 1.0
    if oneshot is not None:
        oneshot.all_reduce(flat_grad)
    else:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return 1.0 / ws if loss_is_mean else 1.0


def shard_ids(ids, rank, ws, equal):
    """Rank r's share ids[r], ids[r+ws], ... of ONE order known to all ranks.
    equal=True keeps complete groups of `ws` only, so that every rank takes the
    same number of steps (the gradient all-reduce is a collective); equal=False
    keeps the tail (evaluation: sums and counts are all-reduced afterwards)."""
    if ws == 1:
        return ids
    n = len(ids)
    if equal:
        n = n // ws * ws
    return ids[rank:n:ws]


def shard_batches(iterator, rank, ws, drop_tail=True):
    """Round-robin shard of a batch iterator for loaders that do not shard
    themselves: rank r takes batches r, r+ws, ...  With drop_tail only complete
    groups of `ws` batches are used (train: same number of steps on every rank);
    without it rank r also gets its batch of the incomplete last group (dev)."""
    if ws == 1:
        for b in iterator:
            yield b
        return
    group = []
    for b in iterator:
        group.append(b)
        if len(group) == ws:
            yield group[rank]
            group = []
    if not drop_tail and rank < len(group):
        yield group[rank]


def broadcast_parameters(flat_params, src=0):
    if active():
        dist.broadcast(flat_params, src=src)


def broadcast_array(arr, src=0, group=None):
    """int64 numpy array, same length on every rank -> rank `src`'s values (src: a rank of `group`)."""
    arr = np.ascontiguousarray(arr, dtype=np.int64)
    if not active() or arr.size == 0:
        return arr
    t = torch.from_numpy(arr.copy()).to(_comm_device())
    dist.broadcast(t, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
    return t.cpu().numpy()


def all_reduce_min(arr, group=None):
    """int64 numpy array, same length on every rank -> elementwise minimum over the ranks."""
    arr = np.ascontiguousarray(arr, dtype=np.int64)
    if not active() or arr.size == 0:
        return arr
    t = torch.from_numpy(arr.copy()).to(_comm_device())
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return t.cpu().numpy()


def all_agree(flag, group=None):
    """True iff `flag` is true on EVERY rank: how the ranks settle a choice each of them can only judge for itself (a batch
    small enough to leave the kernels that carry the cross-replica BatchNorm sums; a library that loads) before any of them
    enters -- or skips -- a collective.  A collective itself: every rank calls it at the same point."""
    return bool(all_reduce_min(np.array([1 if flag else 0]), group=group)[0])


def all_gather_varlen(t):
    """1-d tensors of rank-dependent length -> list of every rank's tensor (on t's
    device), in rank order."""
    rank, ws = world()
    if not active():
        return [t]
    dev = _comm_device()
    n = torch.tensor([t.numel()], dtype=torch.int64, device=dev)
    lens = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(lens, n)
    lens = [int(v.item()) for v in lens]
    cap = max(max(lens), 1)
    mine = torch.zeros(cap, dtype=t.dtype, device=dev)
    mine[:t.numel()] = t.to(dev)
    parts = [torch.empty_like(mine) for _ in range(ws)]
    dist.all_gather(parts, mine)
    return [p[:n_].to(t.device) for p, n_ in zip(parts, lens)]


def _mapped_library(stem):
    """Path of the shared object whose name starts with `stem` among the images THIS process has mapped (the HIP runtime and
    RCCL torch loaded -- wherever they came from), or None."""
    try:
        with open('/proc/self/maps') as fh:
            for line in fh:
                path = line.rsplit(' ', 1)[-1].strip()
                if os.path.basename(path).startswith(stem):
                    return path
    except OSError:
        pass
    return None


class BatchNormSync:
    """Cross-replica BatchNorm statistics (SURVEY.md 8e's exact mode; abn_tower_desc.bn_sync_*): with
    `SiameseNetwork.bn_sync = BatchNormSync()` a TRAINING forward / backward of a BatchNorm tower all-reduces
    each layer's per-call [sum z, sum z^2] (forward) and [sum dy, sum dy xhat] (backward), float64, over the
    replicas, and normalises with world x rows rows: R replicas on B pairs each then step exactly like one
    process on R B pairs (tests/test_gpu_dp.py).  The library calls back between two launches, once per layer
    and direction, with a device pointer into a buffer this object was shown (`buffers`): the collective runs
    on torch's current stream, the stream the launches are on.  Default (bn_sync None): per-replica statistics,
    as torch's DistributedDataParallel without SyncBatchNorm."""

    def __init__(self, group=None):
        from . import _lib
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.collective = active()     # (False: no process group -- the sums take the same path, nobody to add them up with)
        self.buffers = []              # the float32 tensors (forward workspace, backward scratch) of the call in flight
        self.calls = 0
        self._cb = _lib.ALLREDUCE_FN(self._allreduce)      # (kept alive with the object)
        self.fn = _lib.C.cast(self._cb, _lib.C.c_void_p).value

    def _allreduce(self, ctx, ptr, n, stream):
        try:
            for t in self.buffers:
                base = t.data_ptr()
                if base <= ptr and ptr + 8 * n <= base + t.numel() * t.element_size() and (ptr - base) % 8 == 0:
                    off = (ptr - base) // 4
                    v = t.view(-1)[off:off + 2 * n].view(torch.float64)
                    if self.collective:
                        dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group)
                    self.calls += 1
                    return 0
            return 1
        except Exception:              # (an exception must not unwind through the C frames)
            import traceback
            traceback.print_exc()
            return 2


class RcclBatchNormSync:
    """BatchNormSync without Python between the launches: the library's own abn_rccl_allreduce_f64 is the descriptor's
    bn_sync_fn, its context a communicator this object creates on the RCCL torch has loaded (one ncclCommInitRank per
    trainer; the unique id travels over the existing process group).  The per-layer all-reduces are then issued from C on
    the launch stream -- no host round trip through the interpreter, no exception path through C frames.  Needs the
    "nccl" backend (device memory); parallel.bn_sync() picks it there and the Python callback elsewhere (gloo tests)."""

    _UID_BYTES = 128

    def __init__(self, group=None):
        import ctypes as C
        from . import _lib
        if not (dist.is_available() and dist.is_initialized()) or dist.get_backend(group) != 'nccl':
            raise RuntimeError('RcclBatchNormSync needs an initialised "nccl" process group')
        self.group = group
        self.world = dist.get_world_size(group)
        self.collective = True
        self.buffers = []              # (kept for the interface of BatchNormSync: nothing to validate here)
        self._comm = None
        # RCCL as this process has it: the symbols of the image torch mapped (whatever file it came from: a torch built
        # against a system RCCL has no torch/lib/librccl.so), else that file; ABN_RCCL_LIB names another (the tests: a
        # path that does not exist).  Every step below that one rank can fail alone is followed by an agreement: a rank
        # that raised while the others went on into ncclCommInitRank would leave them waiting for it.
        self._rccl, err = None, None
        try:
            self._rccl = self._load_rccl(C)
        except (OSError, AttributeError) as e:
            err = e
        if not all_agree(self._rccl is not None, group):
            raise RuntimeError('RCCL could not be loaded on every rank (%s)' % (err,))
        uid = (C.c_char * self._UID_BYTES)()
        rank = dist.get_rank(group)
        status = 0
        if rank == 0:
            status = int(self._rccl.ncclGetUniqueId(C.byref(uid)))
        words = np.concatenate([[status], np.frombuffer(bytes(uid), dtype=np.int64)]).astype(np.int64)
        if self.world > 1:
            words = broadcast_array(words, src=0, group=group)
        if int(words[0]) != 0:
            raise RuntimeError('ncclGetUniqueId failed on rank 0 (%d)' % int(words[0]))
        C.memmove(uid, words[1:].tobytes(), self._UID_BYTES)
        comm = C.c_void_p()

        class _Uid(C.Structure):
            _fields_ = [('internal', C.c_char * self._UID_BYTES)]
        u = _Uid()
        C.memmove(C.byref(u), uid, self._UID_BYTES)
        self._rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _Uid, C.c_int]
        torch.cuda.synchronize()
        rc = self._rccl.ncclCommInitRank(C.byref(comm), self.world, u, rank)
        good = rc == 0 and bool(comm.value)
        if good:
            self._comm = comm
        if not all_agree(good, group):
            self._destroy()
            raise RuntimeError('ncclCommInitRank failed on some rank (here: %d)' % rc)
        self._ctx = _lib.RcclCtx(comm.value, C.cast(self._rccl.ncclAllReduce, C.c_void_p).value, 0)
        self.ctx = C.addressof(self._ctx)
        self.fn = C.cast(_lib.load().abn_rccl_allreduce_f64, C.c_void_p).value

    @staticmethod
    def _load_rccl(C):
        forced = os.environ.get('ABN_RCCL_LIB')
        candidates = [forced] if forced else [_mapped_library('librccl'), os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so'),
                                              'librccl.so']
        candidates = [c for c in candidates if c]
        last = None
        for path in candidates:
            try:
                lib = C.CDLL(path)
                for name in ('ncclGetUniqueId', 'ncclCommInitRank', 'ncclAllReduce', 'ncclCommDestroy'):
                    getattr(lib, name)
                return lib
            except (OSError, AttributeError) as e:
                last = e
        raise OSError('no RCCL with ncclAllReduce / ncclCommInitRank found (%s)' % (last,))

    def _destroy(self):
        if getattr(self, '_comm', None) is not None and self._comm.value:
            self._rccl.ncclCommDestroy.argtypes = [type(self._comm)]
            self._rccl.ncclCommDestroy(self._comm)
        self._comm = None

    @property
    def calls(self):
        return int(self._ctx.calls)

    def __del__(self):
        try:
            import sys
            if sys.is_finalizing():        # (at interpreter exit the process group / RCCL may be gone already: the OS reclaims the communicator)
                return
            self._destroy()
        except Exception:
            pass


class OneShotAllReduce:
    """The gradient bucket's SUM all-reduce as ONE launch per rank over peer-mapped mailboxes (abn_allreduce_oneshot:
    reduce-scatter + all-gather in two hops over all xGMI links at once, summed in rank order -- SURVEY.md section 5 / 8e)
    instead of torch.distributed's ring.  Each rank allocates its mailbox as fine-grained device memory, exports it with
    hipIpcGetMemHandle, gathers everybody's handle over the existing process group and maps the peers' (hipIpcOpenMemHandle).
    Behind a switch (TrainerBuilder / ABN_ONESHOT_ALLREDUCE=1): it has run between two processes on ONE GPU only (the
    tests), never over xGMI.  `cap_floats`: the largest bucket it will be asked to reduce."""

    _HANDLE_BYTES = 64
    _FINEGRAINED = 0x1                 # hipDeviceMallocFinegrained
    _IPC_LAZY = 0x1                    # hipIpcMemLazyEnablePeerAccess

    def __init__(self, cap_floats, group=None):
        import ctypes as C
        from . import _lib
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError('OneShotAllReduce needs an initialised process group')
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if self.world > 8:
            raise RuntimeError('OneShotAllReduce: at most 8 ranks (one node)')
        self.cap = (int(cap_floats) + 3) // 4 * 4
        self._lib = _lib.load()
        self._hip = C.CDLL(_mapped_library('libamdhip64') or 'libamdhip64.so')       # the HIP runtime torch has loaded
        for name in ('hipExtMallocWithFlags', 'hipIpcGetMemHandle', 'hipIpcOpenMemHandle', 'hipIpcCloseMemHandle', 'hipFree', 'hipMemset'):
            getattr(self._hip, name)
        nbytes = self._lib.abn_oneshot_mail_bytes(self.world, self.cap)
        if nbytes <= 0:
            raise RuntimeError('abn_oneshot_mail_bytes refused (%d ranks, %d floats)' % (self.world, self.cap))
        self._bytes = nbytes
        mine = C.c_void_p()
        rc = self._hip.hipExtMallocWithFlags(C.byref(mine), C.c_size_t(nbytes), C.c_uint(self._FINEGRAINED))
        ok = rc == 0 and bool(mine.value)
        handle = (C.c_char * self._HANDLE_BYTES)()
        if ok:
            ok = self._hip.hipMemset(mine, 0, C.c_size_t(nbytes)) == 0 and self._hip.hipIpcGetMemHandle(C.byref(handle), mine) == 0
        self._mine = mine if mine.value else None
        self._opened = []
        if not all_agree(ok, group):
            self.close()
            raise RuntimeError('OneShotAllReduce: the mailbox could not be allocated / exported on every rank (here: %d)' % rc)
        torch.cuda.synchronize()
        # everybody's handle (and pid: a rank maps only OTHER processes' memory; its own it has already)
        words = np.concatenate([[os.getpid()], np.frombuffer(bytes(handle), dtype=np.int64)]).astype(np.int64)
        dev = _comm_device()
        t = torch.from_numpy(words.copy()).to(dev)
        parts = [torch.empty_like(t) for _ in range(self.world)]
        if self.world > 1:
            dist.all_gather(parts, t, group=group)
        else:
            parts = [t]
        self._ctx = _lib.OneShotCtx()
        self._ctx.rank, self._ctx.world, self._ctx.cap_floats = self.rank, self.world, self.cap
        good = True

        class _Handle(C.Structure):
            _fields_ = [('reserved', C.c_char * self._HANDLE_BYTES)]
        self._hip.hipIpcOpenMemHandle.argtypes = [C.POINTER(C.c_void_p), _Handle, C.c_uint]
        for r in range(self.world):
            if r == self.rank:
                self._ctx.mail[r] = mine.value
                continue
            w = parts[r].cpu().numpy()
            h = _Handle()
            C.memmove(C.byref(h), w[1:].tobytes(), self._HANDLE_BYTES)
            peer = C.c_void_p()
            rc = self._hip.hipIpcOpenMemHandle(C.byref(peer), h, C.c_uint(self._IPC_LAZY))
            if rc != 0 or not peer.value:
                good = False
                self.open_error = rc
                break
            self._opened.append(peer)
            self._ctx.mail[r] = peer.value
        if not all_agree(good, group):
            self.close()
            raise RuntimeError('OneShotAllReduce: hipIpcOpenMemHandle failed on some rank (here: %s)' % getattr(self, 'open_error', 0))
        self.calls = 0

    def all_reduce(self, flat):
        """SUM over the ranks, in place, on torch's current stream; flat: contiguous fp32 on the device, a multiple of 4
        elements (the networks' flat gradient buffer is padded to 64)."""
        from . import _lib
        if flat.dtype != torch.float32 or not flat.is_contiguous() or flat.numel() % 4 or flat.numel() > self.cap:
            raise ValueError('OneShotAllReduce.all_reduce: contiguous fp32, a multiple of 4 elements, at most %d' % self.cap)
        _lib.check(self._lib.abn_allreduce_oneshot(_lib.C.byref(self._ctx), _lib.ptr(flat), flat.numel(), _lib.stream()),
                   'abn_allreduce_oneshot')
        self.calls += 1

    def failed(self):
        """Has a call of this rank given up on a peer (word 16 of the mailbox)?  One 4-byte read that synchronises with the
        device: look where the loss is read back anyway.  The trainer does, and raises on every rank (the gradients of that
        step were NaN on all of them: abn_allreduce_oneshot hands NaN to the peers of a rank that gives up)."""
        import ctypes as C
        word = C.c_uint32(0)
        torch.cuda.synchronize()
        self._hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        rc = self._hip.hipMemcpy(C.byref(word), C.c_void_p(self._mine.value + 64), C.c_size_t(4), C.c_int(2))      # device to host
        return rc != 0 or word.value != 0

    def close(self):
        try:
            for p in getattr(self, '_opened', []):
                self._hip.hipIpcCloseMemHandle(p)
            self._opened = []
            if getattr(self, '_mine', None) is not None:
                self._hip.hipFree(self._mine)
                self._mine = None
        except Exception:
            pass

    def __del__(self):
        import sys
        if not sys.is_finalizing():
            self.close()


def bn_sync(group=None):
    """The cross-replica BatchNorm exchange for this process group: issued from C over RCCL on the "nccl" backend
    (ABN_BN_SYNC_PY=1: the Python callback there too), the Python callback over torch.distributed otherwise."""
    if (dist.is_available() and dist.is_initialized() and dist.get_backend(group) == 'nccl'
            and os.environ.get('ABN_BN_SYNC_PY') != '1'):
        try:
            return RcclBatchNormSync(group)
        except (RuntimeError, OSError) as e:     # (raised on EVERY rank together: RcclBatchNormSync agrees before it raises)
            import warnings
            warnings.warn('abnet3_amd: the cross-replica BatchNorm exchange falls back to the Python callback '
                          '(parallel.BatchNormSync): %s' % (e,))
    return BatchNormSync(group)
