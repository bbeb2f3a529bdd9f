"""ctypes binding of libabnet3_hip.so (include/abnet3_hip.h).

There is no CPU fallback: if the shared object is missing or a tensor is not
on a HIP device the call fails loudly.  torch is imported first so that the
library binds to the HIP runtime torch already loaded (one runtime instance =
shared streams and allocations).
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL: see module docstring)

HERE = os.path.dirname(os.path.abspath(__file__))
# ABNET3_HIP_LIB points at another build of the same library (kernel experiments)
LIB_PATH = os.environ.get('ABNET3_HIP_LIB') or os.path.join(HERE, 'lib', 'libabnet3_hip.so')
ABI_VERSION = 19
MAX_LAYERS = 16

ACT = {'none': 0, None: 0, 'sigmoid': 1, 'relu': 2, 'tanh': 3}
LOSS = {'coscos2': 0, 'cosmargin': 1}
PRECISION = {'fp32': 0, 'bf16': 1, 'bf16x3': 2, 'f16x2': 3}
OPT = {'sgd': 0, 'adadelta': 1, 'adam': 2, 'adagrad': 3, 'RMSprop': 4}
Y_DTYPE = {torch.int8: 0, torch.int32: 1, torch.int64: 2, torch.float32: 3,
           torch.float64: 4}

# every symbol include/abnet3_hip.h declares: (restype, argtypes)
_i64, _i32, _f32, _vp = C.c_int64, C.c_int32, C.c_float, C.c_void_p
SYMBOLS = {
    'abn_abi_version': (C.c_int, []),
    'abn_last_error': (C.c_char_p, []),
    'abn_tower_ws_floats': (_i64, [_vp, _i64, _i64]),
    'abn_tower_out_offset': (_i64, [_vp, _i64, _i64]),
    'abn_tower_bwd_scratch_floats': (_i64, [_vp, _i64]),
    'abn_tower_forward': (C.c_int, [_vp, _vp, _vp, _i64, _i64, C.c_int, _vp, _vp]),
    'abn_tower_backward': (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp,
                                      _i64, _vp, _vp]),
    'abn_tower_wpack_floats': (_i64, [_vp]),
    'abn_tower_uses_planes': (C.c_int, [_vp, _i64, _vp, _vp, _vp, C.c_int]),
    'abn_tower_path': (C.c_int, [_vp, _vp, _vp, _i64, _i64, C.c_int, _vp, C.c_int, _vp]),
    'abn_tower_image_offset': (_i64, [_vp, _i64, _i64, C.c_int, C.c_int]),
    'abn_reload_switches': (None, []),
    'abn_rccl_allreduce_f64': (C.c_int, [_vp, _vp, _i64, _vp]),
    'abn_tower_backward_launch': (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _i64, C.c_int, _vp]),
    'abn_tower_backward_loss_ws_bytes': (_i64, [_i64]),
    'abn_tower_backward_loss': (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, _f32, C.c_int, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    'abn_tower_reduce_step': (C.c_int, [_vp, _i64, _vp, _i64, C.c_int, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32,
                                         _f32, _i64, _f32, _vp]),
    'abn_linear_forward': (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, C.c_int, _vp, _vp]),
    'abn_linear_dgrad': (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, C.c_int, _vp, _vp]),
    'abn_linear_wgrad_scratch_floats': (_i64, [_i64, _i64, _i64]),
    'abn_linear_wgrad': (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _vp]),
    'abn_linear_backward': (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, C.c_int, _vp, _vp, _vp, _vp, _i64, _vp]),
    'abn_linear_backward_prec': (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _i64, _vp]),
    'abn_softmax_rows': (C.c_int, [_vp, _i64, _i64, _vp, _vp]),
    'abn_softmax_rows_backward': (C.c_int, [_vp, _vp, _i64, _i64, _vp, _vp]),
    'abn_pair_loss_ws_bytes': (_i64, [_i64]),
    'abn_pair_loss': (C.c_int, [_vp, _vp, _vp, C.c_int, _i64, _i64, C.c_int,
                                 _f32, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    'abn_pair_loss_padded': (C.c_int, [_vp, _vp, _vp, C.c_int, _i64, _i64, C.c_int, _f32, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'abn_pair_loss_dz': (C.c_int, [_vp, _vp, _vp, C.c_int, _i64, _i64, C.c_int, _f32, C.c_int, C.c_int,
                                    _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'abn_optimizer_step': (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _i64, _f32,
                                      _f32, _f32, _f32, _i64, _f32, _vp]),
    'abn_tower_sync_ws_bytes': (_i64, []),
    'abn_oneshot_mail_bytes': (_i64, [_i32, _i64]),
    'abn_allreduce_oneshot': (C.c_int, [_vp, _vp, _i64, _vp]),
    'abn_dtw_ws_bytes': (_i64, [_vp, _vp, _i64, _i64, _i64]),
    'abn_dtw_host_stage_bytes': (_i64, [_vp, _vp, _i64]),
    'abn_dtw_batched': (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64,
                                   _i64, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp,
                                   _i64, _vp]),
    'abn_dtw_batched_overlap': (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64,
                                           _i64, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp,
                                           _i64, _vp, _vp]),
    'abn_cosine_distance': (C.c_int, [_vp, _i64, _vp, _i64, _i64, _vp, _vp, _vp]),
    'abn_cosine_distance_f64': (C.c_int, [_vp, _i64, _vp, _i64, _i64, _vp, _vp, _vp]),
    'abn_arccos_f32': (C.c_int, [_vp, _i64, C.c_int, _vp, _vp]),
    'abn_gather_rows': (C.c_int, [_vp, _vp, _i64, _i64, _vp, _vp]),
    'abn_gather_pairs': (C.c_int, [_vp, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _vp, _i32, _vp, _vp, _vp, _vp]),
    'abn_stack_frames': (C.c_int, [_vp, _i64, _i64, _i32, _vp, _vp]),
    'abn_stack_frames_batched': (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i32, _vp, _vp]),
    'abn_mvn_ws_bytes': (_i64, [_i64, _i64]),
    'abn_mvn_stats': (C.c_int, [_vp, _i64, _i64, C.c_int, _vp, _vp, _vp, _vp]),
    'abn_mvn_apply': (C.c_int, [_vp, _i64, _i64, _vp, _vp, C.c_int, _f32, _vp, _vp]),
    'abn_fbank': (C.c_int, [_vp, C.c_int, _i64, _i32, C.c_double, _i32, _i32,
                             _f32, _vp, _vp, _vp, _i64, _vp, _vp]),
    'abn_fbank_batched': (C.c_int, [_vp, C.c_int, _vp, _vp, _i64, _i32, C.c_double, _i32, _i32,
                                     _f32, _vp, _vp, _vp, _i64, _vp, _vp]),
    'abn_deltas': (C.c_int, [_vp, _i64, _i64, _vp, _vp]),
}


class TowerDesc(C.Structure):
    """struct abn_tower_desc"""
    _fields_ = [('n_layers', _i32), ('act', _i32), ('last_act', _i32),
                ('batch_norm', _i32), ('dims', _i64 * (MAX_LAYERS + 1))] + [
        (name, _vp * MAX_LAYERS)
        for name in ('W', 'b', 'bn_w', 'bn_b', 'bn_rm', 'bn_rv', 'dW', 'db',
                     'dbn_w', 'dbn_b', 'drop_mask')] + [('precision', _i32), ('d_out_is_dz', _i32), ('defer_reduce', _i32), ('wpack_valid', _i32), ('forward_only', _i32), ('wgrad_part', _i32), ('wpack', _vp), ('drop_seed', _vp), ('drop_p', _f32), ('reserved2_', _i32),
                     ('bn_sync_world', _i32), ('wgrad_split', _i32), ('bn_sync_fn', _vp), ('bn_sync_ctx', _vp), ('n_valid', _vp),
                     ('bn_nbt', _vp * MAX_LAYERS), ('sync_ws', _vp), ('fwd_ws', _vp), ('fwd_calls', _i64), ('source', _vp)]


class StepSource(C.Structure):
    """abn_step_source (include/abnet3_hip.h): a pass's batches as the plan holds them, for steps that need no gather launch."""
    _fields_ = [('table', _vp), ('table_rows', _i64), ('idx1', _vp), ('idx2', _vp), ('labels', _vp), ('steps', _vp), ('step_ctr', _vp)]


class OneShotCtx(C.Structure):
    """struct abn_oneshot_ctx (abn_allreduce_oneshot's context)"""
    _fields_ = [('rank', _i32), ('world', _i32), ('mail', _vp * 8), ('cap_floats', _i64)]


class RcclCtx(C.Structure):
    """struct abn_rccl_ctx (abn_rccl_allreduce_f64's context)"""
    _fields_ = [('comm', _vp), ('all_reduce', _vp), ('calls', _i64)]


# abn_allreduce_fn (abn_tower_desc.bn_sync_fn)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


class HipLibraryError(RuntimeError):
    pass


_lib = None


def load():
    """Returns the bound library; raises HipLibraryError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            'abnet3_amd: %s is missing. Build it with `python -m abnet3_amd.build` '
            '(hipcc, gfx950). There is no CPU fallback.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise HipLibraryError('abnet3_amd: %s does not export %s '
                                  '(stale build?)' % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    if lib.abn_abi_version() != ABI_VERSION:
        raise HipLibraryError('abnet3_amd: ABI version mismatch (library %d, '
                              'binding %d)' % (lib.abn_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def reload_switches():
    """The library reads its A/B switches (ABN_PLANES, ABN_FUSED_MIN_ROWS, ...) from the environment once,
    when it is loaded; tests and A/B tools that change one inside a process call this afterwards."""
    load().abn_reload_switches()


E_WORKSPACE = -3
E_UNSUPPORTED = -4

# abn_tower_path's answers (include/abnet3_hip.h)
PATH_PER_LAYER, PATH_FUSED_F32, PATH_PLANES, PATH_PLANES_INFER, PATH_PLANES_INFER_BN, PATH_BN_LAYERS, PATH_WIDE, PATH_BN_TOWER = range(8)
PRECISION_NAMES = {0: 'fp32', 1: 'bf16', 2: 'bf16x3', 3: 'f16x2'}

# Which kernels the tower calls of this process took: filled in by model.py from abn_tower_path (a pure
# query with the call's own arguments) while `trace_paths` is on -- tests, bench.py and the trainer's log
# read it.  Python-side bookkeeping: the library itself keeps no record of past calls.
trace_paths = False
last_path = {'forward': -1, 'backward': -1, 'forward_precision': -1, 'backward_precision': -1}


def note_path(desc, x1, x2, rows, n_calls, train, ws, backward):
    if not trace_paths:
        return
    prec = C.c_int32(-1)
    path = load().abn_tower_path(C.byref(desc), ptr(x1), ptr(x2), rows, n_calls, int(train), ptr(ws), int(backward),
                                 C.byref(prec))
    key = 'backward' if backward else 'forward'
    last_path[key] = path
    last_path[key + '_precision'] = prec.value


def last_forward_path():
    return last_path['forward']


def last_backward_path():
    return last_path['backward']


_ON_ERROR = []        # callables run when a library call fails (loss.py: scratch whose "left zero" ticket may be dirty now)


def check(rc, what):
    if rc != 0:
        for hook in _ON_ERROR:
            hook()
        msg = load().abn_last_error()
        raise HipLibraryError('%s failed (%d): %s' % (
            what, rc, msg.decode('utf-8', 'replace') if msg else ''))


def require_device(*tensors):
    """The accelerated path only takes contiguous fp32 HIP tensors."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise HipLibraryError(
                'abnet3_amd: expected a tensor on the MI355X (HIP) device, got '
                'device=%s. The accelerated path has no CPU fallback; move the '
                'module and its inputs to the GPU (.cuda()).' % t.device)
        if not t.is_contiguous():
            raise HipLibraryError('abnet3_amd: tensor must be contiguous')


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
