"""Host helpers + HIP front ends for the batch-production side of the hot path.

Mirrors (file:line relative to the reference checkout)
  cosine_distance     abnet3/utils.py:40-60     -> abn_cosine_distance
  get_dtw_alignment   abnet3/utils.py:147-153   -> abn_dtw_batched (1 pair)
  dtw_align_batch     (new) the same for a whole batch of token pairs at once
  Features_Accessor   abnet3/utils.py:118-145   host slicing semantics
  read_dataset / group_pairs / read_pairs   abnet3/utils.py:156-208
  read_vad_file       abnet3/utils.py:238-254
  print_token         abnet3/utils.py:101-105   (pairs-file number format)
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib


PAIR_TYPES = ('same', 'diff')


def print_token(tok):
    """One token of a pairs-file line: "<file> <onset> <offset>", times with two decimals
    (the format abnet3/utils.py:101-105 prints and abnet3/sampler.py writes)."""
    name, onset, offset = tok[0], tok[1], tok[2]
    return '%s %.2f %.2f' % (name, onset, offset)


def read_spkid_file(spkid_file):
    """{file id: speaker id} from a file of "<fid> <spkid>" lines; a file id may appear once
    (abnet3/utils.py:23-31)."""
    speaker_of = {}
    with open(spkid_file) as fh:
        for line in fh:
            fid, spkid = line.strip().split(' ')
            if fid in speaker_of:
                raise AssertionError('%s: %s listed twice' % (spkid_file, fid))
            speaker_of[fid] = spkid
    return speaker_of


def _parse_pair_line(line):
    """"f1 s1 e1 f2 s2 e2 type" -> (f1, s1, e1, f2, s2, e2, type): seven space-separated fields,
    the four times as floats, the type one of PAIR_TYPES (AssertionError otherwise, as in the
    reference's reader, abnet3/utils.py:156-173)."""
    fields = line.strip().split(' ')
    if len(fields) != 7:
        raise AssertionError('a pairs line has 7 fields, got %d: %r' % (len(fields), line))
    kind = fields[6]
    if kind not in PAIR_TYPES:
        raise AssertionError('Unsupported pair type {0}'.format(kind))
    return (fields[0], float(fields[1]), float(fields[2]), fields[3], float(fields[4]), float(fields[5]), kind)


def read_dataset(dataset_file):
    """The word pairs of a pairs file, one 7-tuple per line, in file order."""
    with open(dataset_file) as fh:
        return [_parse_pair_line(line) for line in fh]


def write_dataset(dataset_file, pairs):
    """The sampler's output format (abnet3/sampler.py:697-742): one line per
    word pair, times printed %.2f."""
    with open(dataset_file, 'w') as fh:
        for f1, s1, e1, f2, s2, e2, pair_type in pairs:
            fh.write('%s %s %s\n' % (print_token((f1, s1, e1)),
                                     print_token((f2, s2, e2)), pair_type))


def group_pairs(pairs):
    """Splits 7-tuples by their type into {'same': [6-tuples], 'diff': [6-tuples]}, order kept
    (abnet3/utils.py:176-192)."""
    by_type = {kind: [] for kind in PAIR_TYPES}
    for pair in pairs:
        kind = pair[6]
        if kind not in by_type:
            raise AssertionError('Unsupported pair type {0}'.format(kind))
        by_type[kind].append(tuple(pair[:6]))
    return by_type


def read_pairs(pair_file):
    return group_pairs(read_dataset(pair_file))


def read_vad_file(path):
    """{file: [[start, end], ...]} from a voice-activity file: a header line, then "file,start,stop" lines
    (the ZeroSpeech 2015 format the reference reads, abnet3/utils.py:238-254); windows keep file order."""
    windows = {}
    with open(path) as fh:
        next(fh, None)                                   # header
        for line in fh:
            line = line.strip()
            if not line:
                continue
            name, start, stop = line.split(',')
            windows.setdefault(name, []).append([float(start), float(stop)])
    return windows


def cast_features(features, target_type=np.float32):
    """Every array of the dictionary converted in place (abnet3/utils.py:228-235)."""
    for key, array in list(features.items()):
        features[key] = array.astype(target_type)
    return features


class Features_Accessor(object):
    """Frames of a file by time window or by frame range (abnet3/utils.py:118-145).  A time window
    keeps every frame whose time stamp lies in [start, end], BOTH ends included; keys may be bytes
    (h5features) or str.  The features are float32 from here on."""

    def __init__(self, times, features):
        self.times = times
        first = next(iter(features.values()))
        self.features = features if first.dtype == np.float32 else cast_features(features)

    @staticmethod
    def get_indices_between(time, start, end):
        inside = (time >= start) & (time <= end)
        return np.flatnonzero(inside)

    @staticmethod
    def get_features_between(feature, time, start, end):
        return feature[Features_Accessor.get_indices_between(time, start, end), :]

    def _name(self, f):
        """h5features hands out bytes keys: try the encoded name first."""
        as_bytes = f.encode('UTF-8')
        return as_bytes if as_bytes in self.times else f

    def get(self, f, on, off):
        key = self._name(f)
        return self.get_features_between(self.features[key], self.times[key], on, off)

    def get_between_frames(self, f, frame_on, frame_off):
        return self.features[self._name(f)][frame_on:frame_off]


def _as_device_f32(a):
    if isinstance(a, torch.Tensor):
        t = a
    else:
        t = torch.from_numpy(np.ascontiguousarray(a))
    if t.dtype != torch.float32:
        raise AssertionError('features must be float32 (abnet3/utils.py:41-42)')
    return t.cuda().contiguous()


def cosine_distance(x, y):
    """Angular distance matrix arccos(cos)/pi, float64 [N, M] (utils.py:40-60),
    computed on the MI355X in the precision of the inputs, as the reference does
    (both float32 -- the hot path -- or both float64, utils.py:41-42).  Raises
    AssertionError, like the reference's `assert np.all(d >= 0)`, when an entry is NaN
    (cos rounded above 1)."""
    x = np.asarray(x) if not isinstance(x, torch.Tensor) else x
    y = np.asarray(y) if not isinstance(y, torch.Tensor) else y
    f64 = torch.float64 if isinstance(x, torch.Tensor) else np.float64
    f32 = torch.float32 if isinstance(x, torch.Tensor) else np.float32
    assert (x.dtype == f64 and y.dtype == f64) or (x.dtype == f32 and y.dtype == f32)
    lib = _lib.load()
    if x.dtype == f64:
        xd = (x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))).cuda().contiguous()
        yd = (y if isinstance(y, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(y))).cuda().contiguous()
        fn, what = lib.abn_cosine_distance_f64, 'abn_cosine_distance_f64'
    else:
        xd, yd = _as_device_f32(x), _as_device_f32(y)
        fn, what = lib.abn_cosine_distance, 'abn_cosine_distance'
    N, M, D = xd.shape[0], yd.shape[0], xd.shape[1]
    assert yd.shape[1] == D
    d = torch.empty(N, M, dtype=torch.float64, device=xd.device)
    bad = torch.zeros(1, dtype=torch.int32, device=xd.device)
    _lib.check(fn(_lib.ptr(xd), N, _lib.ptr(yd), M, D, _lib.ptr(d), _lib.ptr(bad), _lib.stream()), what)
    assert int(bad.item()) == 0, 'cosine_distance produced NaN / negative entries'
    return d.cpu().numpy()


class DtwBatchResult(object):
    """Device-resident result of dtw_align_batch.  Path p is RIGHT-ALIGNED in row p of
    path1 / path2: entries [stride - path_len[p], stride), in forward order (see
    include/abnet3_hip.h, abn_dtw_batched)."""

    def __init__(self, path1, path2, path_len, total_cost):
        self.path1, self.path2 = path1, path2          # int32 [P, stride] (device)
        self.path_len = path_len                       # int32 [P] (device); 0 = dropped
        self.total_cost = total_cost                   # float64 [P] (device)

    def mask(self):
        """bool [P, stride]: the entries that belong to a path."""
        stride = self.path1.shape[1]
        return torch.arange(stride, device=self.path1.device)[None, :] >= (stride - self.path_len)[:, None]

    def to_lists(self):
        ln = self.path_len.cpu().numpy()
        p1 = self.path1.cpu().numpy()
        p2 = self.path2.cpu().numpy()
        st = p1.shape[1]
        return [(p1[i, st - ln[i]:].copy(), p2[i, st - ln[i]:].copy()) if ln[i] > 0 else None
                for i in range(len(ln))]


class _DtwScratch(object):
    """Grow-only workspace + pinned staging buffer of dtw_align_batch, one per
    (device, stream): a batch of 10 000 pairs needs ~5 GB of workspace, and
    allocating it per call made one call in three stall 80 ms in hipMalloc /
    hipHostMalloc.  The workspace is reused in stream order; the pinned staging
    buffer (read by an asynchronous H2D copy) is reused only after the event
    recorded behind the previous call has completed."""
    _cache = {}

    def __init__(self):
        self.ws = None
        self.stage = None
        self.done = None
        self.side = None

    def side_stream(self, device):
        """The second stream abn_dtw_batched_overlap runs the traceback on, beside the fill kernel
        (ABN_DTW_OVERLAP=0: one stream, the traceback behind the fill)."""
        if os.environ.get('ABN_DTW_OVERLAP', '1') == '0':
            return None
        if self.side is None:
            self.side = torch.cuda.Stream(device)
        return self.side

    @classmethod
    def get(cls, device):
        key = (device.index, torch.cuda.current_stream(device).cuda_stream)
        sc = cls._cache.get(key)
        if sc is None:
            sc = cls._cache[key] = cls()
        return sc

    def buffers(self, ws_bytes, hs_bytes, device):
        if self.ws is None or self.ws.numel() < ws_bytes:
            self.ws = None                       # release before growing
            self.ws = torch.empty(int(ws_bytes * 1.25), dtype=torch.uint8, device=device)
        if self.done is not None:
            self.done.synchronize()
        if self.stage is None or self.stage.numel() < hs_bytes:
            self.stage = torch.empty(int(hs_bytes * 1.25) + 256, dtype=torch.uint8).pin_memory()
        return self.ws, self.stage

    def mark(self):
        if self.done is None:
            self.done = torch.cuda.Event()
        self.done.record()                       # (one event, recorded again: the previous record has been waited for in buffers())


def release_dtw_scratch():
    """Frees the cached DTW workspaces (they are kept between calls)."""
    _DtwScratch._cache.clear()


def dtw_align_batch(feats1, off1, n1, feats2, off2, n2):
    """Aligns pair p = rows [off1[p], off1[p]+n1[p]) of feats1 with rows
    [off2[p], off2[p]+n2[p]) of feats2 for all p in one launch sequence.

    feats1/feats2: [rows, D] float32 device tensors (the same tensor may be
    passed twice).  off*/n*: host int arrays.  Returns DtwBatchResult."""
    lib = _lib.load()
    _lib.require_device(feats1, feats2)
    off1 = np.ascontiguousarray(off1, dtype=np.int64)
    off2 = np.ascontiguousarray(off2, dtype=np.int64)
    n1 = np.ascontiguousarray(n1, dtype=np.int32)
    n2 = np.ascontiguousarray(n2, dtype=np.int32)
    P = len(n1)
    dev = feats1.device
    stride = max(1, int(np.add(n1, n2, dtype=np.int64).max()) - 1) if P else 1
    both = torch.empty(2, P, stride, dtype=torch.int32, device=dev)      # (one allocation: the host's share of a call is wall time)
    path1, path2 = both[0], both[1]
    if P == 0:
        return DtwBatchResult(path1, path2, torch.zeros(0, dtype=torch.int32, device=dev), torch.zeros(0, dtype=torch.float64, device=dev))
    plen = torch.empty(P, dtype=torch.int32, device=dev)         # (abn_dtw_batched clears both: dropped and empty pairs keep 0)
    cost = torch.empty(P, dtype=torch.float64, device=dev)
    vp = ctypes.c_void_p
    a = lambda arr: arr.ctypes.data_as(vp)
    hs_bytes = lib.abn_dtw_host_stage_bytes(a(n1), a(n2), P)
    scratch = _DtwScratch.get(dev)
    # the call sizes its workspace in the pass over the pairs it makes anyway and says so when the cached one is too
    # small: only then is abn_dtw_ws_bytes asked (a pass of its own) and the workspace grown
    ws, host_stage = scratch.buffers(scratch.ws.numel() if scratch.ws is not None else
                                     lib.abn_dtw_ws_bytes(a(n1), a(n2), P, feats1.shape[0], feats2.shape[0]), hs_bytes, dev)
    # (the side stream's work is ordered inside the call, behind this stream's uploads and in front of its last launch:
    # every buffer is used in this stream's order, nothing to record for the allocator)
    side = scratch.side_stream(dev)
    for attempt in (0, 1):
        rc = lib.abn_dtw_batched_overlap(
            _lib.ptr(feats1), feats1.shape[0], _lib.ptr(feats2), feats2.shape[0],
            a(off1), a(n1), a(off2), a(n2), P, feats1.shape[1], _lib.ptr(path1),
            _lib.ptr(path2), _lib.ptr(plen), stride, _lib.ptr(cost), _lib.ptr(ws), ws.numel(),
            vp(host_stage.data_ptr()), host_stage.numel(), _lib.stream(),
            vp(side.cuda_stream if side is not None else 0))
        if rc != _lib.E_WORKSPACE or attempt:
            break
        ws, host_stage = scratch.buffers(lib.abn_dtw_ws_bytes(a(n1), a(n2), P, feats1.shape[0], feats2.shape[0]), hs_bytes, dev)
    _lib.check(rc, 'abn_dtw_batched_overlap')
    scratch.mark()
    return DtwBatchResult(path1, path2, plen, cost)


def get_dtw_alignment(feat1, feat2):
    """(path1, path2) for one token pair (abnet3/utils.py:147-153).  Raises
    AssertionError when the reference's cosine_distance would (NaN distance)."""
    f1, f2 = _as_device_f32(feat1), _as_device_f32(feat2)
    res = dtw_align_batch(f1, [0], [f1.shape[0]], f2, [0], [f2.shape[0]])
    out = res.to_lists()[0]
    assert out is not None, 'cosine_distance produced NaN / negative entries'
    path1, path2 = out
    assert len(path1) == len(path2)
    return path1, path2
