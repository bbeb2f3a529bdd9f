"""Pair -> frame batch builders on MI355X: the surface of abnet3/dataloader.py.

Mirrors (file:line relative to the reference checkout)
  DataLoader            abnet3/dataloader.py:29-40
  OriginalDataLoader    abnet3/dataloader.py:43-352  (load_frames_from_pairs,
                        batch_iterator, temporal coherence pairs)
  FramesDataLoader      abnet3/dataloader.py:580-739 (align once, frame batches)
What changes underneath: the whole feature corpus lives in HBM as ONE
[frames, D] tensor (DeviceCorpus); "same" word pairs are DTW-aligned on the GPU
in batched launches (utils.dtw_align_batch) and cached, since the alignment of a
pair never changes between epochs (the reference recomputes it every epoch,
dataloader.py:189); frames are gathered on the device (abn_gather_rows), so the
iterator yields device tensors and the trainer's .cuda() is a no-op.  Batch
composition, order, the seed-0 permutation and label dtypes follow the
reference statement by statement.
"""
import os
import random
from collections import defaultdict

import numpy as np
import torch

from . import _lib, parallel
from .utils import (Features_Accessor, dtw_align_batch, group_pairs, read_dataset,
                    read_spkid_file)


class DataLoader:

    def batch_iterator(self, train_mode=True):
        raise NotImplementedError("You must implement batch iterator" +
                                  " in DataLoader class.")

    def whoami(self):
        raise NotImplementedError("You must implement whoami in DataLoader class")


def gather_rows(table, idx):
    """table[idx] on the device (feat[path, :], dataloader.py:204-205)."""
    lib = _lib.load()
    idx = idx.to(torch.int64).contiguous()
    _lib.require_device(table, idx)
    out = torch.empty(idx.numel(), table.shape[1], dtype=torch.float32, device=table.device)
    _lib.check(lib.abn_gather_rows(_lib.ptr(table), _lib.ptr(idx), idx.numel(),
                                   table.shape[1], _lib.ptr(out), _lib.stream()),
               'abn_gather_rows')
    return out


class DeviceCorpus(object):
    """Every utterance's frames as one [total_frames, D] float32 tensor in HBM,
    plus host-side row offsets and frame times (Features_Accessor semantics:
    a token is the frames with on <= t <= off, utils.py:127-131)."""

    def __init__(self, features, times):
        self.names = list(features.keys())
        self.offset, self.length, self.times = {}, {}, {}
        chunks, o = [], 0
        for k in self.names:
            f = np.ascontiguousarray(features[k], dtype=np.float32)
            self.offset[k], self.length[k] = o, f.shape[0]
            self.times[k] = np.asarray(times[k])
            chunks.append(f)
            o += f.shape[0]
        self.dim = chunks[0].shape[1]
        self.table = torch.from_numpy(np.concatenate(chunks, axis=0)).cuda()
        self.total = o
        self._init_lookup()

    @classmethod
    def from_table(cls, table, names, lengths, times):
        """The same corpus from features that are already in HBM (FeaturesGenerator.features_from_waves):
        `table` [sum(lengths), D] float32 on the device, utterance k = its rows
        [sum(lengths[:k]), sum(lengths[:k + 1])); times {name: [T] frame times}."""
        _lib.require_device(table)
        self = cls.__new__(cls)
        self.names = list(names)
        self.offset, self.length, self.times = {}, {}, {}
        o = 0
        for k, n in zip(self.names, lengths):
            self.offset[k], self.length[k] = o, int(n)
            self.times[k] = np.asarray(times[k])
            o += int(n)
        assert o == table.shape[0], 'lengths do not add up to the table'
        self.dim = table.shape[1]
        self.table = table
        self.total = o
        self._init_lookup()
        return self

    def _init_lookup(self):
        self._tokens = {}                    # (name, on, off) -> (first global row, frames)
        self._sorted = {}                    # name -> are its frame times non-decreasing (then: binary search)

    def _name(self, f):
        b = f.encode('UTF-8') if isinstance(f, str) else f
        return b if b in self.offset else f

    def token(self, f, on, off):
        """(first global row, number of frames) of the token (f, on, off)."""
        key = (f, on, off)
        hit = self._tokens.get(key)
        if hit is not None:
            return hit
        k = self._name(f)
        t = self.times[k]
        srt = self._sorted.get(k)
        if srt is None:
            srt = self._sorted[k] = bool(len(t) < 2 or np.all(t[1:] >= t[:-1]))
        if srt:
            # np.where((t >= on) & (t <= off)) of sorted times is one run of frames: two binary searches
            lo, hi = int(np.searchsorted(t, on, 'left')), int(np.searchsorted(t, off, 'right'))
            out = (self.offset[k] + lo, hi - lo) if hi > lo else (self.offset[k], 0)
        else:
            ind = Features_Accessor.get_indices_between(t, on, off)
            if len(ind) == 0:
                out = (self.offset[k], 0)
            else:
                assert ind[-1] - ind[0] + 1 == len(ind), 'frame times must be sorted'
                out = (self.offset[k] + int(ind[0]), len(ind))
        self._tokens[key] = out
        return out

    def token_frames(self, f, frame_on, frame_off):
        k = self._name(f)
        lo, hi, _ = slice(frame_on, frame_off).indices(self.length[k])
        return self.offset[k] + lo, max(0, hi - lo)


class BatchPlan(object):
    """Every batch of a dataset as index lists in HBM: batch b is the frame pairs
    [offsets[b], offsets[b + 1]) of (idx1, idx2, labels) -- global rows of `table` and the pair labels,
    already in the order the reference's iterator would yield them (vstack order, then the seed-0
    permutation, abnet3/dataloader.py:248-255) -- and `order` is the batch ids one pass visits (this
    rank's share).  A trainer that understands plans (TrainerSiamese) gathers batch b with ONE launch
    (abn_gather_pairs) into the buffers of a captured step; materialise(b) gives the tensors the plain
    iterator yields."""

    def __init__(self, table, idx1, idx2, labels, offsets, order, has_arrays=None):
        self.table, self.idx1, self.idx2, self.labels = table, idx1, idx2, labels
        self.offsets = np.asarray(offsets, dtype=np.int64)
        self.order = [int(b) for b in order]
        # batches none of whose word pairs survived (the reference's np.vstack([]) raises ValueError,
        # dataloader.py:247) as opposed to batches of zero frames (empty tokens: an empty batch comes out)
        self.has_arrays = has_arrays

    def __len__(self):
        return len(self.order)

    def span(self, b):
        """(first frame pair, number of frame pairs) of batch b"""
        return int(self.offsets[b]), int(self.offsets[b + 1] - self.offsets[b])

    def materialise(self, b):
        first, n = self.span(b)
        if n == 0 and (self.has_arrays is None or not self.has_arrays[b]):
            raise ValueError('need at least one array to concatenate')
        sl = slice(first, first + n)
        return (gather_rows(self.table, self.idx1[sl]), gather_rows(self.table, self.idx2[sl]), self.labels[sl])

    def __iter__(self):
        for b in self.order:
            yield self.materialise(b)


class OriginalDataLoader(DataLoader):
    """Loads the pairs file created by the sampler, creates the frame pairs and
    shuffles inside the batches (abnet3/dataloader.py:43-352)."""

    TCL_DISTANCE_SAME = [1]
    TCL_DISTANCES_DIFF = [15, 20, 25, 30]

    def __init__(self, pairs_path, features_path, num_max_minibatches=1000,
                 seed=None, batch_size=8, shuffle_between_epochs=False,
                 align_different_words=False,
                 tcl=0.0):
        assert 0 <= tcl < 1
        self.pairs_path = pairs_path
        self.features_path = features_path
        self.statistics_training = defaultdict(int)
        self.seed = seed
        self.num_max_minibatches = num_max_minibatches
        self.batch_size = batch_size
        self.features = None          # DeviceCorpus
        self.shuffle_between_epochs = shuffle_between_epochs
        self.align_different_words = align_different_words
        self.tcl = tcl
        self.train_files = None
        self.pairs = {'train': None, 'dev': None}
        self._align = {}              # (f1,s1,e1,f2,s2,e2) -> (idx1, idx2) | None

    def __getstate__(self):
        return (self.pairs_path, self.features_path, self.statistics_training,
                self.seed, self.num_max_minibatches, self.batch_size)

    def whoami(self):
        return {'params': self.__getstate__(),
                'class_name': self.__class__.__name__}

    # -- data ----------------------------------------------------------------
    def set_data(self, features, times, train_pairs=None, dev_pairs=None):
        """In-memory injection of what load_data() reads from disk: features
        {utt: [T, D]}, times {utt: [T]}, pairs [(f1,s1,e1,f2,s2,e2,type), ...]."""
        self.features = DeviceCorpus(features, times)
        if train_pairs is not None:
            self.pairs['train'] = list(train_pairs)
        if dev_pairs is not None:
            self.pairs['dev'] = list(dev_pairs)
        if self.pairs['train'] is not None:
            self.train_files = list({p[0] for p in self.pairs['train']} |
                                    {p[3] for p in self.pairs['train']})

    def load_data(self):
        """Load only once the features, and the pairs (dataloader.py:125-145)."""
        if self.features is None:
            try:
                import h5features
            except ImportError:
                raise ImportError('features_path is an h5features file and the '
                                  'h5features package is not installed; use '
                                  'set_data() with in-memory arrays')
            with h5features.Reader(self.features_path, 'features') as fh:
                feats = fh.read()
            self.features = DeviceCorpus(feats.dict_features(), feats.dict_labels())
        if self.pairs['train'] is None:
            self.pairs['train'] = read_dataset(
                os.path.join(self.pairs_path, 'train_pairs/dataset'))
        if self.pairs['dev'] is None:
            self.pairs['dev'] = read_dataset(
                os.path.join(self.pairs_path, 'dev_pairs/dataset'))
        self.train_files = list({p[0] for p in self.pairs['train']} |
                                {p[3] for p in self.pairs['train']})

    # -- alignment -------------------------------------------------------------
    def _token(self, f, s, e, frames):
        return self.features.token_frames(f, s, e) if frames else self.features.token(f, s, e)

    # cells (sum of n1*n2) one batched DTW call may cover: bounds the call's workspace whatever the
    # size of the dataset; its [pairs, stride] path arrays (2 x int32) and what _align_chunk makes of
    # them (int64 copies, a mask) are bounded separately: ALIGN_PATH_BYTES per entry, ALIGN_PATH_BUDGET in all
    ALIGN_CELL_BUDGET = 1 << 30
    ALIGN_PATH_BYTES = 28
    ALIGN_PATH_BUDGET = 4 << 30

    shards_itself = True          # see parallel.py: the trainer must not shard again

    def _align_chunk(self, todo):
        """One batched DTW call for the keys in `todo`: (path lengths on the host,
        flat global-row index tensors g1, g2 on the device, concatenated in order)."""
        o1, n1, o2, n2 = [], [], [], []
        for f1, s1, e1, f2, s2, e2, fr in todo:
            a, b = self._token(f1, s1, e1, fr), self._token(f2, s2, e2, fr)
            o1.append(a[0]); n1.append(a[1]); o2.append(b[0]); n2.append(b[1])
        table = self.features.table
        res = dtw_align_batch(table, o1, n1, table, o2, n2)
        lens = res.path_len.cpu().numpy()
        dev = table.device
        mask = res.mask()
        g1 = (res.path1.long() + torch.tensor(o1, device=dev)[:, None])[mask]
        g2 = (res.path2.long() + torch.tensor(o2, device=dev)[:, None])[mask]
        return lens, g1, g2

    def align_pairs(self, same_pairs, frames=False, exchange=False):
        """DTW-aligns every not-yet-cached 'same' pair in batched GPU calls
        (get_dtw_alignment, dataloader.py:189), at most ALIGN_CELL_BUDGET cells per
        call.  Dropped pairs (NaN distance, dataloader.py:188-191) are cached as None.
        exchange=True under torch.distributed: the pairs are split over the ranks
        (pair k goes to rank k % R), each rank aligns its share and the index lists
        are all-gathered, so that every rank ends up with every alignment."""
        rank, ws = parallel.world()
        exchange = exchange and parallel.active()
        keys, queued = [], set()
        for p in same_pairs:
            key = tuple(p) + (frames,)
            if key in queued:
                continue
            f1, s1, e1, f2, s2, e2 = p
            if (s1 > e1) or (s2 > e2):
                continue
            keys.append(key)
            queued.add(key)
        if exchange:
            # The ranks' caches differ (an epoch of sharded batches fills them with rank-specific entries),
            # the pair list does not: agree on the work first -- a key is skipped only when EVERY rank holds
            # it -- so that all ranks cut the same list and nobody leaves before the collectives.
            have = np.array([k in self._align for k in keys], dtype=np.int64)
            have = parallel.all_reduce_min(have)
            todo = [k for k, h in zip(keys, have) if not h]
        else:
            todo = [k for k in keys if k not in self._align]
        if not todo:             # (with exchange: the same list on every rank, see above)
            return
        mine = todo[rank::ws] if exchange else todo
        lens_all, g1_all, g2_all = [], [], []
        chunk, cells, stride = [], 0, 0
        for key in mine + [None]:
            if key is not None:
                f1, s1, e1, f2, s2, e2, fr = key
                a, b = self._token(f1, s1, e1, fr)[1], self._token(f2, s2, e2, fr)[1]
                c = a * b
                st = max(stride, a + b - 1)
                # the [pairs, stride] path arrays (and the gather's temporaries) grow with pairs x the LONGEST
                # path of the chunk, not with the cells: one long token among many short pairs must not blow them up
                if not chunk or (cells + c <= self.ALIGN_CELL_BUDGET and
                                 (len(chunk) + 1) * st * self.ALIGN_PATH_BYTES <= self.ALIGN_PATH_BUDGET):
                    chunk.append(key)
                    cells += c
                    stride = st
                    continue
            if chunk:
                lens, g1, g2 = self._align_chunk(chunk)
                lens_all.append(lens); g1_all.append(g1); g2_all.append(g2)
            chunk, cells, stride = ([key], c, a + b - 1) if key is not None else ([], 0, 0)
        dev = self.features.table.device
        empty = torch.zeros(0, dtype=torch.int64, device=dev)
        lens = np.concatenate(lens_all) if lens_all else np.zeros(0, dtype=np.int32)
        g1 = torch.cat(g1_all) if g1_all else empty
        g2 = torch.cat(g2_all) if g2_all else empty
        if exchange:
            lens_r = parallel.all_gather_varlen(torch.from_numpy(lens.astype(np.int64)))
            g1_r = parallel.all_gather_varlen(g1)
            g2_r = parallel.all_gather_varlen(g2)
            shares = [(todo[r::ws], lens_r[r].cpu().numpy(), g1_r[r], g2_r[r]) for r in range(ws)]
        else:
            shares = [(mine, lens, g1, g2)]
        for keys, lens, g1, g2 in shares:
            lens = [int(v) for v in lens]
            v1, v2 = torch.split(g1, lens), torch.split(g2, lens)      # (one call: thousands of views)
            for key, ln, a, b in zip(keys, lens, v1, v2):
                # ln == 0: an empty token or a NaN distance: the reference's try/except
                # drops the pair (dataloader.py:188-191)
                self._align[key] = (a, b) if ln else None

    def prefetch_alignments(self):
        """Aligns every 'same' pair of the train and dev sets up front, in one
        large batched launch each (what FramesDataLoader.load_all_frames does
        serially, dataloader.py:617-671)."""
        self.load_data()
        for mode in ('train', 'dev'):
            if self.pairs[mode]:
                self.align_pairs(group_pairs(self.pairs[mode])['same'], exchange=True)

    # -- batches ---------------------------------------------------------------
    def same_speaker(self, fid2spk, f1, f2):
        """+1 / -1 speaker label of a word pair.  The reference compares the two
        speaker STRINGS with `is` (dataloader.py:197,237): under CPython that is
        true when both tokens come from the same file (the same dict value), or
        when the ids are equal one-character strings (interned) -- two files of
        one multi-character speaker count as DIFFERENT speakers.  speaker_match =
        'identity' (default) reproduces that; 'equal' compares the ids."""
        a, b = fid2spk[f1], fid2spk[f2]
        if getattr(self, 'speaker_match', 'identity') == 'equal':
            return a == b
        return f1 == f2 or (a == b and len(a) <= 1)

    def frames_from_pairs_device(self, pairs, seed=0, frames=False, fid2spk=None):
        """load_frames_from_pairs (dataloader.py:166-261) producing device
        tensors: X1, X2 float32 [n, D], y float64 [n]; with fid2spk (the
        multitask loader) X1, X2, y_spk, y_phn."""
        self.align_pairs(pairs['same'], frames)
        dev = self.features.table.device
        idx1, idx2, ys, ys_spk = [], [], [], []
        d1, d2 = [], []                       # diff pairs: host index arrays, ONE upload per batch
        for f1, s1, e1, f2, s2, e2 in pairs['same']:
            if (s1 > e1) or (s2 > e2):
                continue
            al = self._align.get((f1, s1, e1, f2, s2, e2, frames))
            if al is None:
                continue
            self.statistics_training['SameType'] += 1
            idx1.append(al[0]); idx2.append(al[1])
            ys.append(np.ones(len(al[0])))
            if fid2spk:
                same = self.same_speaker(fid2spk, f1, f2)
                ys_spk.append((1 if same else -1) * np.ones(len(al[0])))
                self.statistics_training['SameTypeSameSpk' if same else 'SameTypeDiffSpk'] += 1
        for f1, s1, e1, f2, s2, e2 in pairs['diff']:
            if (s1 > e1) or (s2 > e2):
                continue
            (a0, n1), (b0, n2) = self._token(f1, s1, e1, frames), self._token(f2, s2, e2, frames)
            if self.align_different_words:
                # the shorter word is stretched along the diagonal (:216-225);
                # key=len keeps the FIRST on ties, like min()/max()
                if n2 < n1:
                    mn0, mnl, mx0, mxl = b0, n2, a0, n1
                else:
                    mn0, mnl = a0, n1
                    mx0, mxl = (b0, n2) if n2 > n1 else (a0, n1)
                mapping = np.rint(np.linspace(0, mnl - 1, num=mxl)).astype(int)
                w1 = mx0 + np.arange(mxl)
                w2 = mn0 + mapping
            else:
                m = min(n1, n2)
                w1 = a0 + np.arange(m)
                w2 = b0 + np.arange(m)
            d1.append(w1.astype(np.int64))
            d2.append(w2.astype(np.int64))
            ys.append(-1 * np.ones(min(n1, n2)))
            self.statistics_training['DiffType'] += 1
            if fid2spk:
                same = self.same_speaker(fid2spk, f1, f2)
                ys_spk.append((1 if same else -1) * np.ones(min(n1, n2)))
                self.statistics_training['DiffTypeSameSpk' if same else 'DiffTypeDiffSpk'] += 1
        if d1:                                # after the same pairs, like the reference's vstack order
            both = torch.from_numpy(np.concatenate(d1 + d2)).to(dev)
            n_d = both.numel() // 2
            idx1.append(both[:n_d])
            idx2.append(both[n_d:])
        if not idx1:
            raise ValueError('need at least one array to concatenate')
        i1, i2 = torch.cat(idx1), torch.cat(idx2)
        y = np.concatenate(ys)
        np.random.seed(seed)
        ind = np.random.permutation(len(y))
        ind_d = torch.from_numpy(ind).to(dev)
        X1 = gather_rows(self.features.table, i1[ind_d])
        X2 = gather_rows(self.features.table, i2[ind_d])
        if fid2spk:
            y_spk = np.concatenate(ys_spk)
            assert len(y) == len(y_spk), 'not same number of labels...'
            return (X1, X2, torch.from_numpy(y_spk[ind]).to(dev),
                    torch.from_numpy(y[ind]).to(dev))
        return X1, X2, torch.from_numpy(y[ind]).to(dev)

    def load_frames_from_pairs(self, pairs, seed=0, fid2spk=None, frames=False):
        """numpy front end with the reference's return types: (X1, X2, y), or
        (X1, X2, y_spk, y_phn) when a speaker mapping is given."""
        out = self.frames_from_pairs_device(pairs, seed, frames, fid2spk)
        return tuple(t.cpu().numpy() for t in out)

    def batch_iterator(self, train_mode=True):
        """Iterator over (X1, X2, y) batches of `batch_size` WORD pairs
        (dataloader.py:263-312)."""
        self.load_data()
        mode = 'train' if train_mode else 'dev'
        pairs = self.pairs[mode]
        num_pairs = len(pairs)
        if self.shuffle_between_epochs:
            self._shuffle_pairs(pairs)
        batches = [pairs[idx:idx + self.batch_size]
                   for idx in range(0, num_pairs, self.batch_size)]
        selected_batches = self._select_batches(len(batches), train_mode)
        # one batched DTW launch for everything this epoch will touch ON THIS RANK
        self.align_pairs([p[:6] for b in selected_batches for p in batches[b]
                          if p[6] == 'same'])
        for batch_id in selected_batches:
            batch = self.frames_from_pairs_device(group_pairs(batches[batch_id]))
            if self.tcl > 0:
                batch = self.add_tcl_to_batch(batch)
            yield batch

    # -- the same batches as index lists in HBM (BatchPlan) -------------------------------
    _PERMS = {}          # n -> np.random.seed(0); np.random.permutation(n)  (dataloader.py:248-249)

    @classmethod
    def _seed0_permutation(cls, n):
        p = cls._PERMS.get(n)
        if p is None:
            # RandomState(0) is the generator np.random.seed(0) re-creates: the same permutation, and the
            # global stream is left alone until the pass is over (plan(): see there)
            p = cls._PERMS[n] = np.random.RandomState(0).permutation(n)
        return p

    def _plan_store(self, mode):
        """(idx1, idx2, labels, offsets) of EVERY batch of self.pairs[mode] -- batch b = pairs
        [b * batch_size, (b + 1) * batch_size) -- built once per pairs list: the content of a batch never
        changes between epochs (alignments are cached, the permutation is seeded with 0 every time), only
        which batches an epoch visits and in which order does."""
        pairs = self.pairs[mode]
        cached = getattr(self, '_plans', {}).get(mode)
        if cached is not None and cached[0] is pairs and cached[1] == len(pairs):
            return cached[2]
        self.align_pairs([p[:6] for p in pairs if p[6] == 'same'], exchange=True)
        dev = self.features.table.device
        bs = self.batch_size
        nb = (len(pairs) + bs - 1) // bs
        d1, d2 = [], []
        # frames_from_pairs_device's loop, for all batches at once: same pairs first, then diff pairs
        per_batch = [[] for _ in range(nb)]  # [(+1 | -1, the aligned index lists | position in d1 / d2, rows, labels)]
        for b in range(nb):
            chunk = pairs[b * bs:(b + 1) * bs]
            same = [p for p in chunk if p[6] == 'same']
            diff = [p for p in chunk if p[6] == 'diff']
            assert len(same) + len(diff) == len(chunk), 'Unsupported pair type'
            for f1, s1, e1, f2, s2, e2, _ in same:
                if (s1 > e1) or (s2 > e2):
                    continue
                al = self._align.get((f1, s1, e1, f2, s2, e2, False))
                if al is None:
                    continue
                per_batch[b].append((1, al, al[0].shape[0], al[0].shape[0]))
            for f1, s1, e1, f2, s2, e2, _ in diff:
                if (s1 > e1) or (s2 > e2):
                    continue
                (a0, n1), (b0, n2) = self._token(f1, s1, e1, False), self._token(f2, s2, e2, False)
                if self.align_different_words:
                    if n2 < n1:
                        mn0, mnl, mx0, mxl = b0, n2, a0, n1
                    else:
                        mn0, mnl = a0, n1
                        mx0, mxl = (b0, n2) if n2 > n1 else (a0, n1)
                    mapping = np.rint(np.linspace(0, mnl - 1, num=mxl)).astype(int)
                    w1, w2 = mx0 + np.arange(mxl), mn0 + mapping
                else:
                    m = min(n1, n2)
                    w1, w2 = a0 + np.arange(m), b0 + np.arange(m)
                # (the labels count min(n1, n2) frames, dataloader.py:231, whatever the number of rows is)
                per_batch[b].append((-1, len(d1), len(w1), min(n1, n2)))
                d1.append(w1.astype(np.int64)); d2.append(w2.astype(np.int64))
        if d1:
            lens_d = np.array([len(w) for w in d1], dtype=np.int64)
            offs_d = np.concatenate(([0], np.cumsum(lens_d)))
            both = torch.from_numpy(np.concatenate(d1 + d2)).to(dev)
            dd1, dd2 = both[:both.numel() // 2], both[both.numel() // 2:]
        parts1, parts2, lab_kind, lab_n, perm, offsets, row0 = [], [], [], [], [], [0], 0
        for b in range(nb):
            n, rows = 0, 0
            for kind, payload, ln, nlab in per_batch[b]:
                if kind == 1:
                    parts1.append(payload[0]); parts2.append(payload[1])
                else:
                    parts1.append(dd1[offs_d[payload]:offs_d[payload] + ln]); parts2.append(dd2[offs_d[payload]:offs_d[payload] + ln])
                lab_kind.append(float(kind)); lab_n.append(nlab)
                n += nlab
                rows += ln
            # the reference permutes len(y) = n indices and applies them to the rows AND the labels
            # (dataloader.py:247-255): with align_different_words a batch can hold more rows than labels,
            # and the permutation then draws from its first n rows
            perm.append((row0 + self._seed0_permutation(n), offsets[-1] + self._seed0_permutation(n)))
            offsets.append(offsets[-1] + n)
            row0 += rows
        empty = torch.zeros(0, dtype=torch.int64, device=dev)
        if parts1:
            perm_rows = np.concatenate([p[0] for p in perm])
            perm_lab = np.concatenate([p[1] for p in perm])
            perm_d = torch.from_numpy(perm_rows).to(dev)
            i1, i2 = torch.cat(parts1)[perm_d], torch.cat(parts2)[perm_d]
            y = torch.from_numpy(np.repeat(np.asarray(lab_kind), np.asarray(lab_n, dtype=np.int64))[perm_lab]).to(dev)
        else:
            i1, i2, y = empty, empty, torch.zeros(0, dtype=torch.float64, device=dev)
        # (statistics_training counts a pair every time an epoch visits it: plan() adds these per visited batch)
        counts = np.array([[sum(1 for e in pb if e[0] == 1), sum(1 for e in pb if e[0] == -1)] for pb in per_batch], dtype=np.int64).reshape(nb, 2)
        store = (i1, i2, y, np.asarray(offsets, dtype=np.int64), counts, np.array([len(pb) > 0 for pb in per_batch], dtype=bool))
        if not hasattr(self, '_plans'):
            self._plans = {}
        self._plans[mode] = (pairs, len(pairs), store)
        return store

    def plan(self, train_mode=True):
        """batch_iterator(train_mode) as a BatchPlan (same batches, same order, same draws from the global
        RNGs), or None where only the iterator applies (temporal-coherence pairs are drawn per batch)."""
        if self.tcl > 0:
            return None
        self.load_data()
        mode = 'train' if train_mode else 'dev'
        pairs = self.pairs[mode]
        if self.shuffle_between_epochs:
            self._shuffle_pairs(pairs)
            getattr(self, '_plans', {}).pop(mode, None)      # the batches' composition changed
        i1, i2, y, offsets, counts, has_arrays = self._plan_store(mode)
        selected = self._select_batches(len(offsets) - 1, train_mode)
        order = [int(b) for b in selected]
        if order:
            self.statistics_training['SameType'] += int(counts[order, 0].sum())
            self.statistics_training['DiffType'] += int(counts[order, 1].sum())
            # the iterator would leave numpy's global generator where its last batch put it
            # (np.random.seed(0); np.random.permutation(n), dataloader.py:248-249): so does the plan
            np.random.seed(0)
            np.random.permutation(int(offsets[order[-1] + 1] - offsets[order[-1]]))
        return BatchPlan(self.features.table, i1, i2, y, offsets, order, has_arrays)

    @staticmethod
    def _shuffle_pairs(pairs):
        """random.shuffle(pairs) (dataloader.py:276-277); under torch.distributed the
        permutation is rank 0's, whatever the other ranks' `random` state is."""
        if parallel.world()[1] == 1:
            random.shuffle(pairs)
            return
        perm = list(range(len(pairs)))
        random.shuffle(perm)                     # same draws as shuffling the list itself
        perm = parallel.broadcast_array(perm)
        pairs[:] = [pairs[i] for i in perm]

    def _select_batches(self, num_batches, train_mode):
        """The epoch's batch ids in visiting order (dataloader.py:291-299).  Under
        torch.distributed rank 0's draw is broadcast and rank r keeps ids r, r+R, ...
        of it: disjoint, together complete (train: up to R-1 batches at the end are
        left out so that every rank takes the same number of steps)."""
        if self.num_max_minibatches < num_batches:
            selected_batches = np.random.choice(range(num_batches),
                                                self.num_max_minibatches,
                                                replace=False)
        else:
            selected_batches = np.random.permutation(range(num_batches))
        rank, ws = parallel.world()
        if ws > 1:
            selected_batches = parallel.broadcast_array(selected_batches)
            selected_batches = parallel.shard_ids(selected_batches, rank, ws, equal=train_mode)
        return selected_batches

    def add_tcl_to_batch(self, batch):
        X1, X2, Y = batch
        num_pairs = len(Y)
        num_pairs_to_add = int((self.tcl * num_pairs) / (1 - self.tcl))
        X1_tcl, X2_tcl, Y_tcl = self.temporal_coherence_loss(num_pairs_to_add)
        return (torch.cat((X1, X1_tcl)), torch.cat((X2, X2_tcl)),
                torch.cat((Y, Y_tcl.to(Y.dtype))))

    def temporal_coherence_loss(self, num_pairs):
        """Dupoux & Synnaeve (2016) temporal coherence pairs (dataloader.py:324-352)."""
        c = self.features
        i1, i2, Y = [], [], []
        per_it = len(self.TCL_DISTANCES_DIFF) + len(self.TCL_DISTANCE_SAME)
        for _ in range(round(num_pairs / per_it)):
            files = self.train_files if self.train_files is not None else c.names
            f = c._name(random.choice(files))
            t = random.choice(range(c.length[f] - max(self.TCL_DISTANCES_DIFF)))
            for delta in self.TCL_DISTANCE_SAME:
                i1.append(c.offset[f] + t); i2.append(c.offset[f] + t + delta); Y.append(1)
            for delta in self.TCL_DISTANCES_DIFF:
                i1.append(c.offset[f] + t); i2.append(c.offset[f] + t + delta); Y.append(-1)
        dev = c.table.device
        return (gather_rows(c.table, torch.tensor(i1, dtype=torch.int64, device=dev)),
                gather_rows(c.table, torch.tensor(i2, dtype=torch.int64, device=dev)),
                torch.tensor(Y, device=dev))


class FramesDataLoader(OriginalDataLoader):
    """Batches of exactly `batch_size` FRAME pairs, shuffled across the whole
    dataset; alignment happens once (abnet3/dataloader.py:580-739)."""

    def __init__(self, pairs_path, features_path,
                 batch_size=100, randomize_dataset=True, max_batches_per_epoch=None):
        super().__init__(pairs_path, features_path)
        self.randomize_dataset = randomize_dataset
        self.batch_size = batch_size
        self.frame_pairs = {'train': None, 'dev': None}   # (idx1, idx2, y) device
        self.max_batches_per_epoch = max_batches_per_epoch
        if self.max_batches_per_epoch is not None:
            self.batch_position = 0

    def load_data(self):
        super(FramesDataLoader, self).load_data()
        for mode in ('train', 'dev'):
            if self.frame_pairs[mode] is None and self.pairs[mode] is not None:
                self.frame_pairs[mode] = self.load_all_frames(self.pairs[mode])

    def load_all_frames(self, pairs):
        """The frame-pair dataset (global row of frame 1, of frame 2, +-1) for
        all word pairs (dataloader.py:617-671), shuffled once."""
        pairs = group_pairs(pairs)
        self.align_pairs(pairs['same'], exchange=True)     # DTW work split over the ranks
        dev = self.features.table.device
        i1, i2, ys = [], [], []
        for f1, s1, e1, f2, s2, e2 in pairs['same']:
            if (s1 > e1) or (s2 > e2):
                continue
            al = self._align.get((f1, s1, e1, f2, s2, e2, False))
            if al is None:
                continue
            i1.append(al[0]); i2.append(al[1])
            ys.append(np.ones(len(al[0]), dtype=np.int64))
            self.statistics_training['SameType'] += 1
        d1, d2 = [], []                       # diff pairs: host index ranges, one upload for all
        for f1, s1, e1, f2, s2, e2 in pairs['diff']:
            if (s1 > e1) or (s2 > e2):
                continue
            (a0, n1), (b0, n2) = self.features.token(f1, s1, e1), self.features.token(f2, s2, e2)
            m = min(n1, n2)
            d1.append(np.arange(a0, a0 + m, dtype=np.int64))
            d2.append(np.arange(b0, b0 + m, dtype=np.int64))
            ys.append(-np.ones(m, dtype=np.int64))
            self.statistics_training['DiffType'] += 1
        if d1:
            both = torch.from_numpy(np.concatenate(d1 + d2)).to(dev)
            i1.append(both[:both.numel() // 2])
            i2.append(both[both.numel() // 2:])
        if not i1:
            z = torch.zeros(0, dtype=torch.int64, device=dev)
            return z, z, z
        i1, i2 = torch.cat(i1), torch.cat(i2)
        y = torch.from_numpy(np.concatenate(ys)).to(dev)
        return self._shuffle((i1, i2, y))

    @staticmethod
    def _shuffle(fp):
        # np.random.shuffle(list) and np.random.permutation(n) draw the same
        # permutation from the same global state (dataloader.py:670)
        # (under torch.distributed every rank applies rank 0's permutation)
        perm = parallel.broadcast_array(np.random.permutation(len(fp[2])))
        perm = torch.from_numpy(perm).to(fp[2].device)
        return fp[0][perm], fp[1][perm], fp[2][perm]

    def load_batch(self, sl, mode):
        i1, i2, y = self.frame_pairs[mode]
        return (gather_rows(self.features.table, i1[sl]),
                gather_rows(self.features.table, i2[sl]), y[sl])

    def _batch_ids(self, train_mode):
        """The shuffles and the batch ids of one pass (dataloader.py:686-739), this rank's share."""
        self.load_data()
        mode = 'train' if train_mode else 'dev'
        num_pairs = len(self.frame_pairs[mode][2])
        num_batches = num_pairs // self.batch_size
        if num_batches == 0:
            num_batches = 1
        if mode == 'dev' or self.max_batches_per_epoch is None:
            batch_ids = range(num_batches)
            if self.randomize_dataset:
                self.frame_pairs[mode] = self._shuffle(self.frame_pairs[mode])
        else:
            if self.batch_position >= num_batches:
                if self.randomize_dataset:
                    self.frame_pairs[mode] = self._shuffle(self.frame_pairs[mode])
                self.batch_position = 0
            batch_ids = range(self.batch_position,
                              min(self.batch_position + self.max_batches_per_epoch,
                                  num_batches))
            self.batch_position += self.max_batches_per_epoch
        rank, ws = parallel.world()
        if ws > 1:       # rank r gathers batches r, r+R, ... of the epoch's (shared) order
            batch_ids = parallel.shard_ids(list(batch_ids), rank, ws, equal=train_mode)
        return list(batch_ids)

    def batch_iterator(self, train_mode=True):
        """(dataloader.py:686-739)"""
        mode = 'train' if train_mode else 'dev'
        for i in self._batch_ids(train_mode):
            yield self.load_batch(slice(i * self.batch_size,
                                        i * self.batch_size + self.batch_size), mode)

    def plan(self, train_mode=True):
        """batch_iterator(train_mode) as a BatchPlan: the same shuffles, batch b = frame pairs
        [b * batch_size, (b + 1) * batch_size) of the (re)shuffled dataset."""
        ids = self._batch_ids(train_mode)
        mode = 'train' if train_mode else 'dev'
        i1, i2, y = self.frame_pairs[mode]
        num_pairs = len(y)
        num_batches = max(1, num_pairs // self.batch_size)
        offsets = np.minimum(np.arange(num_batches + 1, dtype=np.int64) * self.batch_size, num_pairs)
        return BatchPlan(self.features.table, i1, i2, y, offsets, ids)


class MultiTaskDataLoader(OriginalDataLoader):
    """Batches (X1, X2, y_spk, y_phn) for the multitask siamese network
    (abnet3/dataloader.py:742-792).  fid2spk_file: lines "<file id> <speaker id>".
    speaker_match: see OriginalDataLoader.same_speaker."""

    def __init__(self, pairs_path, features_path, fid2spk_file=None,
                 speaker_match='identity', **kwargs):
        super().__init__(pairs_path, features_path, **kwargs)
        assert speaker_match in ('identity', 'equal')
        self.fid2spk_file = fid2spk_file
        self.speaker_match = speaker_match

    def plan(self, train_mode=True):
        return None                    # (four tensors per batch: the iterator)

    def batch_iterator(self, train_mode=True):
        self.load_data()
        mode = 'train' if train_mode else 'dev'
        pairs = self.pairs[mode]
        num_pairs = len(pairs)
        batches = [pairs[idx:idx + self.batch_size]
                   for idx in range(0, num_pairs, self.batch_size)]
        fid2spk = read_spkid_file(self.fid2spk_file)
        selected_batches = self._select_batches(len(batches), train_mode)
        self.align_pairs([p[:6] for b in selected_batches for p in batches[b]
                          if p[6] == 'same'])
        for idx in selected_batches:
            yield self.frames_from_pairs_device(group_pairs(batches[idx]), fid2spk=fid2spk)

