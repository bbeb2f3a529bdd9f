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


class AlignCache(object):
    """The DTW alignments of a loader: key (f1, s1, e1, f2, s2, e2, frames) -> (idx1, idx2) global-row index tensors on
    the device, or None for a dropped pair (an empty token, a NaN distance: dataloader.py:188-191).  A batched DTW call
    returns its paths as two FLAT device tensors; the cache keeps those and, per key, where its path sits in them
    (`span`: chunk, first element, length) -- the per-key views are made when somebody asks for one (the iterator, the
    tests), not for all of a dataset's 10^4 pairs at once, and a batch plan gathers straight from the flat tensors."""

    def __init__(self):
        self.span = {}
        self.chunks = []                     # [(g1, g2)]: flat int64 device tensors of one share of one align_pairs call

    def add_chunk(self, g1, g2):
        self.chunks.append((g1, g2))
        return len(self.chunks) - 1

    def compact(self):
        """All chunks as ONE.  The iterator path calls align_pairs per batch -- a chunk per call, thousands of small tensors
        -- and every plan build walks and concatenates all of them: once compacted, a plan build finds one tensor to gather
        from (views handed out earlier keep their own storage alive)."""
        if len(self.chunks) <= 1:
            return
        base = np.concatenate(([0], np.cumsum([c[0].numel() for c in self.chunks]))).astype(np.int64)
        g1 = torch.cat([c[0] for c in self.chunks])
        g2 = torch.cat([c[1] for c in self.chunks])
        self.span = {k: (None if sp is None else (0, int(base[sp[0]]) + sp[1], sp[2])) for k, sp in self.span.items()}
        self.chunks = [(g1, g2)]

    def put(self, key, chunk, start, n):
        self.span[key] = (chunk, start, n) if n else None

    def _views(self, sp):
        if sp is None:
            return None
        c, st, n = sp
        g1, g2 = self.chunks[c]
        return g1[st:st + n], g2[st:st + n]

    def __contains__(self, key):
        return key in self.span

    def __len__(self):
        return len(self.span)

    def __getitem__(self, key):
        return self._views(self.span[key])

    def get(self, key, default=None):
        if key not in self.span:
            return default
        return self._views(self.span[key])

    def items(self):
        for key, sp in self.span.items():
            yield key, self._views(sp)

    def keys(self):
        return self.span.keys()


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
        self._align = AlignCache()    # (f1,s1,e1,f2,s2,e2,frames) -> (idx1, idx2) | None

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
            lens = np.asarray(lens, dtype=np.int64)
            chunk = self._align.add_chunk(g1, g2)
            starts = np.cumsum(lens) - lens
            for key, st, ln in zip(keys, starts.tolist(), lens.tolist()):
                # ln == 0: an empty token or a NaN distance: the reference's try/except
                # drops the pair (dataloader.py:188-191)
                self._align.put(key, chunk, st, ln)

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
        which batches an epoch visits and in which order does.
        One pass over the pairs collects, per ENTRY of a batch (frames_from_pairs_device's order: the batch's same
        pairs, then its diff pairs), where its rows sit in one flat source -- the DTW calls' flat path tensors for a
        same pair, a host-built index run for a diff pair -- and how many rows / labels it has; everything after that
        is array arithmetic: one gather index for the whole dataset, the per-batch seed-0 permutations applied to it,
        one upload, two device gathers."""
        pairs = self.pairs[mode]
        cached = getattr(self, '_plans', {}).get(mode)
        if cached is not None and cached[0] is pairs and cached[1] == len(pairs):
            return cached[2]
        self.align_pairs([p[:6] for p in pairs if p[6] == 'same'], exchange=True)
        dev = self.features.table.device
        bs = self.batch_size
        P = len(pairs)
        nb = (P + bs - 1) // bs
        self._align.compact()
        span, token = self._align.span, self.features.token
        chunk_base = np.concatenate(([0], np.cumsum([c[0].numel() for c in self._align.chunks]))).astype(np.int64)
        n_aligned = int(chunk_base[-1])
        # entries: (batch, 0 same | 1 diff, offset in the source, rows, labels); diff runs are appended behind the paths
        e_batch, e_kind, e_off, e_rows, e_lab = [], [], [], [], []
        d_a0, d_b0, d_m = [], [], []             # diff pairs cut to the shorter token: runs a0 .. a0 + m, b0 .. b0 + m
        d1, d2 = [], []                          # align_different_words: explicit index arrays
        d_total = 0
        adw = self.align_different_words
        for i, (f1, s1, e1, f2, s2, e2, kind) in enumerate(pairs):
            if (s1 > e1) or (s2 > e2):
                assert kind in ('same', 'diff'), 'Unsupported pair type'
                continue
            if kind == 'same':
                sp = span.get((f1, s1, e1, f2, s2, e2, False))
                if sp is None:
                    continue
                e_batch.append(i // bs); e_kind.append(0); e_off.append(chunk_base[sp[0]] + sp[1]); e_rows.append(sp[2]); e_lab.append(sp[2])
                continue
            assert kind == 'diff', 'Unsupported pair type'
            (a0, n1), (b0, n2) = token(f1, s1, e1), token(f2, s2, e2)
            if adw:
                # the shorter word is stretched along the diagonal (dataloader.py:216-225); ties keep the FIRST, like min() / max()
                if n2 < n1:
                    mn0, mnl, mx0, mxl = b0, n2, a0, n1
                else:
                    mn0, mnl = a0, n1
                    mx0, mxl = (b0, n2) if n2 > n1 else (a0, n1)
                mapping = np.rint(np.linspace(0, mnl - 1, num=mxl)).astype(int)
                d1.append((mx0 + np.arange(mxl)).astype(np.int64)); d2.append((mn0 + mapping).astype(np.int64))
                rows = mxl
            else:
                rows = min(n1, n2)
                d_a0.append(a0); d_b0.append(b0); d_m.append(rows)
            # (the labels count min(n1, n2) frames, dataloader.py:231, whatever the number of rows is)
            e_batch.append(i // bs); e_kind.append(1); e_off.append(n_aligned + d_total); e_rows.append(rows); e_lab.append(min(n1, n2))
            d_total += rows
        E = len(e_batch)
        e_batch = np.asarray(e_batch, dtype=np.int64)
        e_kind = np.asarray(e_kind, dtype=np.int64)
        e_off = np.asarray(e_off, dtype=np.int64)
        e_rows = np.asarray(e_rows, dtype=np.int64)
        e_lab = np.asarray(e_lab, dtype=np.int64)
        # a batch's same pairs come before its diff pairs, each kind in file order
        order = np.argsort(e_batch * 2 + e_kind, kind='stable')
        e_batch, e_kind, e_off, e_rows, e_lab = e_batch[order], e_kind[order], e_off[order], e_rows[order], e_lab[order]
        rows_b = np.bincount(e_batch, weights=e_rows, minlength=nb).astype(np.int64) if E else np.zeros(nb, dtype=np.int64)
        lab_b = np.bincount(e_batch, weights=e_lab, minlength=nb).astype(np.int64) if E else np.zeros(nb, dtype=np.int64)
        offsets = np.concatenate(([0], np.cumsum(lab_b))).astype(np.int64)
        row0 = np.cumsum(rows_b) - rows_b
        empty = torch.zeros(0, dtype=torch.int64, device=dev)
        if E:
            # the diff pairs' index runs (host arithmetic, one upload), behind the aligned paths in the flat source
            if adw:
                dd1 = np.concatenate(d1) if d1 else np.zeros(0, dtype=np.int64)
                dd2 = np.concatenate(d2) if d2 else np.zeros(0, dtype=np.int64)
            else:
                m = np.asarray(d_m, dtype=np.int64)
                ramp = np.arange(int(m.sum()), dtype=np.int64) - np.repeat(np.cumsum(m) - m, m)
                dd1 = np.repeat(np.asarray(d_a0, dtype=np.int64), m) + ramp
                dd2 = np.repeat(np.asarray(d_b0, dtype=np.int64), m) + ramp
            both = torch.from_numpy(np.concatenate((dd1, dd2))).to(dev)
            src1 = torch.cat([c[0] for c in self._align.chunks] + [both[:dd1.size]])
            src2 = torch.cat([c[1] for c in self._align.chunks] + [both[dd1.size:]])
            R = int(e_rows.sum())
            gidx = np.repeat(e_off - (np.cumsum(e_rows) - e_rows), e_rows) + np.arange(R, dtype=np.int64)
            # the reference permutes len(y) = n indices and applies them to the rows AND the labels
            # (dataloader.py:247-255): with align_different_words a batch can hold more rows than labels,
            # and the permutation then draws from its first n rows
            perm = self._seed0_permutation
            perms = [perm(int(n)) for n in lab_b]
            perm_rows = np.concatenate([r0 + q for r0, q in zip(row0.tolist(), perms)])
            perm_lab = np.concatenate([o + q for o, q in zip(offsets[:-1].tolist(), perms)])
            take = torch.from_numpy(gidx[perm_rows]).to(dev)
            i1, i2 = src1[take], src2[take]
            y = torch.from_numpy(np.repeat(np.where(e_kind == 0, 1.0, -1.0), e_lab)[perm_lab]).to(dev)
        else:
            i1, i2, y = empty, empty, torch.zeros(0, dtype=torch.float64, device=dev)
        # (statistics_training counts a pair every time an epoch visits it: plan() adds these per visited batch)
        counts = np.stack([np.bincount(e_batch[e_kind == 0], minlength=nb), np.bincount(e_batch[e_kind == 1], minlength=nb)], axis=1).astype(np.int64).reshape(nb, 2)
        store = (i1, i2, y, offsets, counts, counts.sum(axis=1) > 0)
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
        # where every kept pair's rows sit in one flat source (the DTW calls' flat path tensors, then the diff pairs'
        # index runs): one gather index for the dataset instead of a view and a label array per pair
        self._align.compact()
        span, token = self._align.span, self.features.token
        chunk_base = np.concatenate(([0], np.cumsum([c[0].numel() for c in self._align.chunks]))).astype(np.int64)
        s_off, s_len = [], []
        for f1, s1, e1, f2, s2, e2 in pairs['same']:
            if (s1 > e1) or (s2 > e2):
                continue
            sp = span.get((f1, s1, e1, f2, s2, e2, False))
            if sp is None:
                continue
            s_off.append(chunk_base[sp[0]] + sp[1]); s_len.append(sp[2])
        self.statistics_training['SameType'] += len(s_len)
        d_a0, d_b0, d_m = [], [], []          # diff pairs: index runs cut to the shorter token
        for f1, s1, e1, f2, s2, e2 in pairs['diff']:
            if (s1 > e1) or (s2 > e2):
                continue
            (a0, n1), (b0, n2) = token(f1, s1, e1), token(f2, s2, e2)
            d_a0.append(a0); d_b0.append(b0); d_m.append(min(n1, n2))
        self.statistics_training['DiffType'] += len(d_m)
        if not s_len and not d_m:
            z = torch.zeros(0, dtype=torch.int64, device=dev)
            return z, z, z
        s_off, s_len = np.asarray(s_off, dtype=np.int64), np.asarray(s_len, dtype=np.int64)
        m = np.asarray(d_m, dtype=np.int64)
        n_same, n_diff = int(s_len.sum()), int(m.sum())
        gidx = np.repeat(s_off - (np.cumsum(s_len) - s_len), s_len) + np.arange(n_same, dtype=np.int64)
        ramp = np.arange(n_diff, dtype=np.int64) - np.repeat(np.cumsum(m) - m, m)
        dd1 = np.repeat(np.asarray(d_a0, dtype=np.int64), m) + ramp
        dd2 = np.repeat(np.asarray(d_b0, dtype=np.int64), m) + ramp
        up = torch.from_numpy(np.concatenate((gidx, dd1, dd2))).to(dev)
        take, r1, r2 = up[:n_same], up[n_same:n_same + n_diff], up[n_same + n_diff:]
        if n_same:
            src1 = torch.cat([c[0] for c in self._align.chunks]) if len(self._align.chunks) > 1 else self._align.chunks[0][0]
            src2 = torch.cat([c[1] for c in self._align.chunks]) if len(self._align.chunks) > 1 else self._align.chunks[0][1]
            i1, i2 = torch.cat((src1[take], r1)), torch.cat((src2[take], r2))
        else:
            i1, i2 = r1, r2
        y = torch.cat((torch.ones(n_same, dtype=torch.int64, device=dev), -torch.ones(n_diff, dtype=torch.int64, device=dev)))
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

