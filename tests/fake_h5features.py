"""A stand-in for the third-party `h5features` package (absent from this image), for tests/test_gpu_file_seam.py ONLY:
the handful of calls the reference's file-level entry points make -- write(), Reader(...).read() with the Data
accessors, Data(), Writer(...).write() -- over a pickle per file.  It exists to exercise abnet3_amd's own
file-level code (FeaturesGenerator.generate, OriginalDataLoader.load_data, EmbedderSiamese.embed) at the seam where
the reference touches files; it is not the package, says nothing about the HDF5 format, and nothing outside tests/
may import it."""
import os
import pickle

import numpy as np


class Data(object):
    def __init__(self, items, labels, features, check=True):
        items, labels, features = list(items), list(labels), list(features)
        if check:
            assert len(items) == len(labels) == len(features) and len(set(items)) == len(items)
            for t, f in zip(labels, features):
                assert len(t) == len(f), 'one time stamp per frame'
        self._items, self._labels, self._features = items, [np.asarray(t) for t in labels], [np.asarray(f) for f in features]

    def items(self):
        return self._items

    def labels(self):
        return self._labels

    def features(self):
        return self._features

    def dict_features(self):
        return dict(zip(self._items, self._features))

    def dict_labels(self):
        return dict(zip(self._items, self._labels))


def _load(path):
    if not os.path.exists(path):
        return {}
    with open(path, 'rb') as fh:
        return pickle.load(fh)


def _store(path, groups):
    with open(path, 'wb') as fh:
        pickle.dump(groups, fh)


def _append(path, group, data):
    groups = _load(path)
    old = groups.get(group.strip('/'))
    if old is not None:                        # h5features.write appends to an existing group
        data = Data(old.items() + data.items(), old.labels() + data.labels(), old.features() + data.features())
    groups[group.strip('/')] = data
    _store(path, groups)


def write(filename, group, items, times, features):
    _append(filename, group, Data(items, times, features))


class Reader(object):
    def __init__(self, filename, groupname=None):
        self.groups = _load(filename)
        if not self.groups:
            raise IOError('%s: no such h5features file' % filename)
        self.group = (groupname or next(iter(self.groups))).strip('/')

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def read(self, from_item=None):
        d = self.groups[self.group]
        if from_item is None:
            return d
        i = d.items().index(from_item)
        return Data([d.items()[i]], [d.labels()[i]], [d.features()[i]])


class Writer(object):
    def __init__(self, filename):
        self.filename = filename

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def write(self, data, groupname='features'):
        _append(self.filename, groupname, data)
