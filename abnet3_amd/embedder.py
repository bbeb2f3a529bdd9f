"""Embedders on MI355X: the class surface of abnet3/embedder.py.

Mirrors (file:line relative to the reference checkout)
  EmbedderBuilder   abnet3/embedder.py:19-50
  EmbedderSiamese   abnet3/embedder.py:53-100
  EmbedderSiameseMultitask  abnet3/embedder.py:103-148
The eval-mode forward of the first tower runs through abn_tower_forward.  The
reference reads/writes h5features files (third-party, absent from this image):
embed() uses that package when it is importable, and embed_features() is the
same loop over in-memory arrays.
"""
import numpy as np
import torch


class EmbedderBuilder:
    """Generic Embedder class for ABnet3 (abnet3/embedder.py:19-50)."""

    def __init__(self, network=None, network_path=None, feature_path=None,
                 output_path=None, cuda=True, batch_size=5000):
        if network is None:
            raise ValueError("network is None.")
        self.network = network
        self.network_path = network_path
        self.feature_path = feature_path
        self.output_path = output_path
        self.cuda = cuda
        self.batch_size = batch_size

    def embed(self):
        raise NotImplementedError('Unimplemented embed for class:',
                                  self.__class__.__name__)


class EmbedderSiamese(EmbedderBuilder):
    """Embedder class for siamese network on monotask (abnet3/embedder.py:53-100)."""

    def __init__(self, *args, **kwargs):
        super(EmbedderSiamese, self).__init__(*args, **kwargs)

    # rows per forward launch: utterances are concatenated up to this many frames
    ROWS_PER_LAUNCH = 65536

    def embed_features(self, feats):
        """What the reference's per-utterance loop (embedder.py:80-96) computes, for a
        list of [T, D] arrays: the list of [T, output_dim] float32 embeddings.

        In eval mode every output row depends on its own input row only (BatchNorm
        uses the running statistics), so the utterances are concatenated into
        launches of up to ROWS_PER_LAUNCH frames instead of one (or several
        `batch_size` chunks) per utterance: the same values up to fp32 summation order
        (large launches take the fused kernel), far fewer launches and copies."""
        self.network.eval()
        self.network.cuda()
        feats = [f if f.dtype == np.float32 else f.astype(np.float32) for f in feats]
        lengths = [len(f) for f in feats]
        out_dim = self.network.output_dim
        total = int(sum(lengths))
        if total == 0:
            return [np.zeros((0, out_dim), np.float32) for _ in feats]
        allrows = np.concatenate([f for f in feats if len(f)], axis=0)
        emb = np.empty((total, out_dim), np.float32)
        with torch.no_grad():
            for r0 in range(0, total, self.ROWS_PER_LAUNCH):
                x = torch.from_numpy(np.ascontiguousarray(allrows[r0:r0 + self.ROWS_PER_LAUNCH])).cuda()
                emb[r0:r0 + len(x)] = self.network.forward_once(x).cpu().numpy()   # first output of network(x, x)
        out, o = [], 0
        for n in lengths:
            out.append(emb[o:o + n])
            o += n
        return out

    def embed_table(self, table):
        """embed_features for features that already sit in HBM as one [frames, D] table (utterances laid end to
        end: rows are independent in eval mode): [frames, output_dim] on the device, ROWS_PER_LAUNCH frames per
        launch."""
        self.network.eval()
        self.network.cuda()
        out = torch.empty(table.shape[0], self.network.output_dim, dtype=torch.float32, device=table.device)
        with torch.no_grad():
            for r0 in range(0, table.shape[0], self.ROWS_PER_LAUNCH):
                out[r0:r0 + self.ROWS_PER_LAUNCH] = self.network.forward_once(table[r0:r0 + self.ROWS_PER_LAUNCH])
        return out

    def embed(self):
        """Embed method to embed features based on a saved network."""
        if self.network_path is not None:
            self.network.load_network(self.network_path)
        print("Done loading network weights")
        try:
            import h5features
        except ImportError:
            raise ImportError('EmbedderSiamese.embed() reads and writes h5features '
                              'files like the reference; the h5features package is '
                              'not installed. Use embed_features() on in-memory arrays.')
        with h5features.Reader(self.feature_path, 'features') as fh:
            features = fh.read()
        items = features.items()
        times = features.labels()
        feats = features.features()
        print("Done loading input feature file")
        embeddings = self.embed_features(feats)
        data = h5features.Data(items, times, embeddings, check=True)
        with h5features.Writer(self.output_path) as fh:
            fh.write(data, 'features')


class EmbedderSiameseMultitask(EmbedderBuilder):
    """Embedder class for siamese network on multitask
    (abnet3/embedder.py:103-148): every utterance goes through the network whole
    (no batch_size chunking in the reference) and yields a speaker and a phone
    embedding; embed() writes <output_path>.spk and <output_path>.phn."""

    def __init__(self, *args, **kwargs):
        super(EmbedderSiameseMultitask, self).__init__(*args, **kwargs)

    ROWS_PER_LAUNCH = 65536

    def embed_features(self, feats):
        """([[T, out] speaker embeddings], [[T, out] phone embeddings]) for a list of
        [T, D] arrays (the loop of embedder.py:130-140); utterances are concatenated
        into launches of up to ROWS_PER_LAUNCH frames (rows are independent in eval
        mode, see EmbedderSiamese.embed_features)."""
        self.network.eval()
        self.network.cuda()
        feats = [f if f.dtype == np.float32 else f.astype(np.float32) for f in feats]
        lengths = [len(f) for f in feats]
        out_dim = self.network.output_dim
        total = int(sum(lengths))
        if total == 0:
            empty = [np.zeros((0, out_dim), np.float32) for _ in feats]
            return empty, [e.copy() for e in empty]
        allrows = np.concatenate([f for f in feats if len(f)], axis=0)
        spk = np.empty((total, out_dim), np.float32)
        phn = np.empty((total, out_dim), np.float32)
        with torch.no_grad():
            for r0 in range(0, total, self.ROWS_PER_LAUNCH):
                x = torch.from_numpy(np.ascontiguousarray(allrows[r0:r0 + self.ROWS_PER_LAUNCH])).cuda()
                emb_spk, emb_phn = self.network.forward_once(x)      # network(x, x)[:2]
                spk[r0:r0 + len(x)] = emb_spk.cpu().numpy()
                phn[r0:r0 + len(x)] = emb_phn.cpu().numpy()
        out_spk, out_phn, o = [], [], 0
        for n in lengths:
            out_spk.append(spk[o:o + n])
            out_phn.append(phn[o:o + n])
            o += n
        return out_spk, out_phn

    def embed(self):
        if self.network_path is not None:
            self.network.load_network(self.network_path)
        try:
            import h5features
        except ImportError:
            raise ImportError('EmbedderSiameseMultitask.embed() reads and writes '
                              'h5features files like the reference; the h5features '
                              'package is not installed. Use embed_features().')
        with h5features.Reader(self.feature_path, 'features') as fh:
            features = fh.read()
        items = features.items()
        times = features.labels()
        feats = features.features()
        embeddings_spk, embeddings_phn = self.embed_features(feats)
        data_spk = h5features.Data(items, times, embeddings_spk, check=True)
        data_phn = h5features.Data(items, times, embeddings_phn, check=True)
        with h5features.Writer(self.output_path + '.spk') as fh:
            fh.write(data_spk, 'features')
        with h5features.Writer(self.output_path + '.phn') as fh:
            fh.write(data_phn, 'features')

