"""EmbedderSiamese.embed_features on a corpus of many short utterances."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from abnet3_amd.model import SiameseNetwork
from abnet3_amd.embedder import EmbedderSiamese
rng = np.random.default_rng(0)
feats = [rng.standard_normal((int(n), 40)).astype(np.float32) for n in rng.integers(200, 1000, 4000)]
net = SiameseNetwork(**bench.C2).cuda()
emb = EmbedderSiamese(network=net, feature_path=None, output_path=None)
emb.embed_features(feats[:10])
t0 = time.perf_counter(); out = emb.embed_features(feats); dt = time.perf_counter() - t0
print('%d utterances, %d frames: %.3f s  %.2f M frames/s' % (len(feats), sum(map(len, feats)), dt, sum(map(len, feats)) / dt / 1e6))
