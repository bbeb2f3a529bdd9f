"""Randomised differential run of the operand-plane kernels: random tower shapes (widths multiples of 4 up to 512, 0..5
hidden layers, every activation, BatchNorm on / off), row counts (ragged, tiny, a few thousand), forward(x1, x2) and
forward_once(x), chains and layer-per-launch kernels -- the default arithmetic (and bf16 x 3) against the exact-fp32 mode:
embeddings per tensor maximum, every gradient tensor.   python tools/fuzz_planes.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['ABN_FUSED_MIN_ROWS'] = '0'
import numpy as np, torch
from abnet3_amd.model import SiameseNetwork
from abnet3_amd import _lib


def run(cases, seed, verbose=True):
  rng = np.random.default_rng(seed)
  lib = _lib.load()
  reload_sw = lib.abn_reload_switches
  bad = 0
  lines = []
  for case in range(cases):
      d_in = int(rng.integers(1, 129)) * 4
      hid = int(rng.integers(1, 129)) * 4
      d_out = int(rng.integers(1, 65)) * 4
      nh = int(rng.integers(0, 5))
      act = str(rng.choice(['sigmoid', 'tanh', 'relu']))
      bn = bool(rng.integers(0, 4) == 0)
      B = int(rng.choice([1, 7, 31, 32, 33, 100, 257, 300, 700, 1023, 1024, 1040, 1500, 2100]))
      pair = bool(rng.integers(0, 2))
      wide = int(rng.integers(0, 2))
      want_dx = bool(rng.integers(0, 3) == 0)            # the input's gradient too (the data-gradient chain one product further)
      os.environ['ABN_WIDE'] = str(wide)
      reload_sw()
      kw = dict(input_dim=d_in, num_hidden_layers=nh, hidden_dim=hid, output_dim=d_out, activation_layer=act, p_dropout=0.0, batch_norm=bn)
      if act == 'relu':
          kw['last_non_linearity'] = None
      xa1, xa2 = rng.standard_normal((B, d_in)).astype(np.float32), rng.standard_normal((B, d_in)).astype(np.float32)
      spread = act == 'relu' and not bn and bool(rng.integers(0, 2))
      if spread:      # rows eight orders of magnitude apart, some of them zero (a ReLU tower without BatchNorm is positive homogeneous:
          # the exact-fp32 mode stays a fair yardstick; the embeddings are then judged row by row)
          xa1 *= (10.0 ** rng.uniform(-5, 3, size=(B, 1))).astype(np.float32)
          xa2 *= (10.0 ** rng.uniform(-5, 3, size=(B, 1))).astype(np.float32)
          xa1[::13] = 0.0
      x1, x2 = torch.from_numpy(xa1).cuda(), torch.from_numpy(xa2).cuda()
      outs = {}
      for prec in ('fp32', 'f16x2', 'bf16x3'):
          torch.manual_seed(case)
          net = SiameseNetwork(**kw).cuda()
          net.precision = prec
          net.train()
          xa = x1.clone().requires_grad_(want_dx)
          if pair:
              e1, e2 = net(xa, x2)
              e = torch.cat([e1, e2])
          else:
              e = net.forward_once(xa)
          if 'g' not in outs:
              outs['g'] = torch.from_numpy(rng.standard_normal(tuple(e.shape)).astype(np.float32)).cuda() * 1e-2
          e.backward(outs['g'])
          grads = {k: q.grad.double().cpu().numpy() for k, q in net.named_parameters() if q.grad is not None}
          if want_dx:
              grads['d input'] = xa.grad.double().cpu().numpy()
          outs[prec] = (e.detach().double().cpu().numpy(), grads, _lib.last_forward_path())
      e32, g32, _ = outs['fp32']
      for prec in ('f16x2', 'bf16x3'):
          e, g, path = outs[prec]
          if spread:
              rmax = np.abs(e32).max(axis=1, keepdims=True)
              okr = rmax[:, 0] > 0
              ee = float((np.abs(e - e32)[okr] / rmax[okr]).max()) if okr.any() else 0.0
          else:
              ee = np.abs(e - e32).max() / max(np.abs(e32).max(), 1e-30)
          worst, wk, nbad = 0.0, '', 0
          for k in g32:
              if bn and k.endswith('.bias') and k[:-5] + '.weight' in g32 and g32[k[:-5] + '.weight'].ndim == 2:
                  continue                               # a Linear bias in front of BatchNorm: its true gradient is zero, both sides are noise
              d = np.abs(g[k] - g32[k]) / max(np.abs(g32[k]).max(), 1e-30)
              if act == 'relu':
                  # a pre-activation at ~0 may fall on the other side of relu' in another arithmetic: one batch row's term in a
                  # unit's gradients (and whatever hangs below it) -- the tensor is judged as a whole
                  if not np.isfinite(d).all() or np.linalg.norm(g[k] - g32[k]) > 3e-2 * max(np.linalg.norm(g32[k]), 1e-30):
                      nbad += 1
              elif int((d > 2e-4).sum()) > 0 or not np.isfinite(d).all():
                  nbad += 1
              if d.max() > worst:
                  worst, wk = d.max(), k
          ok = ee < 5e-5 and nbad == 0 and np.isfinite(e).all()
          if not ok:
              bad += 1
          line = ('%s case %3d %s in %3d hid %3d x%d out %3d %-7s bn %d B %4d pair %d wide %d path %d | emb %.1e  worst grad %.1e %s%s'
                % ('ok ' if ok else 'BAD', case, prec, d_in, hid, nh, d_out, act, bn, B, pair, wide, path, ee, worst, wk, (' spread' if spread else '') + ('' if ok else '  <<<<<')))
          lines.append(line)
          if verbose or not ok:
              print(line, flush=True)
  os.environ.pop('ABN_WIDE', None)
  reload_sw()
  return bad, lines


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    bad, _ = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print('cases', n, 'bad', bad)
    sys.exit(1 if bad else 0)
