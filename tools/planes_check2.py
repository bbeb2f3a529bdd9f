import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from abnet3_amd import _lib, model as M
from abnet3_amd.model import SiameseNetwork
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); from planes_decode import decode
rows = int(os.environ.get('ROWS', 64))
os.environ.setdefault('ABN_FUSED_MIN_ROWS', '0')
cfg = dict(bench.C2)
torch.manual_seed(0)
net = SiameseNetwork(output_path='/tmp/abn_pc', **cfg).cuda()
NP = 3 if net.precision == 'bf16x3' else 1
x1, x2 = torch.randn(rows, 40, device='cuda'), torch.randn(rows, 40, device='cuda')
net.train()
out, (seg, sv, gp) = net.direct_forward(x1, x2)
torch.cuda.synchronize()
lib = _lib.load()
lib.abn_debug_planes_offset.restype = ctypes.c_int64
lib.abn_debug_planes_offset.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int]
desc = seg.descriptor(False)
R = 2 * rows
ws = sv.ws
dims = [40, 500, 500, 500, 100]
Ws = [p for k, p in net.named_parameters() if k.endswith('weight')]
u16 = ws.view(torch.int16).cpu().numpy().view(np.uint16)
def steps(c): return ((c + 15) // 16 + 3) // 4 * 4
x = torch.cat([x1, x2]).double().cpu().numpy()
acts = [x]
bs = [p for k, p in net.named_parameters() if k.endswith('bias')]
for l in range(4):
    z = acts[-1] @ Ws[l].detach().double().cpu().numpy().T + bs[l].detach().double().cpu().numpy()
    acts.append(1 / (1 + np.exp(-z)))
print('embedding err %.3e' % np.abs(out.double().cpu().numpy() - acts[4]).max())
for l in range(4):
    W = Ws[l].detach().double().cpu().numpy()
    off = lib.abn_debug_planes_offset(ctypes.byref(desc), R, 2, 0, l)
    d = decode(u16[2 * off:], (dims[l + 1] + 31) // 32, steps(dims[l]), NP)
    print('wp[%d] err %.3e' % (l, np.abs(d[:dims[l + 1], :dims[l]] - W).max()), 'pad', np.abs(d[dims[l + 1]:]).max() if d.shape[0] > dims[l+1] else 0, np.abs(d[:, dims[l]:]).max())
    if l > 0:
        off = lib.abn_debug_planes_offset(ctypes.byref(desc), R, 2, 1, l)
        d = decode(u16[2 * off:], (dims[l] + 31) // 32, steps(dims[l + 1]), NP)
        print('wpt[%d] err %.3e' % (l, np.abs(d[:dims[l], :dims[l + 1]] - W.T).max()))
    # transposed planes of the layer's input (+ ones)
    off = lib.abn_debug_planes_offset(ctypes.byref(desc), R, 2, 2, l)
    nsteps = (R + 31) // 32 * 2
    d = decode(u16[2 * off:], (dims[l] + 1 + 31) // 32, nsteps, NP)
    a = acts[l]
    print('tp[%d] err %.3e  ones err %.3e  beyond %.3e' % (l, np.abs(d[:dims[l], :R] - a.T).max(), np.abs(d[dims[l], :R] - 1).max(), np.abs(d[dims[l] + 1:]).max()))

# backward: dz planes
dz_top = torch.randn(R, 100, device='cuda') * 1e-3
for p_ in net.parameters(): p_.grad = None
lib2 = lib
grads, _, pending = M._segment_backward(seg, sv, dz_top, gp, False, True, True)
torch.cuda.synchronize()
desc_b, rows_b, scratch, sfl, gbuf = pending
s16 = scratch.view(torch.int16).cpu().numpy().view(np.uint16)
dz = [None] * 4
dz[3] = dz_top.double().cpu().numpy()
for l in (3, 2, 1):
    W = Ws[l].detach().double().cpu().numpy()
    a = acts[l]
    dz[l - 1] = (dz[l] @ W) * a * (1 - a)
nsteps = (R + 31) // 32 * 2
for l in range(4):
    off = lib.abn_debug_planes_offset(ctypes.byref(desc_b), R, 2, 3, l)
    d = decode(s16[2 * off:], (dims[l + 1] + 31) // 32, nsteps, NP)
    print('dzp[%d] rel err %.3e  (max %.3e)' % (l, np.abs(d[:dims[l + 1], :R] - dz[l].T).max() / np.abs(dz[l]).max(), np.abs(dz[l]).max()))
