"""forward_once (eval) time against the row count, planes kernels (forced) vs per-layer GEMM path."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch, bench
from abnet3_amd.model import SiameseNetwork
R = int(sys.argv[1])
torch.manual_seed(0)
net = SiameseNetwork(output_path='/tmp/abn_sw', **bench.C2).cuda()
x = torch.randn(R, 40, device='cuda')
net.eval()
with torch.no_grad():
    for i in range(50): net.forward_once(x)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(200): net.forward_once(x)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 200)
print('%%.4f' %% best)
''' % ROOT
for R in (64, 256, 1024, 2048, 5000, 8192):
    row = []
    for mr in ('0', '1000000000'):
        env = dict(os.environ, ABN_FUSED_MIN_ROWS=mr)
        out = subprocess.run([sys.executable, '-c', CHILD, str(R)], env=env, capture_output=True, text=True)
        row.append(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
    print('rows %5d: planes %s ms   per-layer %s ms' % (R, row[0], row[1]), flush=True)
