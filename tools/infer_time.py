"""us per inference forward (torch.no_grad(), eval mode) of the C2 tower over ROWS frames (default 8192), with and
without BatchNorm, on the operand-plane kernel and on the per-layer kernels (ABN_PLANES=0), both arithmetics; the
training-mode forward of the no-BN tower beside it (what the inference instantiation drops)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from abnet3_amd.model import SiameseNetwork

rows = int(os.environ.get('ROWS', 8192))
x = torch.randn(rows, 40, device='cuda')


def timed(fn):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
    return best


for bn in (False, True):
    for precision in ('bf16x3', 'bf16'):
        torch.manual_seed(0)
        kw = dict(bench.C2); kw['batch_norm'] = bn
        net = SiameseNetwork(output_path='/tmp/abn_inf', **kw).cuda()
        net.precision = precision
        line = 'batch_norm=%-5s %-7s' % (bn, precision)
        for planes in ('1', '0'):
            os.environ['ABN_PLANES'] = planes
            net.eval()
            with torch.no_grad():
                t = timed(lambda: net.forward_once(x))
            line += '  eval/no_grad planes=%s %7.2f us' % (planes, t)
            if not bn and planes == '1':
                net.train()
                line += '  (train-mode forward %7.2f us)' % timed(lambda: net.forward_once(x))
        print(line, flush=True)
