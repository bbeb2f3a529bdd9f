"""pair_loss_kernel (4096 pairs of 100-d embeddings, loss + both gradients) in the library it is run with: us per
launch and the loss / gradient checksums -- for the A/B of the ticket's memory ordering (tools/variants.sh loss
"-DABN_STRICT_FENCES": the C++-memory-model form against the default write-through + drain form, common.h)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import abnet3_amd.loss as L
torch.manual_seed(0)
e1, e2 = torch.randn(4096, 100, device='cuda'), torch.randn(4096, 100, device='cuda')
y = torch.from_numpy(np.random.default_rng(0).choice([1.0, -1.0], 4096)).cuda()
loss = L.coscos2(avg=False)
for _ in range(20):
    lv, de = loss.value_and_grad(e1, e2, y)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(500):
    lv, de = loss.value_and_grad(e1, e2, y)
b.record(); torch.cuda.synchronize()
print('%s: %.2f us per launch, loss %.9g, sum|de| %.9g' % (os.environ.get('ABNET3_HIP_LIB', 'default'), a.elapsed_time(b) * 2, float(lv), float(de.double().abs().sum())))
