"""Does this torch/ROCm build support timing events recorded INSIDE a captured graph?"""
import torch
x = torch.randn(4096, 4096, device='cuda')
e0 = torch.cuda.Event(enable_timing=True, external=True)
e1 = torch.cuda.Event(enable_timing=True, external=True)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    y = x @ x
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    z = x + 1
    e0.record()
    y = x @ x
    e1.record()
    w = y + 1
for _ in range(3):
    g.replay()
    torch.cuda.synchronize()
    print('in-graph matmul: %.1f us' % (e0.elapsed_time(e1) * 1e3))
