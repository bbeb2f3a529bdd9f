"""A second process that keeps the GPU busy with kernels of its own (uneven load for tools/soak.py): large matrix products
and a bandwidth stream, for SECONDS seconds."""
import sys, time, torch
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30
a = torch.randn(4096, 4096, device='cuda'); b = torch.randn(4096, 4096, device='cuda')
big = torch.empty(64 << 20, device='cuda')
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(20):
        c = a @ b
        big.add_(1.0)
    torch.cuda.synchronize(); n += 20
print('hog: %d rounds' % n)
