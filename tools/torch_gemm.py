import torch, time
def t(fn, reps=50):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps
torch.backends.cuda.matmul.allow_tf32 = False
for (m, k, n) in [(8192, 500, 500), (8192, 512, 512), (8192, 40, 500), (8192, 500, 100), (4096, 4096, 4096)]:
    x = torch.randn(m, k, device='cuda'); w = torch.randn(n, k, device='cuda')
    dt = t(lambda: torch.mm(x, w.t()))
    print('torch.mm fp32 NT  M=%d K=%d N=%d: %.1f us  %.1f TF' % (m, k, n, dt * 1e6, 2.0 * m * k * n / dt / 1e12))
    dz = torch.randn(m, n, device='cuda')
    dt = t(lambda: torch.mm(dz, w))
    print('torch.mm fp32 NN  (dgrad)          : %.1f us  %.1f TF' % (dt * 1e6, 2.0 * m * k * n / dt / 1e12))
    dt = t(lambda: torch.mm(dz.t(), x))
    print('torch.mm fp32 TN  (wgrad)          : %.1f us  %.1f TF' % (dt * 1e6, 2.0 * m * k * n / dt / 1e12))
