"""abn_allreduce_oneshot between TWO processes on one GPU (gloo for the rendezvous, IPC-mapped mailboxes for the data), per
call on the C2 gradient bucket (2.29 MB), for several caps on the launch's workgroups (ABN_ONESHOT_WGS; the default is 32):
  python tools/oneshot_time.py [32 64 128]
Two ranks on ONE GPU share its CUs and its memory: what this says about xGMI is nothing; it prices the launch itself."""
import os, sys, subprocess, socket
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch
    from abnet3_amd import parallel
    rank, world, local = parallel.init_from_env('gloo')
    torch.cuda.set_device(0)
    n = 571840
    one = parallel.OneShotAllReduce(n)
    buf = torch.ones(n, device='cuda')
    dist = torch.distributed
    for _ in range(10):
        one.all_reduce(buf)
    torch.cuda.synchronize(); dist.barrier()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 200
    a.record()
    for _ in range(reps):
        one.all_reduce(buf)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / reps
    buf.fill_(float(rank + 1))
    one.all_reduce(buf)
    ok = bool((buf == 3.0).all()) and not one.failed()
    t = torch.tensor([us], dtype=torch.float64, device='cuda')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print('ABN_ONESHOT_WGS=%s  %.1f us per call (max over the two ranks), exact=%s' % (os.environ.get('ABN_ONESHOT_WGS', '32 (default)'), float(t.item()), ok), flush=True)
    one.close()
    dist.barrier()
    dist.destroy_process_group()


if 'RANK' in os.environ:
    child()
else:
    for wgs in (sys.argv[1:] or ['32', '64', '128']):
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = str(s.getsockname()[1]); s.close()
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)],
                                  env=dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=port,
                                           ABN_ONESHOT_WGS=wgs, HSA_ENABLE_IPC_MODE_LEGACY='0')) for r in range(2)]
        for p in procs:
            p.wait(timeout=600)
