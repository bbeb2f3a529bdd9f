import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


DP_JOB = {}


def release_dp_job():
    """Lets the helper processes pytest_configure started (waiting in tests/wait_then_run.py) go: they take the GPU only
    from here on -- tests/test_gpu_dp.py calls this when it wants their results --, not beside every other test's kernels."""
    go = DP_JOB.get('go')
    if go and not os.path.exists(go):
        open(go, 'w').close()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')
    # tests/test_gpu_dp.py needs two data-parallel ranks as FRESH processes.  A process
    # that has initialised the GPU must not exec another program on these boxes, so the
    # ranks are started HERE, before this (pytest) process has touched the card
    # (torch.cuda.device_count() does not initialise it); the test joins them later.
    expr = config.getoption('markexpr', default='') or ''
    if 'gpu' in expr and 'not gpu' not in expr and os.environ.get('ABN_SKIP_DP_WORKERS') != '1':
        try:
            import torch
            have_gpu = torch.cuda.device_count() > 0
        except Exception:
            have_gpu = False
        if have_gpu:
            import subprocess
            import tempfile
            out = os.path.join(tempfile.mkdtemp(prefix='abn_dp_'), 'res')
            port = str(_free_port())
            procs = []
            # every helper sleeps in tests/wait_then_run.py until release_dp_job() (tests/test_gpu_dp.py): started here because
            # this process may not start programs once it holds the GPU, released there so that they share the card with
            # nothing but each other and the tests that wait for them
            go = out + '.go'
            wait = [sys.executable, os.path.join(ROOT, 'tests', 'wait_then_run.py'), go]
            DP_JOB.update(go=go)
            # (these helper processes share the test box's ONE GPU with each other and with pytest itself: their BatchNorm
            # towers train one launch per layer -- two resident grids dispatched in the same microsecond could each hold part
            # of the CUs and wait out their bounded spins for the rest; the resident tower is under test in
            # this process, tests/test_gpu_bn_tower.py)
            shared_gpu = dict(os.environ, ABN_BN_PERSIST='0')
            for r in range(2):
                log = open(out + '.rank%d.log' % r, 'w')
                procs.append(subprocess.Popen(wait + [os.path.join(ROOT, 'tests', 'dp_worker.py'),
                                               str(r), '2', port, out], stdout=log, stderr=subprocess.STDOUT, env=shared_gpu))
            DP_JOB.update(procs=procs, out=out)
            # ... and bench.py --gpus 2 the way the driver starts it for N > 1 (torch.distributed.run's
            # environment), two fresh ranks sharing GPU 0 over gloo: tests/test_gpu_dp.py reads rank 0's JSON line
            port = str(_free_port())
            bprocs = []
            for r in range(2):
                env = dict(shared_gpu, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=port,
                           ABN_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
                log = open(out + '.bench%d.log' % r, 'w')
                bprocs.append(subprocess.Popen(wait + [os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '2',
                                                '--repeats', '2', '--dtw-pairs', '200', '--pipeline-utts', '0'],
                                               stdout=log, stderr=subprocess.STDOUT, env=env))
            DP_JOB.update(bench=bprocs)
            # ... and ONE rank on the real backend (RCCL, a process group of one on this box's single GPU): it waits for
            # the four processes above to leave the card first (a box allows few processes on its GPU at once)
            log = open(out + '.rccl.log', 'w')
            pids = [str(p.pid) for p in procs + bprocs]
            DP_JOB.update(rccl=subprocess.Popen(wait + [os.path.join(ROOT, 'tests', 'rccl_worker.py'), str(_free_port()), out] + pids,
                                                stdout=log, stderr=subprocess.STDOUT, env=shared_gpu))


def pytest_unconfigure(config):
    for p in DP_JOB.get('procs', []) + DP_JOB.get('bench', []) + ([DP_JOB['rccl']] if 'rccl' in DP_JOB else []):
        if p.poll() is None:
            p.kill()


@pytest.fixture(autouse=True)
def _library_switches(monkeypatch):
    """libabnet3_hip.so reads its A/B switches (ABN_PLANES, ABN_FUSED_MIN_ROWS, ...) from the environment once;
    a test that sets one through monkeypatch gets the library to read them again, and every test starts from
    the environment as it stands (the previous test's changes are undone by then)."""
    lib_path = os.path.join(ROOT, 'abnet3_amd', 'lib', 'libabnet3_hip.so')

    def reload():
        if os.path.exists(lib_path):
            from abnet3_amd import _lib
            _lib.reload_switches()
            _lib.trace_paths = True          # tests ask which kernel family a tower call took (_lib.last_forward_path())
    reload()
    setenv, delenv = monkeypatch.setenv, monkeypatch.delenv

    def setenv_and_reload(name, value, *a, **k):
        setenv(name, value, *a, **k)
        if name.startswith('ABN_'):
            reload()

    def delenv_and_reload(name, *a, **k):
        delenv(name, *a, **k)
        if name.startswith('ABN_'):
            reload()
    monkeypatch.setenv, monkeypatch.delenv = setenv_and_reload, delenv_and_reload
    yield


@pytest.fixture(params=['per_layer', 'fused'])
def forward_path(request, monkeypatch):
    """Runs a test once on the per-layer GEMM forward and once on the fused
    whole-tower kernels (which the library picks by itself from 256 rows up for bf16x3 / bf16, from 6144 for fp32)."""
    monkeypatch.setenv('ABN_FUSED_MIN_ROWS', '0' if request.param == 'fused' else '1000000000')
    return request.param


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def rel_err(a, b, floor=1e-30):
    """max|a-b| / max(max|b|, floor): the '1e-5 relative fp32' bar of
    BASELINE.json is applied per tensor against the tensor's largest magnitude.
    `floor` keeps tensors that are mathematically zero (the Linear bias gradient
    in front of a BatchNorm is pure rounding noise, ~1e-10) from dividing noise
    by noise."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    denom = max(np.abs(b).max(), floor)
    return np.abs(a - b).max() / denom


@pytest.fixture(scope='session')
def golden():
    return load_golden


def is_pre_bn_bias(k, batch_norm):
    if not batch_norm or not k.endswith('.bias'):
        return False
    parts = k.split('.')
    return int(parts[1]) % 4 == 0 if parts[0].startswith('hidden_layers') else parts[1] == '0'


def check_params(mine, ref, keys, batch_norm, tol=1e-5):
    """Parameters after some optimizer steps.  A Linear bias feeding a BatchNorm
    has no effect on the network function and receives a rounding-noise
    gradient, which normalising optimizers (Adam, RMSprop, Adagrad) blow up to
    +-lr steps of arbitrary sign: those entries are not comparable and are
    skipped."""
    for k in keys:
        if is_pre_bn_bias(k, batch_norm):
            continue
        assert rel_err(mine[k], ref[k]) < tol, (k, rel_err(mine[k], ref[k]))


def check_grads(mine, ref, keys, batch_norm, tol=2e-5):
    """Per-tensor relative check of parameter gradients.  With BatchNorm the
    gradient of the Linear bias feeding it is mathematically zero (BN removes
    the column mean), so both sides hold rounding noise there: bound its
    magnitude against the model-wide gradient scale instead of comparing noise
    with noise."""
    gmax = max(np.abs(ref[k]).max() for k in keys)
    for k in keys:
        if is_pre_bn_bias(k, batch_norm):
            assert np.abs(mine[k]).max() <= 1e-4 * gmax, k
            assert np.abs(ref[k]).max() <= 1e-4 * gmax, k
        else:
            # a tensor whose entries are 100x below the model's largest
            # gradient is a heavily cancelled sum: its fp32 error scales with
            # the summands, not with the result
            e = rel_err(mine[k], ref[k], floor=1e-2 * gmax)
            assert e < tol, (k, e)


def check_loss_grads(de1, de2, ref1, ref2, lname, tag=''):
    """Row-wise comparison of d loss/d e against tests/golden/loss_edge.npz.
    Rows span 1e-10 .. 1e4 in gradient magnitude, so each row is judged
    against its own largest entry.  Rows 2-4 hold identical / opposite vectors:
    the true gradient is zero and the reference holds ~1e-10 of rounding noise
    there, so they get an absolute bound.  Row 11 has cos == margin to 1 ulp:
    the hinge subgradient is decided by rounding and is not compared for
    cosmargin(margin=0.5)."""
    for r in range(len(de1)):
        if r in (2, 3, 4):
            assert np.abs(de1[r]).max() < 1e-8 and np.abs(de2[r]).max() < 1e-8, (r, tag)
            continue
        if r == 11 and lname == 'cosmargin' and '_m' not in tag:
            continue
        for mine, ref in ((de1[r], ref1[r]), (de2[r], ref2[r])):
            scale = max(np.abs(ref).max(), 1e-30)
            assert np.abs(mine - ref).max() <= 2e-5 * scale, (r, tag)


@pytest.fixture(params=['bf16x3', 'f16x2'])
def split(request):
    """The two split arithmetics of the operand-plane kernels (SiameseNetwork.precision): three bf16 terms with six
    products, two scaled fp16 terms with three -- a test that takes this fixture runs once in each."""
    return request.param
