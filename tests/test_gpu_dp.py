"""Data parallelism on real hardware without an 8-GPU node: two ranks (fresh processes
started by conftest.py, sharing GPU 0, gloo transport) run the HIP trainer and the
self-sharding loaders; this process computes what ONE process does with the whole
batch / the whole epoch and compares.  Needs an MI355X: run with -m gpu."""
import ast

import numpy as np
import pytest
import torch

import conftest
from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ranks():
    job = conftest.DP_JOB
    if not job:
        pytest.skip('the data-parallel workers were not started (run with -m gpu on a GPU box)')
    conftest.release_dp_job()
    for r, p in enumerate(job['procs']):
        try:
            rc = p.wait(timeout=600)
        except Exception:
            p.kill()
            rc = -9
        log = open(job['out'] + '.rank%d.log' % r).read()
        assert rc == 0, 'rank %d failed (%s):\n%s' % (r, rc, log[-3000:])
    return [dict(np.load(job['out'] + '.rank%d.npz' % r)) for r in range(2)]


@pytest.mark.parametrize('avg', [False, True])
@pytest.mark.parametrize('oname,lr', [('adadelta', 0.1), ('sgd', 0.01)])
def test_two_ranks_equal_one_process_on_the_full_batch(ranks, avg, oname, lr):
    """avg=False: SUM all-reduce, no scaling; avg=True: each rank's loss is a mean over
    its half, the reduced gradient is scaled by 1/2 inside the optimizer kernel.  After
    3 steps both ranks hold the parameters of a single process stepping on all 192 pairs
    (no BatchNorm: per-replica statistics would differ)."""
    import abnet3_amd.loss as L
    from abnet3_amd.model import SiameseNetwork
    from abnet3_amd.trainer import TrainerSiamese
    g = load_golden('train_mid_bn0.npz')
    kw = ast.literal_eval(str(g['kw']))
    rng = np.random.default_rng(123)
    x1 = rng.standard_normal((192, 40)).astype(np.float32)
    x2 = rng.standard_normal((192, 40)).astype(np.float32)
    y = rng.choice([1.0, -1.0], 192)
    net = SiameseNetwork(output_path='/tmp/abn_dp_single', **kw)
    net.load_state_dict({k[2:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith('p.')})
    tr = TrainerSiamese(network=net, loss=L.coscos2(avg=avg), optimizer_type=oname, lr=lr,
                        dataloader=None, log_dir='/tmp/abn_runs_dp')
    assert tr.world_size == 1
    net.train()
    batch = (torch.from_numpy(x1).cuda(), torch.from_numpy(x2).cuda(), torch.from_numpy(y).cuda())
    losses = [float(tr.train_step(batch, True)) for _ in range(3)]
    tag = 'avg%d.%s' % (int(avg), oname)
    both = [r[tag + '.losses'] for r in ranks]
    total = (both[0] + both[1]) * (0.5 if avg else 1.0)
    assert np.allclose(total, losses, rtol=2e-6), (total, losses)
    for k, p in net.named_parameters():
        mine = p.detach().cpu().numpy()
        for r in ranks:
            # 1e-6 of the tensor's scale; the biases start at zero and are still ~1e-5
            # after three steps, so they are judged against the weights' scale (0.05)
            e = rel_err(r[tag + '.p.' + k], mine, floor=0.05)
            assert e < 1e-6, (k, e)
        assert (ranks[0][tag + '.p.' + k] == ranks[1][tag + '.p.' + k]).all()     # replicas stay bit-identical


def test_two_bucket_backward_equals_the_single_call(ranks):
    """Data-parallel steps cut the backward in two (abn_tower_desc.wgrad_part): data gradients + the upper layers' weight
    gradients, all-reduce of that bucket started, the lower layers' weight gradients, second all-reduce.  1600 pairs per
    rank (the single-launch chains): bit-identical to the one-call, one-all-reduce step, and equal to one process on the
    3200 pairs."""
    import abnet3_amd.loss as L
    from abnet3_amd.model import SiameseNetwork
    from abnet3_amd.trainer import TrainerSiamese
    g = load_golden('train_mid_bn0.npz')
    kw = ast.literal_eval(str(g['kw']))
    rng = np.random.default_rng(77)
    x1 = rng.standard_normal((3200, 40)).astype(np.float32)
    x2 = (x1 + 0.3 * rng.standard_normal((3200, 40))).astype(np.float32)
    y = rng.choice([1.0, -1.0], 3200)
    net = SiameseNetwork(output_path='/tmp/abn_dp_c_single', **kw)
    net.load_state_dict({k[2:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith('p.')})
    tr = TrainerSiamese(network=net, loss=L.coscos2(avg=False), optimizer_type='adadelta', lr=0.1,
                        dataloader=None, log_dir='/tmp/abn_runs_dp')
    net.train()
    batch = (torch.from_numpy(x1).cuda(), torch.from_numpy(x2).cuda(), torch.from_numpy(y).cuda())
    losses = [float(tr.train_step(batch, True)) for _ in range(3)]
    for r in ranks:
        assert (r['chain.overlap1.losses'] == r['chain.overlap0.losses']).all()
    total = ranks[0]['chain.overlap1.losses'] + ranks[1]['chain.overlap1.losses']
    assert np.allclose(total, losses, rtol=2e-6), (total, losses)
    for k, p in net.named_parameters():
        mine = p.detach().cpu().numpy()
        for r in ranks:
            assert (r['chain.overlap1.p.' + k] == r['chain.overlap0.p.' + k]).all(), k
            assert rel_err(r['chain.overlap1.p.' + k], mine, floor=0.05) < 2e-6, k


def test_batch_norm_with_cross_replica_statistics_equals_one_process(ranks):
    """TrainerBuilder(sync_batch_norm=True) -> parallel.BatchNormSync -> abn_tower_desc.bn_sync_*: every BatchNorm layer's
    per-call [sum z, sum z^2] (forward) and [sum dy, sum dy xhat] (backward) are all-reduced between two launches, so two
    ranks on 512 pairs each step like ONE process on the 1024 (parameters, running statistics, losses); with the flag off
    every replica normalises with its own rows and does not."""
    import abnet3_amd.loss as L
    from abnet3_amd.model import SiameseNetwork
    from abnet3_amd.trainer import TrainerSiamese
    g = load_golden('train_mid_bn1.npz')
    kw = ast.literal_eval(str(g['kw']))
    rng = np.random.default_rng(321)
    x1 = rng.standard_normal((1024, 40)).astype(np.float32)
    x2 = (x1 + 0.5 * rng.standard_normal((1024, 40))).astype(np.float32)
    y = rng.choice([1.0, -1.0], 1024)
    net = SiameseNetwork(output_path='/tmp/abn_dp_bn_single', **kw)
    net.load_state_dict({k[2:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith('p.')})
    tr = TrainerSiamese(network=net, loss=L.coscos2(avg=False), optimizer_type='adadelta', lr=0.1,
                        dataloader=None, log_dir='/tmp/abn_runs_dp')
    net.train()
    batch = (torch.from_numpy(x1).cuda(), torch.from_numpy(x2).cuda(), torch.from_numpy(y).cuda())
    losses = [float(tr.train_step(batch, True)) for _ in range(3)]
    total = ranks[0]['bn.sync1.losses'] + ranks[1]['bn.sync1.losses']
    assert np.allclose(total, losses, rtol=1e-5), (total, losses)
    # one collective per BatchNorm layer, direction and step on every rank
    n_bn = sum(1 for k in net.state_dict() if k.endswith('running_mean'))
    assert int(ranks[0]['bn.sync1.calls']) == 3 * 2 * n_bn and int(ranks[1]['bn.sync1.calls']) == 3 * 2 * n_bn
    worst_off = 0.0
    for k, v in net.state_dict().items():
        if k.endswith('num_batches_tracked'):
            continue
        mine = v.detach().cpu().numpy()
        for r in ranks:
            e = rel_err(r['bn.sync1.p.' + k], mine, floor=0.05)
            assert e < 2e-5, (k, e)
        assert (ranks[0]['bn.sync1.p.' + k] == ranks[1]['bn.sync1.p.' + k]).all(), k     # replicas stay bit-identical
        worst_off = max(worst_off, rel_err(ranks[0]['bn.sync0.p.' + k], mine, floor=0.05))
    assert worst_off > 1e-4            # per-replica statistics are a different computation
    # batches of different sizes on the ranks (384 + 640 pairs): still one process on the 1024 -- the row count the statistics
    # divide by is all-reduced with the sums (every rank used to divide by ITS rows x the world size)
    total = ranks[0]['bn.uneven.losses'] + ranks[1]['bn.uneven.losses']
    assert np.allclose(total, losses, rtol=1e-5), (total, losses)
    for k, v in net.state_dict().items():
        if k.endswith('num_batches_tracked'):
            continue
        mine = v.detach().cpu().numpy()
        for r in ranks:
            e = rel_err(r['bn.uneven.p.' + k], mine, floor=0.05)
            assert e < 2e-5, (k, e)
        assert (ranks[0]['bn.uneven.p.' + k] == ranks[1]['bn.uneven.p.' + k]).all(), k
    # one rank's batch below the launches that carry the sums: the ranks agree, nobody enters the exchange, both finish
    for r in ranks:
        assert int(r['bn.small.calls']) == 0 and np.isfinite(r['bn.small.losses']).all()
        assert int(r['bn.small.warned']) == 1


def test_one_shot_all_reduce_between_two_processes(ranks):
    """abn_allreduce_oneshot (SURVEY.md 8e's small-message exchange: reduce-scatter + all-gather in two hops over peer-mapped
    mailboxes, summed in rank order): two fresh processes map each other's mailbox through hipIpcMemHandle on this box's one
    GPU and reduce buckets of several sizes beside torch.distributed's all-reduce -- the same sums (fp32: two terms, exact
    whatever the order), bit-identical on both ranks, call after call; three trainer steps with ABN_ONESHOT_ALLREDUCE=1 end
    where the same steps over gloo end.  What this does not show: anything about xGMI (one GPU here)."""
    if not all(int(r['oneshot.available']) for r in ranks):
        pytest.skip('peer mapping over hipIpcMemHandle between two processes on one device is not available here: %s'
                    % [str(r.get('oneshot.why')) for r in ranks])
    for r in ranks:
        assert float(r['oneshot.worst_rel']) == 0.0
        assert int(r['oneshot.replicas_identical']) == 1 and int(r['oneshot.calls']) == 6
        print('one-shot all-reduce of 571 712 floats, two processes on one GPU: %.1f us per call' % float(r['oneshot.us_per_call']))
        assert float(r['oneshot.train_losses_diff']) < 1e-6 and float(r['oneshot.train_param_diff']) < 1e-6


def _corpus():
    gl = load_golden('frames_loader.npz')
    feats = {k[5:]: v for k, v in gl.items() if k.startswith('feat.')}
    times = {k: np.arange(len(v)) * 0.01 + 0.0025 for k, v in feats.items()}

    def parse(line):
        t = str(line).split(' ')
        return (t[0], float(t[1]), float(t[2]), t[3], float(t[4]), float(t[5]), t[6])
    return feats, times, [parse(l) for l in gl['train_pairs']], [parse(l) for l in gl['dev_pairs']]


def test_frames_loader_shards_one_order_disjoint_and_complete(ranks):
    """Ranks start from different numpy RNG states; rank 0's shuffles are broadcast, the
    DTW work is split (each rank aligns every other 'same' pair and the index lists are
    all-gathered), rank r takes batches r, r+2, ...  Interleaving the two ranks' batches
    must give exactly the single-process epoch run from rank 0's seed."""
    from abnet3_amd.dataloader import FramesDataLoader
    feats, times, train, devp = _corpus()
    np.random.seed(1000)
    dl = FramesDataLoader('unused', 'unused', batch_size=25)
    dl.set_data(feats, times, train, devp)
    for ep, mode in enumerate('TTD'):
        ref = [(np.concatenate([a.cpu().numpy(), b.cpu().numpy()], axis=1), c.cpu().numpy())
               for a, b, c in dl.batch_iterator(train_mode=(mode == 'T'))]
        n = [len(r['frames.ep%d.y' % ep]) for r in ranks]
        if mode == 'T':
            assert n[0] == n[1] == len(ref) // 2              # same number of steps on every rank
        else:
            assert n[0] + n[1] == len(ref)                    # the dev pass keeps its tail
        for i in range(n[0] + n[1]):
            r, k = i % 2, i // 2
            assert (ranks[r]['frames.ep%d.y' % ep][k] == ref[i][1]).all(), (ep, i)
            assert (ranks[r]['frames.ep%d.x' % ep][k] == ref[i][0]).all(), (ep, i)


def test_word_pair_loader_shards_batches_by_rank(ranks):
    from abnet3_amd.dataloader import OriginalDataLoader
    feats, times, train, devp = _corpus()
    np.random.seed(2000)
    ol = OriginalDataLoader('unused', 'unused', batch_size=2)
    ol.set_data(feats, times, train, devp)
    for ep, mode in enumerate('TD'):
        ref = [(len(c), a[0].cpu().numpy()) for a, b, c in ol.batch_iterator(train_mode=(mode == 'T'))]
        n = [len(r['words.ep%d.sizes' % ep]) for r in ranks]
        assert (n[0] == n[1] == len(ref) // 2) if mode == 'T' else (n[0] + n[1] == len(ref))
        for i in range(n[0] + n[1]):
            r, k = i % 2, i // 2
            assert ranks[r]['words.ep%d.sizes' % ep][k] == ref[i][0]
            assert (ranks[r]['words.ep%d.first' % ep][k] == ref[i][1]).all()


def test_bench_runs_as_two_ranks():
    """bench.py --gpus 2 exactly as the driver launches it for N > 1 (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
    the environment), rehearsed with two fresh processes on GPU 0 over gloo: barriers, the MAX over ranks of the
    timed region, rank 0's single JSON line."""
    import json
    job = conftest.DP_JOB
    if not job.get('bench'):
        pytest.skip('the bench ranks were not started (run with -m gpu on a GPU box)')
    conftest.release_dp_job()
    logs = []
    for r, p in enumerate(job['bench']):
        try:
            rc = p.wait(timeout=900)
        except Exception:
            p.kill()
            rc = -9
        logs.append(open(job['out'] + '.bench%d.log' % r).read())
        assert rc == 0, 'bench rank %d failed (%s):\n%s' % (r, rc, logs[-1][-3000:])
    lines = [l for l in logs[0].splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1 and not [l for l in logs[1].splitlines() if l.startswith('{"metric"')]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['global_pairs'] == 8192 and out['config']['parallelism'] == 'dp2'
    assert out['steps'] == 5 and out['warmup'] == 2 and out['scaling'] == 'weak' and out['higher_is_better'] is True
    assert np.isfinite(out['value']) and out['value'] > 0 and np.isfinite(out['last_loss'])
    assert abs(out['value'] - 8192 / (out['ms_per_step'] * 1e-3)) <= 1e-3 * out['value']
    assert 'cpu_baseline' not in out and out['dtw']['pairs_per_gpu'] == 200
    # who the ranks are and what the gradient exchange costs on its own: here two
    # processes on ONE GPU over gloo + IPC mailboxes -- the fields must exist and say so
    co = out['collective']
    assert co['backend'] == 'gloo' and co['world'] == 2 and co['rank_count_seen_by_rccl'] is None and 'NOT RCCL' in co['note']
    assert co["bucket_bytes"] == 4 * 571840 and 'torch.distributed.all_reduce' in co['step_uses']
    us = co['allreduce_us']
    assert np.isfinite(us['torch.distributed.all_reduce']) and us['torch.distributed.all_reduce'] > 0
    assert isinstance(us['abn_allreduce_oneshot'], float) and us['abn_allreduce_oneshot'] > 0, us      # the mailboxes mapped between the two processes
    assert co['oneshot_gave_up'] is False


def test_planned_passes_under_two_ranks(ranks):
    """An epoch of planned passes (captured forward + backward per bucket, the gradient all-reduce and the
    optimizer outside the graph) on two ranks: the replicas stay bit-identical and the loss falls."""
    for k in ranks[0]:
        if k.startswith('planned.p.'):
            assert (ranks[0][k] == ranks[1][k]).all(), k
    for r in ranks:
        tl = r['planned.train_losses']
        assert np.isfinite(tl).all() and len(tl) == 4 and tl[-1] < tl[0]
        assert int(r['planned.graphs']) >= 1


def test_one_rank_on_rccl():
    """The data-parallel step on the REAL backend: torch.distributed 'nccl' = RCCL with a process group of one on this
    box's single GPU (tests/rccl_worker.py, a fresh process).  Every collective of the step executes -- the two
    gradient buckets (the first all-reduce asynchronous, under the lower layers' weight gradients), BatchNormSync's
    per-layer float64 all-reduces on the launch stream, the planned passes with the optimizer outside the graph, the
    parameter broadcast, the loaders' agreements -- and over one rank each is the identity: the run equals the run
    without a process group BIT FOR BIT.  (No scaling is measured here: that takes the driver's 8-GPU node.)"""
    job = conftest.DP_JOB
    if 'rccl' not in job:
        pytest.skip('the RCCL worker was not started (run with -m gpu on a GPU box)')
    conftest.release_dp_job()
    p = job['rccl']
    try:
        rc = p.wait(timeout=900)
    except Exception:
        p.kill()
        rc = -9
    log = open(job['out'] + '.rccl.log').read()
    assert rc == 0, 'RCCL worker failed (%s):\n%s' % (rc, log[-4000:])
    r = dict(np.load(job['out'] + '.rccl.npz'))
    assert str(r['backend']) == 'nccl'
    # bench.py's collective leg ran on RCCL: RCCL itself counts one rank in the communicator the leg made, the all-reduce was timed
    assert str(r['collective.backend']) == 'nccl' and str(r['collective.rank_count']) == '1', (r['collective.backend'], r['collective.rank_count'])
    assert 'torch.distributed.all_reduce' in str(r['collective.allreduce_us'])
    keys = sorted(k[6:] for k in r if k.startswith('plain.'))
    assert keys and keys == sorted(k[5:] for k in r if k.startswith('rccl.'))
    off = {}
    for k in keys:
        a, b = r['plain.' + k], r['rccl.' + k]
        assert a.shape == b.shape and a.dtype == b.dtype, k
        if not np.array_equal(a, b, equal_nan=False):
            off[k] = float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max())
    assert not off, off
    # the paths that were meant to run, ran
    assert int(r['rccl.bn.calls']) >= 3 * 2 * 3        # 3 steps x (forward + backward) x one all-reduce per BatchNorm layer
    assert int(r['bn_native']) == 1                    # ... issued by the library itself (abn_rccl_allreduce_f64), not by a Python callback
    assert int(r['rccl.planned.graphs']) >= 1
    assert r['rccl.planned.train_losses'][-1] < r['rccl.planned.train_losses'][0]
    assert int(r['rccl.bntrain.warned']) == 1 and int(r['rccl.bntrain.params_file']) == 1
    assert np.isfinite(r['rccl.bntrain.train_losses']).all()
    # a step with cross-replica BatchNorm statistics captured into a hipGraph (the exchange is the library's own stream-ordered
    # ncclAllReduce): its replays against eager steps of the same start
    assert np.allclose(r['graphbn.replay_losses'], r['graphbn.eager_losses'], rtol=2e-6), (r['graphbn.replay_losses'], r['graphbn.eager_losses'])
    assert float(r['graphbn.worst_param_diff']) < 2e-6
    # RCCL that cannot be loaded: the Python callback, and a warning
    assert str(r['fallback.type']) == 'BatchNormSync' and int(r['fallback.warned']) == 1
