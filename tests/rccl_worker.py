"""ONE rank on the real backend: torch.distributed's 'nccl' (= RCCL on ROCm) with a process group of one, on the
single GPU of a test box.  Started as a FRESH process by conftest.py (before pytest touches the GPU); waits for the
other helper processes to leave the card, then runs every data-parallel code path twice -- without a process group,
and under the one-rank RCCL group with ABN_DP_SINGLE_RANK=1 (parallel.active(): the collectives execute, and over
one rank each is the identity) -- and writes both results; tests/test_gpu_dp.py::test_one_rank_on_rccl compares
them bit for bit.

    python tests/rccl_worker.py PORT OUT_PREFIX [PID_TO_WAIT_FOR ...]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def alive(pid):
    """Is the process still running (a zombie waiting for its parent's wait() has left the GPU: not alive)?"""
    try:
        with open('/proc/%s/stat' % pid) as fh:
            return fh.read().split(')')[-1].split()[0] != 'Z'
    except (FileNotFoundError, ProcessLookupError, IndexError):
        return False


def wait_for(pids, limit=600.0):
    t0 = time.time()
    while time.time() - t0 < limit and any(alive(p) for p in pids):
        time.sleep(0.5)


def main():
    port, out = sys.argv[1], sys.argv[2]
    wait_for(sys.argv[3:])
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=port, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
                      HSA_ENABLE_IPC_MODE_LEGACY='0', ABN_DP_SINGLE_RANK='1',
                      # (one process sums a small batch's weight gradients and steps in ONE launch -- tower_wgrad_step.h: all rows in
                      # one sum --, a data-parallel rank in slabs + all-reduce + step: the same products in another order.  What is
                      # compared bit for bit here is a rank's collectives against no collectives: both runs on the slab launches.)
                      ABN_WGRAD_STEP='0')
    import ast
    import warnings
    import numpy as np
    import torch
    from abnet3_amd import parallel
    import abnet3_amd.loss as L
    from abnet3_amd.model import SiameseNetwork
    from abnet3_amd.trainer import TrainerSiamese
    from abnet3_amd.dataloader import OriginalDataLoader
    from conftest import load_golden
    res = {}

    g = load_golden('train_mid_bn0.npz')
    kw = ast.literal_eval(str(g['kw']))
    gb = load_golden('train_mid_bn1.npz')
    kwb = ast.literal_eval(str(gb['kw']))
    gl = load_golden('frames_loader.npz')
    feats = {k[5:]: v for k, v in gl.items() if k.startswith('feat.')}
    times = {k: np.arange(len(v)) * 0.01 + 0.0025 for k, v in feats.items()}

    def parse(line):
        t = str(line).split(' ')
        return (t[0], float(t[1]), float(t[2]), t[3], float(t[4]), float(t[5]), t[6])
    train = [parse(l) for l in gl['train_pairs']]
    devp = [parse(l) for l in gl['dev_pairs']]

    def run(tag):
        grouped = parallel.active()
        # (1) the single-launch chains, 1600 pairs: the backward in two calls, the upper bucket's all-reduce in
        # flight under the second (async_op on RCCL's own stream), then the lower bucket's
        rng = np.random.default_rng(77)
        B = 1600
        x1 = rng.standard_normal((B, 40)).astype(np.float32)
        x2 = (x1 + 0.3 * rng.standard_normal((B, 40))).astype(np.float32)
        y = rng.choice([1.0, -1.0], B)
        net = SiameseNetwork(output_path='/tmp/abn_rccl_c', **kw)
        net.load_state_dict({k[2:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith('p.')})
        tr = TrainerSiamese(network=net, loss=L.coscos2(avg=False), optimizer_type='adadelta', lr=0.1,
                            dataloader=None, log_dir='/tmp/abn_runs_rccl')
        assert tr.dp == grouped and tr.overlap_allreduce
        net.train()
        batch = (torch.from_numpy(x1).cuda(), torch.from_numpy(x2).cuda(), torch.from_numpy(y).cuda())
        res[tag + '.chain.losses'] = np.array([float(tr.train_step(batch, True)) for _ in range(3)])
        for k, p in net.named_parameters():
            res[tag + '.chain.p.' + k] = p.detach().cpu().numpy()
        # (2) BatchNorm with the statistics going through BatchNormSync (one all-reduce per layer, call pair and
        # direction, on the launch stream)
        rngb = np.random.default_rng(321)
        Bb = 512
        xb1 = rngb.standard_normal((Bb, 40)).astype(np.float32)
        xb2 = (xb1 + 0.5 * rngb.standard_normal((Bb, 40))).astype(np.float32)
        yb = rngb.choice([1.0, -1.0], Bb)
        net = SiameseNetwork(output_path='/tmp/abn_rccl_bn', **kwb)
        net.load_state_dict({k[2:]: torch.from_numpy(v.copy()) for k, v in gb.items() if k.startswith('p.')})
        tr = TrainerSiamese(network=net, loss=L.coscos2(avg=False), optimizer_type='adadelta', lr=0.1,
                            dataloader=None, log_dir='/tmp/abn_runs_rccl', sync_batch_norm=True)
        if not grouped:
            net.bn_sync = parallel.BatchNormSync()          # the same launches, nobody to add the sums up with
        assert net.bn_sync.collective == grouped
        if grouped:                    # (on RCCL the exchange is issued from C: parallel.RcclBatchNormSync, no Python between the launches)
            res['bn_native'] = np.array(int(type(net.bn_sync).__name__ == 'RcclBatchNormSync'))
        net.train()
        batch = (torch.from_numpy(xb1).cuda(), torch.from_numpy(xb2).cuda(), torch.from_numpy(yb).cuda())
        res[tag + '.bn.losses'] = np.array([float(tr.train_step(batch, True)) for _ in range(3)])
        res[tag + '.bn.calls'] = np.array(net.bn_sync.calls)
        for k, v in net.state_dict().items():
            res[tag + '.bn.p.' + k] = v.detach().cpu().numpy()
        # (3) TrainerBuilder.train() on planned passes (batch plans + captured steps; under a group the all-reduce and
        # the optimizer stay outside the graph)
        np.random.seed(3000)
        torch.manual_seed(5)
        pl = OriginalDataLoader('unused', 'unused', batch_size=2)
        pl.set_data(feats, times, train, devp)
        net = SiameseNetwork(input_dim=40, num_hidden_layers=1, hidden_dim=64, output_dim=32, p_dropout=0.0,
                             activation_layer='sigmoid', output_path='/tmp/abn_rccl_planned')
        tr = TrainerSiamese(network=net, loss=L.coscos2(avg=False), optimizer_type='adadelta', lr=0.1, num_epochs=3,
                            patience=5, dataloader=pl, log_dir='/tmp/abn_runs_rccl')
        parallel.seed_all(0)           # (what a trainer under a process group does in its constructor: same start for both runs)
        tr.train()
        res[tag + '.planned.train_losses'] = np.array(tr.train_losses)
        res[tag + '.planned.graphs'] = np.array(sum(1 for v in getattr(tr, '_buckets', {}).values() if v['graph'] is not None))
        for k, p in net.named_parameters():
            res[tag + '.planned.p.' + k] = p.detach().cpu().numpy()
        # (4) TrainerBuilder.train() with sync_batch_norm=True on ragged word-pair batches (mostly < 256 rows: those
        # fall back to per-replica statistics, said once; the first pass runs train mode under no_grad; rank 0 pickles
        # the description with the BatchNormSync object on the network)
        np.random.seed(4000)
        torch.manual_seed(6)
        pl = OriginalDataLoader('unused', 'unused', batch_size=2)
        pl.set_data(feats, times, train, devp)
        net = SiameseNetwork(input_dim=40, num_hidden_layers=1, hidden_dim=64, output_dim=32, p_dropout=0.0,
                             activation_layer='sigmoid', batch_norm=True, output_path='/tmp/abn_rccl_bn_train')
        tr = TrainerSiamese(network=net, loss=L.coscos2(avg=False), optimizer_type='adadelta', lr=0.1, num_epochs=2,
                            patience=5, dataloader=pl, log_dir='/tmp/abn_runs_rccl', sync_batch_norm=True)
        if not grouped:
            net.bn_sync = parallel.BatchNormSync()
        parallel.seed_all(0)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter('always')
            tr.train()
        res[tag + '.bntrain.train_losses'] = np.array(tr.train_losses)
        res[tag + '.bntrain.warned'] = np.array(sum('per-replica statistics' in str(w.message) for w in caught))
        res[tag + '.bntrain.params_file'] = np.array(int(os.path.exists(net.output_path + '.params')))
        for k, v in net.state_dict().items():
            res[tag + '.bntrain.p.' + k] = v.detach().cpu().numpy()

    def graphed_bn():
        """(5) a step with cross-replica BatchNorm statistics captured into a hipGraph (on RCCL the per-layer exchange is a
        stream-ordered ncclAllReduce issued by the library: it is captured with the launches): three replays behind the
        capture's three warm-up steps against six eager steps of the same start."""
        rngb = np.random.default_rng(654)
        Bb = 512
        xb1 = rngb.standard_normal((Bb, 40)).astype(np.float32)
        xb2 = (xb1 + 0.5 * rngb.standard_normal((Bb, 40))).astype(np.float32)
        yb = rngb.choice([1.0, -1.0], Bb)
        batch = (torch.from_numpy(xb1).cuda(), torch.from_numpy(xb2).cuda(), torch.from_numpy(yb).cuda())
        out_ = {}
        for mode in ('eager', 'graph'):
            net = SiameseNetwork(output_path='/tmp/abn_rccl_bng_' + mode, **kwb)
            net.load_state_dict({k[2:]: torch.from_numpy(v.copy()) for k, v in gb.items() if k.startswith('p.')})
            tr = TrainerSiamese(network=net, loss=L.coscos2(avg=False), optimizer_type='adadelta', lr=0.1,
                                dataloader=None, log_dir='/tmp/abn_runs_rccl', sync_batch_norm=True)
            assert type(net.bn_sync).__name__ == 'RcclBatchNormSync'
            net.train()
            if mode == 'eager':
                losses = [float(tr.train_step(batch, True)) for _ in range(6)][3:]
            else:
                step = tr.make_graphed_step(batch, warmup=3)
                losses = [float(step(batch)) for _ in range(3)]
            out_[mode] = (np.array(losses), {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()})
        res['graphbn.eager_losses'], res['graphbn.replay_losses'] = out_['eager'][0], out_['graph'][0]
        worst = 0.0
        for k, v in out_['eager'][1].items():
            if k.endswith('num_batches_tracked'):
                assert int(v) == int(out_['graph'][1][k]), k
                continue
            worst = max(worst, float(np.abs(out_['graph'][1][k] - v).max() / max(np.abs(v).max(), 0.05)))
        res['graphbn.worst_param_diff'] = np.array(worst)

    def forced_fallback():
        """(6) RCCL that cannot be loaded (ABN_RCCL_LIB names a file that is not there): every rank agrees and the exchange
        falls back to the Python callback -- a warning, not an exception out of the trainer's constructor."""
        os.environ['ABN_RCCL_LIB'] = '/nonexistent/librccl.so'
        try:
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter('always')
                bs = parallel.bn_sync()
            res['fallback.type'] = np.array(type(bs).__name__)
            res['fallback.warned'] = np.array(sum('falls back to the Python callback' in str(w.message) for w in caught))
        finally:
            del os.environ['ABN_RCCL_LIB']

    run('plain')
    assert not parallel.active()
    parallel.init_from_env('nccl')
    assert parallel.active() and torch.distributed.get_backend() == 'nccl' and torch.distributed.get_world_size() == 1
    run('rccl')
    graphed_bn()
    forced_fallback()
    # bench.py's `collective` leg on the real backend (world of one): the communicator of its own that it counts the ranks
    # of, torch.distributed's all-reduce timed on the step's bucket; the one-shot exchange needs a peer and says so
    import bench
    net = SiameseNetwork(output_path='/tmp/abn_rccl_cb', **kw)
    tr = TrainerSiamese(network=net, loss=L.coscos2(avg=False), optimizer_type='adadelta', lr=0.1, dataloader=None,
                        log_dir='/tmp/abnet3_rccl_runs')
    co = bench.collective_bench(torch, tr, net, 1, reps=10)
    res['collective.backend'] = np.array(str(co['backend']))
    res['collective.rank_count'] = np.array(str(co['rank_count_seen_by_rccl']))
    res['collective.allreduce_us'] = np.array(str(co['allreduce_us']))
    res['backend'] = np.array(torch.distributed.get_backend())
    np.savez(out + '.rccl.npz', **res)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    print('rccl worker done', flush=True)


if __name__ == '__main__':
    main()
