"""One data-parallel rank of tests/test_gpu_dp.py: started as a FRESH process (by
conftest.py, before the pytest process has touched the GPU), ranks share GPU 0 and
talk over gloo -- the code path of an 8-GPU RCCL job with another transport.

    python tests/dp_worker.py RANK WORLD PORT OUT_PREFIX
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    import ast
    import numpy as np
    import torch
    from abnet3_amd import parallel
    import abnet3_amd.loss as L
    from abnet3_amd.model import SiameseNetwork
    from abnet3_amd.trainer import TrainerSiamese
    from abnet3_amd.dataloader import FramesDataLoader, OriginalDataLoader
    from conftest import load_golden
    parallel.init_from_env('gloo')
    res = {}

    # --- (1) the HIP trainer under a process group: each rank steps on its half of a
    # 2B batch; the test compares with one process on the whole batch
    g = load_golden('train_mid_bn0.npz')
    kw = ast.literal_eval(str(g['kw']))
    rng = np.random.default_rng(123)
    B2 = 192
    x1 = rng.standard_normal((B2, 40)).astype(np.float32)
    x2 = rng.standard_normal((B2, 40)).astype(np.float32)
    y = rng.choice([1.0, -1.0], B2)
    half = B2 // world
    sl = slice(rank * half, rank * half + half)
    for avg in (False, True):
        for oname, lr in (('adadelta', 0.1), ('sgd', 0.01)):
            net = SiameseNetwork(output_path='/tmp/abn_dp_%d' % rank, **kw)
            sd = {k[2:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith('p.')}
            if rank != 0:          # only rank 0 holds the real weights: the trainer must broadcast them
                sd = {k: torch.zeros_like(v) for k, v in sd.items()}
            net.load_state_dict(sd)
            tr = TrainerSiamese(network=net, loss=L.coscos2(avg=avg), optimizer_type=oname, lr=lr,
                                dataloader=None, log_dir='/tmp/abn_runs_dp')
            assert tr.world_size == world
            net.train()
            batch = (torch.from_numpy(x1[sl]).cuda(), torch.from_numpy(x2[sl]).cuda(), torch.from_numpy(y[sl]).cuda())
            losses = [float(tr.train_step(batch, True)) for _ in range(3)]
            tag = 'avg%d.%s' % (int(avg), oname)
            res[tag + '.losses'] = np.array(losses)
            for k, p in net.named_parameters():
                res[tag + '.p.' + k] = p.detach().cpu().numpy()

    # --- (1b) the same with 1600 pairs per rank: the single-launch chains instead of the layer-per-launch kernels, the
    # backward in two calls with the upper layers' all-reduce in flight under the second (abn_tower_desc.wgrad_part)
    rngc = np.random.default_rng(77)
    Bc = 3200
    xc1 = rngc.standard_normal((Bc, 40)).astype(np.float32)
    xc2 = (xc1 + 0.3 * rngc.standard_normal((Bc, 40))).astype(np.float32)
    yc = rngc.choice([1.0, -1.0], Bc)
    hc = Bc // world
    slc = slice(rank * hc, rank * hc + hc)
    for overlap in (True, False):
        net = SiameseNetwork(output_path='/tmp/abn_dp_c_%d' % rank, **kw)
        net.load_state_dict({k[2:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith('p.')})
        tr = TrainerSiamese(network=net, loss=L.coscos2(avg=False), optimizer_type='adadelta', lr=0.1,
                            dataloader=None, log_dir='/tmp/abn_runs_dp')
        tr.overlap_allreduce = overlap
        net.train()
        batch = (torch.from_numpy(xc1[slc]).cuda(), torch.from_numpy(xc2[slc]).cuda(), torch.from_numpy(yc[slc]).cuda())
        losses = [float(tr.train_step(batch, True)) for _ in range(3)]
        tag = 'chain.overlap%d' % int(overlap)
        res[tag + '.losses'] = np.array(losses)
        for k, p in net.named_parameters():
            res[tag + '.p.' + k] = p.detach().cpu().numpy()

    # --- (2) loaders shard themselves: disjoint, complete, one shared order
    gl = load_golden('frames_loader.npz')
    feats = {k[5:]: v for k, v in gl.items() if k.startswith('feat.')}
    times = {k: np.arange(len(v)) * 0.01 + 0.0025 for k, v in feats.items()}

    def parse(line):
        t = str(line).split(' ')
        return (t[0], float(t[1]), float(t[2]), t[3], float(t[4]), float(t[5]), t[6])
    train = [parse(l) for l in gl['train_pairs']]
    devp = [parse(l) for l in gl['dev_pairs']]
    np.random.seed(1000 + rank)            # ranks start from DIFFERENT RNG states on purpose
    dl = FramesDataLoader('unused', 'unused', batch_size=25)
    dl.set_data(feats, times, train, devp)
    for ep, mode in enumerate('TTD'):
        ys, xs = [], []
        for a, b, c in dl.batch_iterator(train_mode=(mode == 'T')):
            xs.append(np.concatenate([a.cpu().numpy(), b.cpu().numpy()], axis=1))
            ys.append(c.cpu().numpy())
        res['frames.ep%d.x' % ep] = np.stack(xs) if xs else np.zeros((0, 25, 80), np.float32)
        res['frames.ep%d.y' % ep] = np.stack(ys) if ys else np.zeros((0, 25), np.int64)
    np.random.seed(2000 + rank)
    ol = OriginalDataLoader('unused', 'unused', batch_size=2)
    ol.set_data(feats, times, train, devp)
    for ep, mode in enumerate('TD'):
        sizes, firsts = [], []
        for a, b, c in ol.batch_iterator(train_mode=(mode == 'T')):
            sizes.append(len(c))
            firsts.append(a[0].cpu().numpy())
        res['words.ep%d.sizes' % ep] = np.array(sizes)
        res['words.ep%d.first' % ep] = np.stack(firsts) if firsts else np.zeros((0, 40), np.float32)
    # --- (3) planned passes (batch plans + captured steps) under the process group
    np.random.seed(3000 + rank)
    torch.manual_seed(5)
    pl = OriginalDataLoader('unused', 'unused', batch_size=2)
    pl.set_data(feats, times, train, devp)
    net = SiameseNetwork(input_dim=40, num_hidden_layers=1, hidden_dim=64, output_dim=32, p_dropout=0.0, activation_layer='sigmoid',
                         output_path='/tmp/abn_dp_planned_%d' % rank)
    tr = TrainerSiamese(network=net, loss=L.coscos2(avg=False), optimizer_type='adadelta', lr=0.1, num_epochs=3, patience=5,
                        dataloader=pl, log_dir='/tmp/abn_runs_dp')
    tr.train()
    res['planned.train_losses'] = np.array(tr.train_losses)
    res['planned.graphs'] = np.array(sum(1 for v in getattr(tr, '_buckets', {}).values() if v['graph'] is not None))
    for k, p in net.named_parameters():
        res['planned.p.' + k] = p.detach().cpu().numpy()
    # --- (4) BatchNorm with cross-replica statistics (TrainerBuilder(sync_batch_norm=True)): each rank steps on its
    # half of a 2B batch; the test compares with one process on the whole batch
    gb = load_golden('train_mid_bn1.npz')
    kwb = ast.literal_eval(str(gb['kw']))
    rngb = np.random.default_rng(321)
    Bb = 1024
    xb1 = rngb.standard_normal((Bb, 40)).astype(np.float32)
    xb2 = (xb1 + 0.5 * rngb.standard_normal((Bb, 40))).astype(np.float32)
    yb = rngb.choice([1.0, -1.0], Bb)
    hb = Bb // world
    slb = slice(rank * hb, rank * hb + hb)
    for sync in (True, False):
        net = SiameseNetwork(output_path='/tmp/abn_dp_bn_%d' % rank, **kwb)
        net.load_state_dict({k[2:]: torch.from_numpy(v.copy()) for k, v in gb.items() if k.startswith('p.')})
        tr = TrainerSiamese(network=net, loss=L.coscos2(avg=False), optimizer_type='adadelta', lr=0.1,
                            dataloader=None, log_dir='/tmp/abn_runs_dp', sync_batch_norm=sync)
        net.train()
        batch = (torch.from_numpy(xb1[slb]).cuda(), torch.from_numpy(xb2[slb]).cuda(), torch.from_numpy(yb[slb]).cuda())
        losses = [float(tr.train_step(batch, True)) for _ in range(3)]
        tag = 'bn.sync%d' % int(sync)
        res[tag + '.losses'] = np.array(losses)
        res[tag + '.calls'] = np.array(net.bn_sync.calls if sync else 0)
        for k, v in net.state_dict().items():
            res[tag + '.p.' + k] = v.detach().cpu().numpy()
    # --- (5) ... with batches of DIFFERENT sizes on the ranks (OriginalDataLoader's ragged word-pair batches): 384 + 640 pairs.
    # The statistics divide by the all-reduced row count (it travels with the sums): one process on the 1024.
    cut = [0, 384, 1024]
    slu = slice(cut[rank], cut[rank + 1])
    net = SiameseNetwork(output_path='/tmp/abn_dp_bnu_%d' % rank, **kwb)
    net.load_state_dict({k[2:]: torch.from_numpy(v.copy()) for k, v in gb.items() if k.startswith('p.')})
    tr = TrainerSiamese(network=net, loss=L.coscos2(avg=False), optimizer_type='adadelta', lr=0.1,
                        dataloader=None, log_dir='/tmp/abn_runs_dp', sync_batch_norm=True)
    net.train()
    batch = (torch.from_numpy(xb1[slu]).cuda(), torch.from_numpy(xb2[slu]).cuda(), torch.from_numpy(yb[slu]).cuda())
    res['bn.uneven.losses'] = np.array([float(tr.train_step(batch, True)) for _ in range(3)])
    for k, v in net.state_dict().items():
        res['bn.uneven.p.' + k] = v.detach().cpu().numpy()
    # --- (6) ... and with one rank's batch too small for the launches that carry the sums (100 pairs = 200 rows on rank 0,
    # 512 pairs on rank 1): the ranks agree (one MIN all-reduce per forward) and BOTH use per-replica statistics -- no rank
    # waits in an exchange the other never enters
    import warnings
    nsm = 100 if rank == 0 else 512
    net = SiameseNetwork(output_path='/tmp/abn_dp_bns_%d' % rank, **kwb)
    net.load_state_dict({k[2:]: torch.from_numpy(v.copy()) for k, v in gb.items() if k.startswith('p.')})
    tr = TrainerSiamese(network=net, loss=L.coscos2(avg=False), optimizer_type='adadelta', lr=0.1,
                        dataloader=None, log_dir='/tmp/abn_runs_dp', sync_batch_norm=True)
    net.train()
    batch = (torch.from_numpy(xb1[:nsm]).cuda(), torch.from_numpy(xb2[:nsm]).cuda(), torch.from_numpy(yb[:nsm]).cuda())
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter('always')
        res['bn.small.losses'] = np.array([float(tr.train_step(batch, True)) for _ in range(2)])
    res['bn.small.calls'] = np.array(net.bn_sync.calls)
    res['bn.small.warned'] = np.array(sum('per-replica statistics' in str(w.message) for w in caught))
    # --- (7) the one-shot all-reduce (abn_allreduce_oneshot: peer-mapped mailboxes over hipIpcMemHandle, one launch per rank)
    # beside torch.distributed's: random buckets of several sizes, repeated calls (the tags move on), then three trainer steps
    # with ABN_ONESHOT_ALLREDUCE=1 against the same steps over gloo
    try:
        one = parallel.OneShotAllReduce(600000)
        res['oneshot.available'] = np.array(1)
    except (RuntimeError, OSError, AttributeError) as e:
        one = None
        res['oneshot.available'] = np.array(0)
        res['oneshot.why'] = np.array(str(e))
    if one is not None:
        worst, same = 0.0, 1
        rngo = np.random.default_rng(900 + rank)
        for n in (4, 64, 4096, 571712, 1000, 571712):
            v = torch.from_numpy(rngo.standard_normal(n).astype(np.float32)).cuda()
            ref = v.clone()
            torch.distributed.all_reduce(ref)
            one.all_reduce(v)
            torch.cuda.synchronize()
            worst = max(worst, float((v - ref).abs().max() / ref.abs().max()))
            both = [torch.empty_like(v) for _ in range(world)]
            torch.distributed.all_gather(both, v)
            same = same and all(bool(torch.equal(both[0], b)) for b in both[1:])
        res['oneshot.worst_rel'] = np.array(worst)
        res['oneshot.replicas_identical'] = np.array(int(same))
        res['oneshot.calls'] = np.array(one.calls)
        # (what a call of the C2 bucket costs between two processes sharing this GPU: the launches' and the two hand-overs'
        # latency, not a wire time -- DESIGN.md section 4 quotes it as such)
        v = torch.zeros(571712, device='cuda')
        torch.distributed.barrier()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            one.all_reduce(v)
        e1.record()
        torch.cuda.synchronize()
        res['oneshot.us_per_call'] = np.array(e0.elapsed_time(e1) * 1e3 / 50)
        one.close()
        runs = {}
        for flag in ('1', '0'):
            os.environ['ABN_ONESHOT_ALLREDUCE'] = flag
            net = SiameseNetwork(output_path='/tmp/abn_dp_one_%d' % rank, **kw)
            net.load_state_dict({k[2:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith('p.')})
            tr = TrainerSiamese(network=net, loss=L.coscos2(avg=False), optimizer_type='adadelta', lr=0.1,
                                dataloader=None, log_dir='/tmp/abn_runs_dp')
            assert (tr.oneshot is not None) == (flag == '1')
            net.train()
            batch = (torch.from_numpy(x1[sl]).cuda(), torch.from_numpy(x2[sl]).cuda(), torch.from_numpy(y[sl]).cuda())
            losses = [float(tr.train_step(batch, True)) for _ in range(3)]
            runs[flag] = (losses, {k: p.detach().cpu().numpy() for k, p in net.named_parameters()})
        del os.environ['ABN_ONESHOT_ALLREDUCE']
        res['oneshot.train_losses_diff'] = np.array(max(abs(a - b) / abs(b) for a, b in zip(runs['1'][0], runs['0'][0])))
        res['oneshot.train_param_diff'] = np.array(max(float(np.abs(runs['1'][1][k] - v).max() / max(np.abs(v).max(), 1e-6)) for k, v in runs['0'][1].items()))
    np.savez(out + '.rank%d.npz' % rank, **res)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    print('dp worker %d done' % rank, flush=True)


if __name__ == '__main__':
    main()
