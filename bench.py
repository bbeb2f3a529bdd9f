#!/usr/bin/env python3
"""Headline benchmark: frame-pairs/s of one Siamese train step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[1] / SURVEY.md section 8 "C2"): SiameseNetwork
40 -> 500 x2 -> 100 (input_dim=40, num_hidden_layers=2, hidden_dim=500,
output_dim=100, sigmoid, no BN, dropout 0), coscos2(avg=False), Adadelta(lr=0.1),
4096 synthetic frame pairs per GPU per step.  Tower arithmetic: the package default,
"f16x2" -- every fp32 operand scaled by a power of two and split into two fp16 terms
(22 significant bits), three fp16 MFMA products per operand pair, fp32 accumulation:
fp32-grade results (every golden parity test passes at the 1e-5 bar, as far from a
float64 evaluation as the reference's own fp32) on the 16-bit matrix cores
BASELINE.json names; the bf16 x 3 split (six bf16 products, round 2's headline), the
exact-fp32 MFMA mode and the plain bf16 mode (outside the parity bar) of the same step
are timed beside it.  One step = what TrainerSiamese.train_step runs:
forward of both towers, pair loss, backward, [RCCL all-reduce of the flat gradient
bucket], optimizer.  Inputs are resident in HBM before the timed region.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     fp32-MFMA roofline of the dominant kernel, timed live with
               HIP events on the launch stream
  cpu_baseline the oracle's torch-CPU restatement of the same step timed on
               this host's cores (N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

C2 = dict(input_dim=40, num_hidden_layers=2, hidden_dim=500, output_dim=100,
          p_dropout=0.0, batch_norm=False, type_init='xavier_uni',
          activation_layer='sigmoid')
BATCH = 4096
POOL = 64                          # a fresh batch per step from a pre-generated pool of 64
SIDE_STEPS = 50                    # fewest timed steps of a side mode (the other arithmetics, the variants) whatever --steps is
FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA
# bf16 x 3 spends six bf16 MFMA products per algorithmic fp32 product: its matrix-core roof,
# in algorithmic FLOP/s, is the bf16 peak / 6
X3_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0
# fp16 x 2 spends three fp16 MFMA products (the F16 forms take the bf16 forms' cycles): peak / 3
F16X2_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 3.0
PLANES = {'bf16x3': 3, 'f16x2': 2, 'bf16': 1}
PREC_CODE = {'fp32': 0, 'bf16': 1, 'bf16x3': 2, 'f16x2': 3}
# MACs per tower row: fwd 570 000, wgrad 570 000, dgrad 550 000
FLOP_PER_PAIR = 2 * 2 * (570000 + 570000 + 550000)


def make_pool(seed, device):
    import numpy as np
    import torch
    g = torch.Generator().manual_seed(seed)
    pool = []
    for i in range(POOL):
        x1 = torch.randn(BATCH, 40, generator=g)
        x2 = torch.randn(BATCH, 40, generator=g)
        y = torch.from_numpy(np.random.default_rng(seed * 100 + i).choice([1, -1], BATCH))
        pool.append((x1.to(device), x2.to(device), y.to(device)))
    return pool


def kernel_names(prec):
    """(forward kernel, dominant backward kernel) of the step in `prec`."""
    if prec == 'fp32':
        return ('void abn::tower_fwd_fused_kernel<0>(abn::FusedFwdP)',
                'void abn::gemm_bwd_pair_kernel<128, 64, 0>(abn::GemmP, int, abn::GemmP)')
    planes = PLANES[prec]
    # (fp16 x 2 from 2048 tower rows on -- the bench's 8192 -- : every layer on 128 x 128 tiles, the shape's own kernel)
    return ('void abn::tower_fwd_planes_kernel<%d, 0>(abn::PlanesFwdP)' % planes,
            'abn::wgrad_planes128_kernel(abn::WgradP)' if prec == 'f16x2' else 'void abn::wgrad_planes_kernel<%d>(abn::WgradP)' % planes)


SETTLE_SCALE = 1.0      # --trace-run sets 0: the side legs' settling launches would drown the step's own in a kernel trace


def _time_launches(torch, fn, reps):
    """Average GPU time of one fn() (a fixed launch sequence): `reps` calls are
    captured into ONE hipGraph, so the HIP events around its replays (recorded on
    the stream the graph -- and in the step, the C-ABI calls -- launch on) see the
    kernels back to back, as the rocprofv3 kernel trace does, not the host's
    launch gaps."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(graph):
        for _ in range(reps):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    # (the capture above left the GPU idle: replays until the clock has settled, as everywhere in this file -- settle())
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2 * SETTLE_SCALE:
        for _ in range(4):
            graph.replay()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / (8 * reps)


def _traffic(kernel):
    """HBM bytes per launch from the TCC counters (FETCH_SIZE x2 per the gfx950
    correction + WRITE_SIZE), collected by tools/collect_profiles.sh in separate
    --pmc passes and committed under profiles/ (the newest round's file that names the kernel)."""
    for rnd in ('r06', 'r05', 'r04', 'r03', 'r02', 'r01'):
        try:
            t = json.load(open(os.path.join(ROOT, 'profiles', '%s_pmc_traffic.json' % rnd)))
            for name, v in t.items():
                if kernel in name:
                    return round(v['hbm_bytes_per_launch'])
        except Exception:
            pass
    return None


def _trace_us(kernel):
    """(average duration in us of `kernel` inside the step, file) from the newest committed rocprofv3 --kernel-trace --stats
    summary of this command (profiles/rNN_bench_kernel_stats.csv): which launch is dominant IN THE STEP is read there --
    a kernel re-run alone, back to back, sees another cache than between its neighbours."""
    import csv
    for rnd in ('r06', 'r05', 'r04', 'r03', 'r02'):
        path = os.path.join(ROOT, 'profiles', '%s_bench_kernel_stats.csv' % rnd)
        try:
            for row in csv.DictReader(open(path)):
                if kernel in row['Name'] and int(row['Calls']) >= 500:         # (the timed loop's launches, not a side mode's)
                    return round(float(row['AverageNs']) * 1e-3, 2), 'profiles/%s_bench_kernel_stats.csv' % rnd
        except Exception:
            pass
    return None, None


def _trace_binary(trace_file):
    """Is the committed kernel trace of the library build this run has loaded?  (a stale trace may name another
    dominant launch than the binary being timed.)  profiles/rNN_lib_digest.txt is written when the trace is collected."""
    if not trace_file:
        return None
    try:
        import hashlib
        from abnet3_amd import build as _b
        want = open(os.path.join(ROOT, trace_file.replace('_bench_kernel_stats.csv', '_lib_digest.txt'))).read().strip()
        h = hashlib.sha1()
        with open(os.environ.get('ABNET3_HIP_LIB') or _b.LIB, 'rb') as f:
            for chunk in iter(lambda: f.read(1 << 20), b''):
                h.update(chunk)
        return 'the library build this run loaded' if h.hexdigest() == want else \
            'ANOTHER build of the library (%s..): in_step_trace_us and `dominant` may be stale' % want[:10]
    except OSError:
        return 'unknown (no digest beside the trace)'


def planes_roofline(torch, net, reps=20):
    """precision f16x2 / bf16x3 / bf16 (csrc/tower_planes.h): the step is three launches of similar length -- the
    forward chain, the data-gradient chain, the weight gradients -- plus the fused reduction + optimizer.
    The line's top level is whichever of the three took LONGEST in this run, the other two follow under their
    names; `whole_step` (added by main) is the timed loop's own figure against the same roof.  wgrad_planes_kernel: every
    layer's dW = dZ^T [A | 1] over the 2 x 4096 rows, split over the rows into slabs, one launch (its tiles
    placed so that an XCD's L2 serves the re-reads).  Each of the three is timed live, alone: `reps`
    launches captured into one hipGraph and bracketed by HIP events on the launch stream (the backward
    kernels through abn_tower_backward_launch, the header's measurement entry that issues one of the backward's two
    launches; the forward with its packed weight image still valid, i.e. without the ~6.6 us
    pack_planes_scaled_kernel a step's forward starts with); `forward_backward_sequence_us` is the three in the
    step's order.  Algorithmic FLOPs: weight gradients 2 * 8192 * sum_l N_l (K_l + 1), the chains
    2 * 8192 * sum_l N_l K_l (the data-gradient chain without the first layer).
    Roof: the dense 16-bit MFMA peak divided by the MFMA products each algorithmic product costs (three fp16
    products for f16x2: 2500 / 3 = 833.3 TFLOP/s algorithmic; six for bf16x3: 416.7; one for bf16).  Both peaks assume the 2.4 GHz
    boost clock; under this workload the chip holds ~1.6-1.8 GHz (s_memtime against wall clock,
    tools/probes/mfma_peak.hip).  `hbm` gives the dominant launch's measured TCC traffic over its time."""
    prec = net.precision
    peak = {'bf16x3': X3_PEAK_TFLOPS, 'f16x2': F16X2_PEAK_TFLOPS, 'bf16': BF16_MFMA_PEAK_TFLOPS}[prec]
    fwd_name, wgrad_name = kernel_names(prec)
    dims = [40, 500, 500, 500, 100]
    rows = 2 * BATCH
    x12 = torch.randn(rows, 40, device='cuda')
    net.train()
    emb, state = net.direct_forward(x12[:BATCH].contiguous(), x12[BATCH:].contiguous())
    d_out = torch.randn_like(emb) * 1e-3

    x1, x2 = x12[:BATCH].contiguous(), x12[BATCH:].contiguous()

    def bwd():
        net.direct_backward(state, d_out, d_out_is_dz=True, defer_reduce=True)
    bwd()                                           # a complete backward: both kernels' outputs are in place

    # Each launch alone: `reps` launches captured into one hipGraph, HIP events around its replays.  (Against the
    # in-step durations of the committed rocprofv3 trace -- `in_step_trace_us` beside each entry -- this reads the
    # weight gradients a few % high or low and the two chains up to 15 % high: a step moves ~400 MB through HBM /
    # Infinity Cache, and twenty launches of one kernel rewriting the same images see a different cache.  Events
    # BETWEEN the launches of a real sequence add ~10 us each, and sequences with one launch left out do not subtract
    # cleanly: both were tried.)
    times = {}
    from abnet3_amd import _lib
    lib = _lib.load()
    part_fn = lib.abn_tower_backward_launch       # the header's measurement entry: one of the backward's two launches
    part_fn.restype = _lib.C.c_int
    seg, sv, gp = state
    gbuf, _ = gp.views(seg)
    desc = seg.descriptor(with_grads=True, grad_buf=gbuf, masks=sv.masks, d_out_is_dz=True, defer_reduce=True)
    sc_n = lib.abn_tower_bwd_scratch_floats(_lib.C.byref(desc), rows)
    pending = net._pending_reduce                     # (desc, rows, scratch, ...) of the complete backward above
    scratch = pending[2]

    def part(which):
        _lib.check(part_fn(_lib.C.byref(desc), _lib.ptr(sv.x1), _lib.ptr(sv.x2), _lib.ptr(d_out), _lib.C.c_int64(rows),
                           _lib.C.c_int64(2), _lib.ptr(sv.ws), _lib.ptr(scratch), _lib.C.c_int64(sc_n), which, _lib.stream()),
                   'abn_tower_backward_launch')
    times['dgrad'] = _time_launches(torch, lambda: part(1), reps)
    times['wgrad'] = _time_launches(torch, lambda: part(2), reps)
    times['both'] = _time_launches(torch, bwd, reps)
    times['forward'] = _time_launches(torch, lambda: net.direct_forward(x1, x2), reps)

    def fwd_bwd():
        net.direct_forward(x1, x2)
        bwd()
    times['sequence'] = _time_launches(torch, fwd_bwd, reps)
    net.take_pending_reduce()
    fl_w = 2.0 * rows * sum(dims[l + 1] * (dims[l] + 1) for l in range(4))
    fl_d = 2.0 * rows * sum(dims[l + 1] * dims[l] for l in range(1, 4))
    fl_f = 2.0 * rows * sum(dims[l + 1] * dims[l] for l in range(4))
    planes = PLANES[prec]
    launches = {
        'weight_gradients': (wgrad_name, 'wgrad_planes128_kernel' if prec == 'f16x2' else 'wgrad_planes_kernel<%d>' % planes, times['wgrad'], fl_w,
                             "all four layers' weight and bias gradients, one launch"),
        'dgrad_chain': ('void abn::tower_dgrad_planes_kernel<%d>(abn::PlanesBwdP)' % planes,
                        'tower_dgrad_planes_kernel<%d>' % planes, times['dgrad'], fl_d, 'dZ through the three upper layers, one launch'),
        'forward': (fwd_name, 'tower_fwd_planes_kernel<%d, 0>' % planes, times['forward'], fl_f,
                    'the whole training forward of both towers; the persistent weight image is valid across these '
                    'launches: no pack_planes_kernel in between'),
    }
    entries = {}
    for key, (name, short, t, fl, what) in launches.items():
        in_step, trace_file = _trace_us(short)
        entries[key] = {'kernel': '%s  (%s)' % (name, what),
                        'achieved': round(fl / t / 1e12, 2), 'frac': round(fl / t / 1e12 / peak, 4),
                        'avg_launch_us': round(t * 1e6, 2),
                        'avg_launch_us_is': 'this run, the launch ALONE: %d launches back to back from one hipGraph, HIP events on the launch stream' % reps,
                        'in_step_trace_us': in_step,
                        'in_step_trace_us_is': 'the same kernel between its neighbours in the step: committed rocprofv3 --kernel-trace --stats of this command (%s)' % trace_file,
                        'in_step_trace_is_of': _trace_binary(trace_file),
                        'flop_per_launch': fl, 'traffic': _traffic(short)}
    # What the chains are actually bound by: every workgroup of 32 rows pulls the whole packed layer through
    # its CU's L1 from the XCD's L2 -- bytes per launch = workgroups x packed weight image --, at most 64 B/clk per CU.
    steps = lambda c: ((c + 15) // 16 + 3) // 4 * 4
    blocks = lambda f: (f + 31) // 32
    img_f = sum(blocks(dims[l + 1]) * steps(dims[l]) for l in range(4)) * planes * 1024
    img_d = sum(blocks(dims[l]) * steps(dims[l + 1]) for l in range(1, 4)) * planes * 1024
    for key, img in (('forward', img_f), ('dgrad_chain', img_d)):
        tb = img * (rows // 32) / (entries[key]['avg_launch_us'] * 1e-6) / 1e12
        entries[key]['operand_stream'] = {'bytes_per_workgroup': img, 'workgroups': rows // 32, 'achieved_TB_s_from_L2': round(tb, 2),
                                          'peak_TB_s': round(256 * 64 * 2.4e9 / 1e12, 1),
                                          'note': 'packed weights streamed L2 -> L1 by every workgroup; peak = 256 CUs x 64 B/clk x 2.4 GHz '
                                                  '(the k-loops run at 45-58 B/clk per CU; the rest of a launch is epilogue without loads)'}
    # The line's top level names the launch that is dominant IN THE STEP: the order is the committed rocprofv3 trace's
    # (in_step_trace_us), its duration and fraction are this run's live measurement of that launch alone (avg_launch_us;
    # against the trace the standalone harness reads the chains up to 15 % high: the top-level fraction is the
    # conservative one); without a committed trace the longest standalone launch.  The other two follow under their names.
    if all(entries[k]['in_step_trace_us'] for k in entries):
        dominant = max(entries, key=lambda k: entries[k]['in_step_trace_us'])
        dominant_by = 'in_step_trace_us (committed rocprofv3 trace)'
    else:
        dominant = max(entries, key=lambda k: entries[k]['avg_launch_us'])
        dominant_by = 'avg_launch_us (no committed trace found)'
    e = entries[dominant]
    out = {'bound': 'mfma', 'achieved': e['achieved'], 'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': e['frac'],
           'traffic': e['traffic'], 'arithmetic': prec,
           'peak_note': {'bf16x3': 'dense bf16 MFMA 2500 TFLOP/s / 6 bf16 products per algorithmic product',
                         'f16x2': 'dense fp16 MFMA 2500 TFLOP/s / 3 fp16 products per algorithmic product',
                         'bf16': 'dense bf16 MFMA'}[prec],
           # the same launch priced both ways, so that rounds stay comparable: `frac` = executed-MFMA utilisation
           # (algorithmic FLOP/s over the dense 16-bit peak divided by the MFMA products one algorithmic product
           # costs in this arithmetic); `frac_of_dense_16bit_peak` = algorithmic FLOP/s over the raw 2500 TFLOP/s
           'frac_of_dense_16bit_peak': round(e['achieved'] / BF16_MFMA_PEAK_TFLOPS, 4),
           'frac_of_fp32_mfma_peak': round(e['achieved'] / FP32_MFMA_PEAK_TFLOPS, 4),
           'frac_of_bf16x3_peak': round(e['achieved'] / X3_PEAK_TFLOPS, 4),
           'dominant': dominant, 'dominant_by': dominant_by,
           'kernel': e['kernel'], 'avg_launch_us': e['avg_launch_us'], 'avg_launch_us_is': e['avg_launch_us_is'],
           'in_step_trace_us': e['in_step_trace_us'], 'in_step_trace_us_is': e['in_step_trace_us_is'],
           'in_step_trace_is_of': e['in_step_trace_is_of'], 'flop_per_launch': e['flop_per_launch']}
    if e['in_step_trace_us']:
        out['frac_at_in_step_duration'] = round(e['flop_per_launch'] / (e['in_step_trace_us'] * 1e-6) / 1e12 / peak, 4)
    if 'operand_stream' in e:
        out['operand_stream'] = e['operand_stream']
    if e['traffic']:
        gbs = e['traffic'] / (e['avg_launch_us'] * 1e-6) / 1e9
        out['hbm'] = {'achieved': round(gbs, 1), 'peak': 8000.0, 'unit': 'GB/s', 'frac': round(gbs / 8000.0, 4),
                      'note': 'measured TCC traffic per launch (profiles/rNN_pmc_traffic.json, the newest: FETCH_SIZE x 2 + WRITE_SIZE, '
                              'Infinity-Cache hits included) / launch time'}
    for key, v in entries.items():
        if key != dominant:
            out[key] = v
    out['backward_sequence_us'] = round(times['both'] * 1e6, 2)
    out['forward_backward_sequence_us'] = round(times['sequence'] * 1e6, 2)
    return out


def tower_roofline(torch, net, reps=20):
    """(precision fp32; planes_roofline covers bf16x3 / bf16.)
    The dominant kernel of the step by time: gemm_bwd_pair_kernel<128, 64, .> -- the wgrad and
    the dgrad of a 500x500 layer over the 2 x 4096 tower rows in ONE grid, two launches per step.
    Timed live in the network's arithmetic through abn_linear_backward_prec with dW = NULL (the
    single-layer entry that issues exactly the grid the tower backward issues, without the
    layer's slab reduction), `reps` launches captured into one hipGraph and bracketed by HIP
    events on the launch stream.  Algorithmic FLOPs per launch: 2 GEMMs x 2 * 8192 * 500 * 500 (+ the bias column).
    Roof: the matrix cores in the arithmetic the kernel uses -- fp32 MFMA 157.3 TFLOP/s for
    'fp32'; for 'bf16x3' the dense bf16 MFMA peak divided by the six bf16 products each
    algorithmic product costs (2500 / 6 = 416.7 TFLOP/s algorithmic).  The whole-forward kernel
    and the whole backward sequence follow as further entries."""
    from abnet3_amd import _lib
    lib = _lib.load()
    prec = net.precision
    peak = {'fp32': FP32_MFMA_PEAK_TFLOPS, 'bf16x3': X3_PEAK_TFLOPS, 'bf16': BF16_MFMA_PEAK_TFLOPS}[prec]
    fused_name, pair_name = kernel_names(prec)
    rows, k, n = 2 * BATCH, 500, 500
    dz, a = torch.randn(rows, n, device='cuda'), torch.rand(rows, k, device='cuda')
    W = torch.randn(n, k, device='cuda') * 0.05
    dW, db, dx = torch.empty(n, k, device='cuda'), torch.empty(n, device='cuda'), torch.empty(rows, k, device='cuda')
    sc_n = lib.abn_linear_wgrad_scratch_floats(rows, k, n)
    sc = torch.empty(sc_n, device='cuda')

    def pair():
        _lib.check(lib.abn_linear_backward_prec(_lib.ptr(dz), _lib.ptr(W), _lib.ptr(a), rows, k, n, _lib.ACT['sigmoid'],
                                                PREC_CODE[prec], None, None, _lib.ptr(dx), _lib.ptr(sc),
                                                sc_n, _lib.stream()), 'abn_linear_backward_prec')
    t_pair = _time_launches(torch, pair, reps)
    # what a back-to-back launch costs on top of the kernel itself in this harness: the same
    # measurement with a kernel that does (almost) nothing
    tiny_idx = torch.zeros(1, dtype=torch.int64, device='cuda')
    tiny_out = torch.empty(1, k, device='cuda')

    def tiny():
        _lib.check(lib.abn_gather_rows(_lib.ptr(a), _lib.ptr(tiny_idx), 1, k, _lib.ptr(tiny_out), _lib.stream()), 'gather')
    t_gap = _time_launches(torch, tiny, reps)
    flop = 2.0 * rows * k * n + 2.0 * rows * (k + 1) * n
    achieved = flop / t_pair / 1e12
    out = {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': round(peak, 1),
           'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4),
           'traffic': _traffic(pair_name.split('(')[0].replace('void ', '')),
           'arithmetic': prec,
           'peak_note': {'fp32': 'fp32 MFMA (v_mfma_f32_32x32x2_f32)',
                         'bf16x3': 'dense bf16 MFMA 2500 TFLOP/s / 6 bf16 products per algorithmic product',
                         'bf16': 'dense bf16 MFMA'}[prec],
           'frac_of_fp32_mfma_peak': round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
           'kernel': pair_name + '  (wgrad + dgrad of a 500x500 layer in one grid: that grid alone, launched '
                     'back to back from one hipGraph; avg_launch_us includes the ~2 us dispatch gap between launches)',
           'avg_launch_us': round(t_pair * 1e6, 2), 'flop_per_launch': flop,
           'empty_launch_us_same_harness': round(t_gap * 1e6, 2)}

    # the whole forward of both towers, one launch (the largest single launch of the step)
    x12 = torch.randn(rows, 40, device='cuda')
    net.train()

    def fwd():
        with torch.no_grad():
            net.forward_pair_rows(x12)
    t = _time_launches(torch, fwd, reps)
    fl = 2.0 * rows * (40 * 500 + 2 * 500 * 500 + 500 * 100)
    out['forward'] = {'kernel': fused_name + '  (whole forward of both towers in one launch)',
                      'achieved': round(fl / t / 1e12, 2), 'frac': round(fl / t / 1e12 / peak, 4),
                      'avg_launch_us': round(t * 1e6, 2), 'flop_per_launch': fl,
                      'traffic': _traffic('tower_fwd_fused_kernel')}
    # the backward of the same step: three (wgrad + dgrad) grids, the input layer's wgrad, the slab
    # reduction -- timed as one sequence through abn_tower_backward (dz of the output layer given)
    emb, state = net.direct_forward(x12[:BATCH].contiguous(), x12[BATCH:].contiguous())
    d_out = torch.randn_like(emb) * 1e-3

    def bwd():
        net.direct_backward(state, d_out, d_out_is_dz=True)
    t = _time_launches(torch, bwd, reps)
    fl = 2.0 * rows * (2 * (40 * 500 + 2 * 500 * 500 + 500 * 100) - 40 * 500)      # wgrad everywhere, no dgrad into the input
    out['backward'] = {
        'kernels': 'gemm_bwd_pair_kernel<128, 64, .> x2 + <64, 64, .> x1, gemm_f32_kernel<64, 64, false, false, 2, ...> '
                   '(input-layer wgrad), slab_reduce',
        'achieved': round(fl / t / 1e12, 2), 'frac': round(fl / t / 1e12 / peak, 4),
        'avg_sequence_us': round(t * 1e6, 2), 'flop_per_sequence': fl}
    return out


def host_cores():
    """the box's CPU share for one GPU is 16 cores (more threads than that only
    oversubscribe the cgroup); never more than the affinity mask allows"""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(16, avail))


def cpu_baseline(torch, budget_s=12.0, threads=None):
    """The reference's train step as the oracle restates it with torch.nn on
    the CPU (oracle/torch_ref.py, pinned by tests/golden/train_c2_*), timed on
    this host's cores on a bounded sample of the same workload.  threads: all of the
    box's share (default), or 1 (BASELINE.md section 3 lists both)."""
    from oracle import torch_ref
    torch.set_num_threads(threads or host_cores())
    net = torch_ref.build(seed=0, **C2)
    opt = torch.optim.Adadelta(net.parameters(), lr=0.1)
    pool = [torch_ref.make_inputs(BATCH, 40, s) for s in range(2)]
    pool = [(a, b, torch.from_numpy(y)) for a, b, y in pool]
    net.train()
    for i in range(3):
        torch_ref.train_step(net, opt, *pool[i % 2])
    t0 = time.perf_counter()
    n = 0
    while True:
        torch_ref.train_step(net, opt, *pool[n % 2])
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 2000:
            break
    dt = time.perf_counter() - t0
    return {'value': round(n * BATCH / dt, 1), 'unit': 'frame-pairs/s',
            'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '%d steps of the same C2 step (B=%d, fp32, torch-CPU restatement '
                      'of the reference, %.1f s)' % (n, BATCH, dt)}


def synth_dtw_pairs(P, seed, D=40):
    """SURVEY.md 8d "C4": lengths ~ clip(round(N(300, 60)), 50, 600); the second
    token of a pair is a time-warped noisy copy of the first."""
    import numpy as np
    rng = np.random.default_rng(seed)
    n1 = np.clip(np.rint(rng.normal(300, 60, P)), 50, 600).astype(np.int32)
    n2 = np.clip(np.rint(rng.normal(300, 60, P)), 50, 600).astype(np.int32)
    o1 = np.concatenate(([0], np.cumsum(n1)[:-1])).astype(np.int64)
    o2 = np.concatenate(([0], np.cumsum(n2)[:-1])).astype(np.int64)
    f1 = rng.standard_normal((int(n1.sum()), D), dtype=np.float32)
    f2 = np.empty((int(n2.sum()), D), dtype=np.float32)
    noise = rng.standard_normal(f2.shape, dtype=np.float32)
    for p in range(P):
        src = np.rint(np.linspace(0, n1[p] - 1, n2[p])).astype(np.int64)
        f2[o2[p]:o2[p] + n2[p]] = f1[o1[p] + src] + 0.1 * noise[o2[p]:o2[p] + n2[p]]
    return f1, o1, n1, f2, o2, n2


def make_stepper(trainer, pool, use_graph):
    """step(i) for batch i of the pool.  Default: TrainerSiamese.train_step, which drives
    the kernels itself (no autograd, no graph, no input copy).  --graph replays a captured
    hipGraph instead (fed by packed batch blobs): it needs one device copy per step and is
    ~3 % slower at this size; it pays for batches small enough to be launch-bound."""
    if not use_graph:
        return lambda i: trainer.train_step(pool[i % POOL], True)
    stepper = trainer.make_graphed_step(pool[0])
    packed = [trainer.pack_batch(b) for b in pool]
    return lambda i: stepper(packed[i % POOL])


def variants_bench(torch, pool, args, rank, world):
    """SURVEY.md 8d also lists the C2 step with BatchNorm on, and with the trainer's
    class-default optimizer SGD(0.001, 0.9); the third variant is the model's class-default
    p_dropout = 0.1: same workload, shorter runs, reported beside the headline (canonical:
    no BatchNorm, no dropout, Adadelta(0.1))."""
    from abnet3_amd.model import SiameseNetwork
    from abnet3_amd.loss import coscos2
    from abnet3_amd.trainer import TrainerSiamese
    out = {}
    for name, bn, opt, lr, pdrop in (('batch_norm_adadelta', True, 'adadelta', 0.1, 0.0), ('no_bn_sgd', False, 'sgd', 0.001, 0.0),
                                     ('dropout_0.1_adadelta', False, 'adadelta', 0.1, 0.1)):      # (0.1: the reference's default p_dropout)
        torch.manual_seed(0)
        net = SiameseNetwork(output_path='/tmp/abnet3_bench_v%d' % rank, **dict(C2, batch_norm=bn, p_dropout=pdrop))
        tr = TrainerSiamese(network=net, loss=coscos2(avg=False), optimizer_type=opt, lr=lr,
                            dataloader=None, log_dir='/tmp/abnet3_bench_runs')
        net.train()
        step = make_stepper(tr, pool, args.graph)
        steps = max(SIDE_STEPS, args.steps // 4)
        settle(torch, step)
        for i in range(20):
            step(i)
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=pool[0][0].device)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            elapsed = float(t.item())
        out[name] = {'value': round(steps * BATCH * world / elapsed, 1), 'unit': 'frame-pairs/s',
                     'ms_per_step': round(elapsed / steps * 1e3, 4), 'steps': steps}
    return out


def settle(torch, fn, seconds=0.3, chunk=16):
    """Untimed calls of fn(i) for `seconds`: a side leg starts behind CPU-side set-up (a new network, synthetic inputs) during
    which the GPU idles and its clock falls; the first ~30 ms of work behind an idle second run up to 25 % slower
    (tools/dtw_time_many.py: 3.79, 3.30, 3.15, 3.08, 3.00 .. 2.87 ms for consecutive DTW calls).  The headline loop settles the
    same way (0.5 s) before its warm-up."""
    t0, i = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds * SETTLE_SCALE:
        for _ in range(chunk):
            fn(i)
            i += 1
        torch.cuda.synchronize()


def mode_bench(torch, trainer, net, pool, args, world, prec, note):
    """The same C2 step in another arithmetic of the tower GEMMs (SiameseNetwork.precision),
    reported BESIDE the headline value, with its embedding error against the exact-fp32 mode."""
    x1, x2, _ = pool[0]
    default = net.precision
    net.eval()
    with torch.no_grad():
        x12 = torch.cat([x1, x2])                   # both towers' rows: the launch shape of the timed step
        net.precision = 'fp32'
        ref = torch.cat(net.forward_pair_rows(x12))
        net.precision = prec
        got = torch.cat(net.forward_pair_rows(x12))
    err = float((got - ref).abs().max() / ref.abs().max())
    net.train()
    step = make_stepper(trainer, pool, args.graph)
    steps = max(SIDE_STEPS, args.steps // 2)
    settle(torch, step)
    for i in range(20):
        step(i)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=x1.device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    net.precision = default
    return {'value': round(steps * BATCH * world / elapsed, 1), 'unit': 'frame-pairs/s',
            'ms_per_step': round(elapsed / steps * 1e3, 4), 'steps': steps,
            'arithmetic': prec, 'max_rel_err_embeddings_vs_exact_f32': err, 'note': note}


def dtw_bench(torch, P, rank, world, reps=15, cpu_pairs=4000):
    """BASELINE.json configs[3]: DTW alignment of P synthetic token pairs (40-d,
    ~300 frames) on this rank's GPU; cells/s = sum(N*M) / wall time of the
    batched call (distance matrix + DP + traceback; features already in HBM)."""
    import numpy as np
    from abnet3_amd.utils import dtw_align_batch
    f1, o1, n1, f2, o2, n2 = synth_dtw_pairs(P, seed=1000 + rank)
    cells = int((n1.astype(np.int64) * n2).sum())
    d1, d2 = torch.from_numpy(f1).cuda(), torch.from_numpy(f2).cuda()
    res = dtw_align_batch(d1, o1, n1, d2, o2, n2)          # (allocations)
    torch.cuda.synchronize()
    # settled like the headline loop (the synthesis above left the GPU idle), then `reps` calls timed ONE BY ONE, each
    # bracketed by a synchronisation (and a barrier across ranks): the MEDIAN call is reported, min / max beside it
    settle(torch, lambda i: dtw_align_batch(d1, o1, n1, d2, o2, n2), seconds=0.3 if P >= 1000 else 0.05, chunk=4)
    calls = []
    for _ in range(reps):
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = dtw_align_batch(d1, o1, n1, d2, o2, n2)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device='cuda')
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        calls.append(dt)
    best = sorted(calls)[len(calls) // 2]
    dropped = int((res.path_len == 0).sum().item())
    out = {'metric': 'DTW cells/sec (cosine distance + DP + traceback, 40-d, ~300-frame tokens)',
           'value': round(cells * world / best, 1), 'unit': 'cells/s', 'pairs_per_gpu': P,
           'cells_per_gpu': cells, 'ms': round(best * 1e3, 3), 'ms_min_max': [round(min(calls) * 1e3, 3), round(max(calls) * 1e3, 3)],
           'calls': len(calls), 'ms_is': 'the median of %d calls timed one by one behind 0.3 s of untimed calls (clocks settled, as in the headline loop)' % len(calls),
           'dropped_pairs': dropped,
           # SURVEY.md 8d names the vector ALU as the binding unit: per cell one CORRECTLY ROUNDED
           # division, glibc's acosf (two polynomials, a second division), a division by pi -- the
           # reference's arithmetic operation by operation -- then the float64 three-way minimum; the dot
           # products run on the fp32 matrix cores (same rate as the packed vector fma: 32 MAC / clk / SIMD).
           # `roofline` prices SURVEY's 100 FLOP per cell against the fp32 vector peak.  `hbm`: MEASURED bytes per
           # call (TCC counters collected under rocprofv3, profiles/) over this run's time -- what the kernel
           # really pulls -- beside the ALGORITHMIC bytes (inputs once (N+M)*40*4, paths <= (N+M)*8): the
           # cost matrix never leaves the CU, 2-bit back-pointers and the paths are the writes, the fetches
           # are token 2's rows once per 64 rows of token 1 (two coupled 32-row bands, round 6) and the boundary rows.
           'roofline': {'bound': 'valu', 'achieved': round(cells * 100.0 / best / 1e12, 2), 'peak': 157.3,
                        'unit': 'TFLOP/s', 'flop_per_cell': 100,
                        'note': 'VALU-bound: exact division / acosf / pi per cell (all 64 lanes of a producer wavefront per '
                                'slot) + N+M-1 sequential anti-diagonals of f64 min / add per band (32 lanes per pair, ~20 '
                                'instructions per step in the consumer wavefront); dot products on the fp32 matrix cores; '
                                'one fused kernel, the matrix stays in LDS',
                        # (the committed counters are of the default workload, 10 000 pairs: no figure for another size)
                        'traffic': _traffic('dtw_gang_kernel') if P == 10000 else None}}
    alg_bytes = int((n1.astype(np.int64) + n2).sum()) * 168
    meas = out['roofline']['traffic']
    out['roofline']['hbm'] = {
        'measured_bytes_per_call': meas, 'algorithmic_bytes_per_call': alg_bytes, 'peak': 8000.0, 'unit': 'GB/s',
        'achieved': round(meas / best / 1e9, 2) if meas else None,                       # measured traffic / this run's time
        'algorithmic_achieved': round(alg_bytes / best / 1e9, 2),
        'bytes_per_cell_measured': round(meas / cells, 3) if meas else None,
        'bytes_per_cell_algorithmic': round(alg_bytes / cells, 3)}
    out['roofline']['frac'] = round(out['roofline']['achieved'] / 157.3, 4)
    out['roofline']['hbm']['frac'] = round(out['roofline']['hbm']['achieved'] / 8000.0, 5) if meas else None
    if rank == 0 and world == 1:
        from oracle import dtw_oracle
        q = min(cpu_pairs, P)
        sub_cells = int((n1[:q].astype(np.int64) * n2[:q]).sum())
        t0 = time.perf_counter()
        p1, p2, ln, c = dtw_oracle.dtw_batch(f1, o1[:q], n1[:q], f2, o2[:q], n2[:q], 1200)
        dt = time.perf_counter() - t0
        got = res.to_lists()
        exact = all(got[i] is not None and (got[i][0] == p1[i, :ln[i]]).all() and
                    (got[i][1] == p2[i, :ln[i]]).all() for i in range(q))
        out['cpu_baseline'] = {'value': round(sub_cells / dt, 1), 'unit': 'cells/s', 'cores': 1,
                               'kind': 'port', 'sample': '%d of the %d pairs (%d cells, %.1f s), oracle/dtw.c'
                               % (q, P, sub_cells, dt)}
        out['paths_bit_exact_on_sample'] = bool(exact)
        # ... and on all of the box's cores (OpenMP over the pairs, which are independent: the same paths)
        cores = host_cores()
        qa = min(P, cpu_pairs * min(cores, 4))
        cells_a = int((n1[:qa].astype(np.int64) * n2[:qa]).sum())
        t0 = time.perf_counter()
        a1, a2, aln, _ = dtw_oracle.dtw_batch(f1, o1[:qa], n1[:qa], f2, o2[:qa], n2[:qa], 1200, threads=cores)
        dta = time.perf_counter() - t0
        out['cpu_baseline_all_cores'] = {'value': round(cells_a / dta, 1), 'unit': 'cells/s', 'cores': cores, 'kind': 'port',
                                         'sample': '%d of the %d pairs (%d cells, %.1f s), oracle/dtw.c, OpenMP over the pairs'
                                         % (qa, P, cells_a, dta),
                                         'same_paths_as_one_core': bool((aln[:q] == ln).all() and (a1[:q] == p1).all() and (a2[:q] == p2).all())}
    return out


def collective_bench(torch, trainer, net, world, reps=50):
    """What the data-parallel step's gradient exchange costs ON ITS OWN, per call, on the step's own bucket (the flat
    gradient buffer: 2.29 MB at C2), with HIP events on the launch stream -- BASELINE.md section 3, row C3: "all-reduce us" --
    for (a) torch.distributed.all_reduce (RCCL's ring / tree on the "nccl" backend) and (b) abn_allreduce_oneshot (one
    launch per rank over peer-mapped mailboxes); plus who the ranks are: the backend's name, the world size as
    torch.distributed sees it, and the rank count of a communicator this process made itself on the RCCL library torch has
    loaded (ncclCommCount on RcclBatchNormSync's communicator: RCCL's own word for how many ranks it connected).
    EVERY rank runs this (the calls are collectives); rank 0 reports.  Nothing here has been measured over xGMI unless
    `backend` says "nccl" and `world` > 1 on a multi-GPU node."""
    import ctypes as C
    from abnet3_amd import parallel
    dist = torch.distributed
    out = {'backend': dist.get_backend(), 'world': dist.get_world_size(), 'rank_count_seen_by_rccl': None,
           'bucket_bytes': int(net.flat_parameters().numel()) * 4,
           'step_uses': 'abn_allreduce_oneshot' if trainer.oneshot is not None else
                        ('torch.distributed.all_reduce, two buckets (the upper layers\' under the rest of the backward)'
                         if trainer.overlap_allreduce else 'torch.distributed.all_reduce, one bucket')}
    if out['backend'] == 'nccl':
        # (in a thread with a deadline: a second communicator beside torch's has only ever been made on ONE rank here, and a
        # rendezvous that never completes must not take the measured line with it)
        import threading
        dev_index = torch.cuda.current_device()

        def count_ranks():
            try:
                torch.cuda.set_device(dev_index)
                sync = parallel.RcclBatchNormSync()
                n = C.c_int(-1)
                sync._rccl.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
                if sync._rccl.ncclCommCount(sync._comm, C.byref(n)) == 0:
                    out['rank_count_seen_by_rccl'] = int(n.value)
                sync._destroy()
            except Exception as e:                  # noqa: BLE001
                out['rank_count_seen_by_rccl'] = 'unavailable: %s' % (e,)
        th = threading.Thread(target=count_ranks, daemon=True)
        th.start()
        th.join(90.0)
        if th.is_alive():
            out['rank_count_seen_by_rccl'] = 'unavailable: making a second communicator did not finish within 90 s'
    buf = torch.zeros_like(net.flat_parameters())

    def timed(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        t = torch.tensor([a.elapsed_time(b) * 1e3 / reps], dtype=torch.float64, device=buf.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return round(float(t.item()), 2)
    out['allreduce_us'] = {'torch.distributed.all_reduce': timed(lambda: parallel.all_reduce_gradients(buf, False))}
    one, made_here = trainer.oneshot, False
    if one is None:
        try:
            one, made_here = parallel.OneShotAllReduce(buf.numel()), True
        except (RuntimeError, OSError, AttributeError, ValueError) as e:
            out['allreduce_us']['abn_allreduce_oneshot'] = 'unavailable: %s' % (e,)
    if one is not None:
        # one call first, checked: a rank that gives up waits a minute per call, and nothing here has run over xGMI yet
        probe = torch.ones_like(buf)
        one.all_reduce(probe)
        torch.cuda.synchronize()
        good = (not one.failed()) and bool((probe == float(world)).all())
        if parallel.all_agree(good):
            out['allreduce_us']['abn_allreduce_oneshot'] = timed(lambda: one.all_reduce(buf))
        else:
            out['allreduce_us']['abn_allreduce_oneshot'] = 'not timed: the first call gave up or summed wrongly on some rank'
        out['oneshot_gave_up'] = bool(one.failed())
        if made_here:
            one.close()
    out['note'] = ('per call, alone on the stream (in the step the first bucket\'s exchange runs under the rest of the backward); '
                   'max over ranks; ' + ('RCCL' if out['backend'] == 'nccl' else 'NOT RCCL: a rehearsal backend, host memory in the path'))
    return out


def fbank_bench(torch, seconds=3000, fs=16000, cpu_seconds=600):
    """Filterbank leg (BASELINE.json configs[4] front end): log-mel energies of
    `seconds` of synthetic 16 kHz int16 audio already resident in HBM (50 minutes: a short
    input is launch-bound -- 10 minutes take 0.135 ms); frames/s.
    CPU: the oracle's numpy restatement on a bounded sample."""
    import numpy as np
    from abnet3_amd.features import FeaturesGenerator
    rng = np.random.default_rng(7)
    n = seconds * fs
    t = np.arange(n) / fs
    sig = (2000 * np.sin(2 * np.pi * 440 * t) + 500 * rng.standard_normal(n)).astype(np.int16)
    fg = FeaturesGenerator()
    d = torch.from_numpy(sig).cuda()
    out = fg.fbank_from_samples(d, fs)
    torch.cuda.synchronize()
    settle(torch, lambda i: fg.fbank_from_samples(d, fs), seconds=0.2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        out = fg.fbank_from_samples(d, fs)
    e1.record()
    torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) * 1e-3 / 20
    frames = out.shape[0]
    res = {'metric': 'filterbank frames/sec (25 ms / 10 ms, nfft 1024, 40 mel bands)', 'value': round(frames / dt, 1),
           'unit': 'frames/s', 'frames': frames, 'ms': round(dt * 1e3, 3),
           'roofline': {'bound': 'hbm', 'achieved': round((n * 2 + frames * 160) / dt / 1e9, 3), 'peak': 8000.0,
                        'unit': 'GB/s', 'traffic': _traffic('fbank1024_kernel'),
                        'note': 'algorithmic bytes = 2 B/sample in + 160 B/frame out; the kernel is VALU/LDS-bound: '
                                'one wavefront per frame, 512-point complex radix-8 FFT + split + sparse mel (~900 '
                                'instructions per frame)'}}
    res['roofline']['frac'] = round(res['roofline']['achieved'] / 8000.0, 6)
    from oracle import features_np
    m = cpu_seconds * fs
    t0 = time.perf_counter()
    ref = features_np.fbank(sig[:m], fs)
    cdt = time.perf_counter() - t0
    res['cpu_baseline'] = {'value': round(ref.shape[0] / cdt, 1), 'unit': 'frames/s', 'cores': 1, 'kind': 'port',
                           'sample': 'first %d s of the same signal (%d frames, %.1f s), oracle/features_np.py' % (cpu_seconds, ref.shape[0], cdt)}
    res['max_abs_err_vs_oracle_on_sample'] = float(np.abs(out[:ref.shape[0] - 3].cpu().numpy() - ref[:-3]).max())
    return res


def pipeline_bench(torch, n_utts, n_pairs, epochs, with_cpu):
    """BASELINE.json configs[4] on this GPU (tools/c5_pipeline.py): synthetic ZeroSpeech-shaped corpus (16 kHz, 2-10 s
    utterances, Zipfian word types, sampler-format word pairs, 70 / 30 split) -> batched filterbanks -> mean / variance
    normalisation -> 7-frame stacking (280-d) -> DTW mining -> TrainerSiamese over OriginalDataLoader (8 word pairs per
    batch: the reference's canonical loader, test/data/buckeye.yaml:21-26) and over FramesDataLoader (4096 frame pairs)
    -> embedding of every utterance.  Per-stage seconds and frame pairs / s of the training passes; with the CPU
    baselines the oracles' rates for the same stages on a bounded slice of the same corpus."""
    import numpy as np
    from tools import c5_pipeline
    out, kept, dc, corpus, (train_pairs, dev_pairs) = c5_pipeline.run(n_utts=n_utts, n_pairs=n_pairs, epochs=epochs, keep=True)
    for kind in out['training']:
        st = out['training'][kind]
        st['losses_decrease'] = bool(st['train_losses'][-1] < st['train_losses'][0])
    if with_cpu:
        from oracle import dtw_oracle, features_np, torch_ref
        cpu = {'kind': 'port', 'cores_train': None}
        # filterbanks: the numpy restatement, one core, the first utterances (~60 s of audio)
        t0, frames, k = time.perf_counter(), 0, 0
        while k < len(corpus.waves) and time.perf_counter() - t0 < 4.0:
            frames += features_np.fbank(corpus.waves[k], corpus.fs).shape[0]
            k += 1
        cpu['fbank_frames_per_s'] = round(frames / (time.perf_counter() - t0), 1)
        cpu['fbank_sample'] = 'first %d utterances (%d frames), oracle/features_np.py, 1 core' % (k, frames)
        # DTW mining: oracle/dtw.c on the stacked 280-d features of the first 'same' pairs, one core
        table = dc.table.cpu().numpy()
        same = [p for p in train_pairs if p[6] == 'same'][:400]
        o1, n1, o2, n2 = [], [], [], []
        for f1, s1, e1, f2, s2, e2, _ in same:
            (a0, na), (b0, nb) = dc.token(f1, s1, e1), dc.token(f2, s2, e2)
            o1.append(a0); n1.append(na); o2.append(b0); n2.append(nb)
        o1, o2 = np.asarray(o1, dtype=np.int64), np.asarray(o2, dtype=np.int64)
        n1, n2 = np.asarray(n1, dtype=np.int32), np.asarray(n2, dtype=np.int32)
        t0 = time.perf_counter()
        dtw_oracle.dtw_batch(table, o1, n1, table, o2, n2, int((n1 + n2).max()))
        dt = time.perf_counter() - t0
        cells = int((n1.astype(np.int64) * n2).sum())
        cpu['dtw_cells_per_s'] = round(cells / dt, 1)
        cpu['dtw_sample'] = '%d same pairs of the training set (%d cells of 280-d frames, %.2f s), oracle/dtw.c, 1 core' % (len(same), cells, dt)
        # training: the torch-CPU restatement on the first batches of the OriginalDataLoader plan (ragged sizes)
        torch.set_num_threads(host_cores())
        plan = kept['original'][1].plan(True)
        batches = [tuple(t.cpu() for t in plan.materialise(b)) for b in plan.order[:60]]
        net = torch_ref.build(seed=0, **c5_pipeline.C5_NET)
        opt = torch.optim.Adadelta(net.parameters(), lr=0.1)
        net.train()
        for b in batches[:3]:
            torch_ref.train_step(net, opt, *b)
        t0, pairs, k = time.perf_counter(), 0, 0
        while time.perf_counter() - t0 < 6.0:
            b = batches[k % len(batches)]
            torch_ref.train_step(net, opt, *b)
            pairs += len(b[2])
            k += 1
        dt = time.perf_counter() - t0
        cpu['train_frame_pairs_per_s'] = round(pairs / dt, 1)
        cpu['cores_train'] = torch.get_num_threads()
        cpu['train_sample'] = '%d steps over the first %d batches of the same OriginalDataLoader plan (%.1f s), oracle/torch_ref.py' % (k, len(batches), dt)
        out['cpu_baseline'] = cpu
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=30)
    ap.add_argument('--repeats', type=int, default=9, help='timed regions of --steps steps each; the median is reported')
    ap.add_argument('--trace-run', action='store_true',
                    help='for a run under rocprofv3 --kernel-trace (tools/collect_profiles.sh): the side legs do not settle the '
                         'clock with thousands of untimed launches, so the trace\'s per-kernel averages are the timed step\'s own')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--dtw-pairs', type=int, default=10000,
                    help='token pairs per GPU for the DTW leg (0 = skip)')
    ap.add_argument('--pipeline-utts', type=int, default=2000, help='utterances of the C5 pipeline leg (0 = skip; N = 1 only)')
    ap.add_argument('--graph', action='store_true',
                    help='replay a captured hipGraph per step instead of the eager direct step')
    ap.add_argument('--no-graph', action='store_true', help='(default; kept for old command lines)')
    args = ap.parse_args()
    if args.trace_run:
        global SETTLE_SCALE
        SETTLE_SCALE = 0.0

    import torch
    from abnet3_amd import parallel
    # backend nccl = RCCL over xGMI; ABN_DIST_BACKEND=gloo lets the multi-process
    # path be rehearsed on a box with fewer GPUs than ranks
    rank, world, local = parallel.init_from_env(os.environ.get('ABN_DIST_BACKEND', 'nccl'))
    local = local % max(1, torch.cuda.device_count())
    if world != args.gpus:
        if args.gpus > 1:
            raise SystemExit('bench.py --gpus %d must be launched with torch.distributed.run '
                             '--nproc-per-node %d' % (args.gpus, args.gpus))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    from abnet3_amd.model import SiameseNetwork
    from abnet3_amd.loss import coscos2
    from abnet3_amd.trainer import TrainerSiamese

    torch.manual_seed(0)                       # identical replicas
    net = SiameseNetwork(output_path='/tmp/abnet3_bench_r%d' % rank, **C2)
    trainer = TrainerSiamese(network=net, loss=coscos2(avg=False), optimizer_type='adadelta',
                             lr=0.1, dataloader=None, log_dir='/tmp/abnet3_bench_runs')
    pool = make_pool(seed=rank, device=dev)    # rank r sees its own pairs
    net.train()
    step = make_stepper(trainer, pool, args.graph)
    from abnet3_amd import _lib as _L
    _L.trace_paths = True                      # one untimed step records which kernel family / arithmetic the step takes
    step(0)
    _L.trace_paths = False

    # untimed, and independent of --warmup: let clocks, allocator and code caches settle
    # (a short `--steps 20 --warmup 5` run otherwise times the first 8 ms after start-up)
    t_settle, i = time.perf_counter(), 0
    while time.perf_counter() - t_settle < 0.5:
        for _ in range(32):
            step(i)
            i += 1
        torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    # EXACTLY --steps steps between barrier + synchronize on both sides, MAX over ranks -- repeated --repeats
    # times back to back; the line's value is the MEDIAN repeat (a 20-step region is a 4 ms sample: one
    # repeat alone reads +-5 % from run to run), min / max beside it
    elapsed_all = []
    for rep in range(max(1, args.repeats)):
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            loss = step(i)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            el = float(t.item())
        elapsed_all.append(el)
    elapsed = sorted(elapsed_all)[len(elapsed_all) // 2]
    last_loss = float(loss)

    err_default = None
    if rank == 0:
        with torch.no_grad():
            net.eval()
            x1, x2 = pool[0][0], pool[0][1]
            keep = net.precision
            x12 = torch.cat([x1, x2])               # both towers' rows: the launch shape of the timed step
            net.precision = 'fp32'
            ref = torch.cat(net.forward_pair_rows(x12))
            net.precision = keep
            err_default = float((torch.cat(net.forward_pair_rows(x12)) - ref).abs().max() / ref.abs().max())
            net.train()
    headline_precision = net.precision

    def side(fn, *a):
        # (the headline is measured; a side leg that fails says so in its place instead of taking the line with it)
        try:
            return fn(*a)
        except Exception as e:                      # noqa: BLE001
            import traceback
            traceback.print_exc(file=sys.stderr)
            return {'error': '%s: %s' % (type(e).__name__, e)}
    x3 = side(mode_bench, torch, trainer, net, pool, args, world, 'bf16x3',
              'fp32 operands split into 3 bf16 terms, 6 bf16 MFMA products per operand pair: parity-grade, the default of rounds 2-3') \
        if net.precision != 'bf16x3' else None
    f32x = side(mode_bench, torch, trainer, net, pool, args, world, 'fp32',
                'exact-fp32 MFMA (v_mfma_f32_32x32x2_f32): one sequential fp32 fma chain per output, round 1\'s headline arithmetic')
    bf16 = side(mode_bench, torch, trainer, net, pool, args, world, 'bf16',
                'operands rounded to bf16 once (~3 digits): outside the 1e-5 parity bar, never the headline value')
    variants = side(variants_bench, torch, pool, args, rank, world)
    dtw = side(dtw_bench, torch, args.dtw_pairs, rank, world) if args.dtw_pairs > 0 else None
    coll = side(collective_bench, torch, trainer, net, world) if world > 1 else None
    net.precision = headline_precision              # (a side mode that failed half-way may have left its own)

    if rank == 0:
        value = args.steps * BATCH * world / elapsed
        out = {
            'metric': 'frame-pairs/sec (Siamese train step: fwd both towers + coscos2 + bwd + Adadelta)',
            'value': round(value, 1), 'unit': 'frame-pairs/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 4),
            'repeats': len(elapsed_all),
            'ms_per_step_min_max': [round(min(elapsed_all) / args.steps * 1e3, 4), round(max(elapsed_all) / args.steps * 1e3, 4)],
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'fp32': 'f32', 'bf16': 'bf16', 'bf16x3': 'bf16x3', 'f16x2': 'f16x2'}[net.precision],
            'data': 'synthetic',
            'config': {'workload': 'C2: SiameseNetwork 40->500x2->100 sigmoid, coscos2(avg=False), '
                                   'Adadelta(0.1), 4096 frame pairs per GPU per step, 40-d N(0,1) frames',
                       'pairs_per_gpu': BATCH, 'global_pairs': BATCH * world,
                       'parallelism': 'dp%d' % world, 'graph_replay': bool(args.graph),
                       'dp_gradient_buckets': (2 if world > 1 and trainer.overlap_allreduce else 1),   # (the first all-reduce under the rest of the backward)
                       'arithmetic': {'f16x2': 'f16x2: fp32 operands scaled by a power of two per row / block and split into 2 fp16 '
                                               'terms (22 bits), 3 fp16 MFMA products per operand pair, fp32 accumulate / storage / '
                                               'loss / optimizer; parity-grade (all golden tests at 1e-5; no further from float64 '
                                               'than the reference\'s fp32), see bf16x3_mode / f32_exact_mode for the same step in '
                                               'the other parity-grade arithmetics',
                                      'bf16x3': 'bf16x3: fp32 operands split into 3 bf16 terms, 6 bf16 MFMA products per '
                                                'operand pair, fp32 accumulate / storage / loss / optimizer; parity-grade '
                                                '(all golden tests at 1e-5), see f32_exact_mode for the exact-fp32 MFMA step'
                                      }.get(net.precision, net.precision),
                       'max_rel_err_embeddings_vs_exact_f32': err_default},
            'tflops_whole_step': round(value * FLOP_PER_PAIR / 1e12, 2),
            'last_loss': last_loss,
        }
        out['roofline'] = tower_roofline(torch, net) if net.precision == 'fp32' else planes_roofline(torch, net)
        # the whole step against the same roof, from the timed loop above (pack, both chains, weight gradients,
        # reduction + optimizer, every gap): the number the per-launch fractions have to be read beside
        out['roofline']['whole_step'] = {'achieved': round(value * FLOP_PER_PAIR / world / 1e12, 2), 'unit': 'TFLOP/s',
                                         'frac': round(value * FLOP_PER_PAIR / world / 1e12 / out['roofline']['peak'], 4),
                                         'frac_of_dense_16bit_peak': round(value * FLOP_PER_PAIR / world / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4),
                                         'ms_per_step': round(elapsed / args.steps * 1e3, 4)}
        # the arithmetic the step's launches really ran in (abn_tower_path: a 'f16x2' tower that falls to the GEMM
        # kernels computes in bf16x3 there; C2 stays on the operand planes)
        from abnet3_amd import _lib as _L
        if _L.last_path['forward'] >= 0:
            out['config']['kernel_path'] = {'forward': _L.last_path['forward'], 'backward': _L.last_path['backward'],
                                            'forward_arithmetic': _L.PRECISION_NAMES.get(_L.last_path['forward_precision']),
                                            'backward_arithmetic': _L.PRECISION_NAMES.get(_L.last_path['backward_precision'])}
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(torch)
            out['cpu_baseline'] = cb
            out['gpu_over_cpu'] = round(value / cb['value'], 1)
            out['cpu_baseline_1_thread'] = cpu_baseline(torch, budget_s=8.0, threads=1)
            torch.set_num_threads(host_cores())
        if x3 is not None:
            out['bf16x3_mode'] = x3
        out['f32_exact_mode'] = f32x
        if bf16 is not None:
            out['bf16_throughput_mode'] = bf16
        if variants is not None:
            out['f32_variants'] = variants
        if dtw is not None:
            out['dtw'] = dtw
        if coll is not None:
            out['collective'] = coll
        if world == 1 and not args.no_cpu_baseline:
            out['fbank'] = fbank_bench(torch)
        if world == 1 and args.pipeline_utts > 0:
            import contextlib
            with contextlib.redirect_stdout(sys.stderr):      # (the trainer reports its epochs like the reference: not on this line's stream)
                out['pipeline'] = pipeline_bench(torch, args.pipeline_utts, 25 * args.pipeline_utts, 2, not args.no_cpu_baseline)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
