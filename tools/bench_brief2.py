"""The figures of a bench.py line the docs quote:  python tools/bench_brief2.py gpurun_out/r04/bench_line.json"""
import json, sys
d = json.load(open(sys.argv[1]))
r = d['roofline']
print('C2', d['value'], d['ms_per_step'], d['ms_per_step_min_max'], 'whole', r.get('whole_step'))
print('dominant', r['dominant'], r['avg_launch_us'], r['frac'], r.get('frac_of_dense_16bit_peak'), 'traffic', r['traffic'])
for k in ('weight_gradients', 'dgrad_chain', 'forward'):
    if k in r:
        print(' ', k, r[k]['avg_launch_us'], r[k]['frac'], r[k]['traffic'])
print('variants', {k: (v['value'], v['ms_per_step']) for k, v in d['f32_variants'].items()})
print('cpu', d.get('cpu_baseline'))
print('dtw', d['dtw']['value'], d['dtw']['ms'], d['dtw']['roofline']['frac'], d['dtw']['roofline']['hbm'])
if 'pipeline' in d:
    p = d['pipeline']
    print('stages', p['stages'])
    for k, v in p['training'].items():
        print(' ', k, v.get('train_frame_pairs_per_s'), v.get('us_per_step'), 'mining', v['mining_s'], v.get('mining_s_second_loader'), 'epoch', v['train_pass_s'], 'dev', v['dev_pass_s'], 'embed', v['embed_s'])
