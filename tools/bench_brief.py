import json, sys
for f in sys.argv[1:]:
    for line in open(f):
        if line.startswith('{"metric"'):
            d = json.loads(line)
            print(f, 'ms/step', d['ms_per_step'], 'err', d['config']['max_rel_err_embeddings_vs_exact_f32'], '| fp32', d['f32_exact_mode']['ms_per_step'],
                  '| bf16', d['bf16_throughput_mode']['ms_per_step'], d['bf16_throughput_mode']['max_rel_err_embeddings_vs_exact_f32'])
