#!/bin/bash
# copies what tools/collect_profiles.sh <tag> (+ `python bench.py > gpurun_out/<tag>/bench_line.json`) left under gpurun_out/<tag>/ into profiles/<tag>_*
set -e
tag=${1:-r04}
src=gpurun_out/$tag
cp $src/kernel_stats.csv profiles/${tag}_bench_kernel_stats.csv
cp $src/kernel_stats.txt profiles/${tag}_bench_kernel_stats.txt
[ -s $src/lib_digest.txt ] && cp $src/lib_digest.txt profiles/${tag}_lib_digest.txt
cp $src/bench_line_under_rocprof.json profiles/${tag}_bench_line_under_rocprof.json
[ -s $src/bench_line.json ] && cp $src/bench_line.json profiles/${tag}_bench_line.json
cp $src/small_batch_kernel_stats.txt profiles/${tag}_small_batch_kernel_stats.txt
cp $src/sq_counters.txt profiles/${tag}_sq_counters.txt
cp $src/traffic_small_summary.json profiles/${tag}_pmc_traffic_small.json
python3 - "$src" "$tag" <<'PY'
import json, sys
src, tag = sys.argv[1:3]
t = json.load(open(src + '/traffic_summary.json'))
t.update(json.load(open(src + '/traffic_fb_summary.json')))        # the filterbank leg's own passes
json.dump(t, open('profiles/%s_pmc_traffic.json' % tag, 'w'), indent=1)
PY
[ -s $src/bn_step_kernel_stats.txt ] && cp $src/bn_step_kernel_stats.txt profiles/${tag}_bn_step_kernel_stats.txt
[ -s $src/fbank_counters.txt ] && cp $src/fbank_counters.txt profiles/${tag}_fbank_counters.txt
echo installed profiles/${tag}_*
