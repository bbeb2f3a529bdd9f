#!/bin/bash
# the bench stage of collect_profiles.sh with its summaries made ON the box and the raw traces dropped there (the traces of the
# settled bench.py exceed what gpurun copies back): tools/collect_bench_only.sh r06
set -e
tag=${1:-r06}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
ONLY=bench tools/collect_profiles.sh $tag
cd $root
python3 tools/prof_summary.py $out/stats 20 > $out/kernel_stats.txt
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
python3 tools/traffic_summary.py $out/traffic > $out/traffic.txt
python3 tools/traffic_summary.py $out/traffic_fb > $out/traffic_fb.txt
python3 tools/pmc_summary.py $out/sq abn:: > $out/sq_counters.txt
grep -h "^{\"metric\"" $out/stats.log | tail -1 > $out/bench_line_under_rocprof.json
rm -rf $out/stats $out/traffic $out/traffic_fb $out/sq
du -sh $root/gpurun_out
echo bench stage summarised
