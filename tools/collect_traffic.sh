#!/bin/bash
# HBM traffic of the step kernels (fused forward, dgrad, wgrad, ...) from the TCC counters, collected as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc
# passes (3 + 2 TCC slots), kernel-trace only.  Run on the GPU box:
#   tools/collect_traffic.sh gpurun_out/traffic
set -e
out=${1:-gpurun_out/traffic}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --dtw-pairs 0 > $out.fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --dtw-pairs 0 > $out.write.log 2>&1
python3 tools/traffic_summary.py $out
