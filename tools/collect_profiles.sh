#!/bin/bash
# Everything profiles/ holds for a round, collected on the GPU box as MI355X_MICROARCH.md prescribes:
#   1. rocprofv3 --kernel-trace --stats of the default bench.py command (kernel durations)
#   2. FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (kernel-trace only) -> HBM bytes per launch
#   3. SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES / SQ_WAVE_CYCLES / SQ_INSTS_VALU ... pass (matrix-core and VALU use)
#   4. the same for the small-batch step (planned passes, 485 pairs of 280-d frames: tools/small_batch_probe.py)
#   tools/collect_profiles.sh r04      (results under gpurun_out/r03*, summaries copied by the caller)
#   ONLY="small extras summary" tools/collect_profiles.sh r06     (stages: bench small extras summary; default all.  The
#   summaries read only the CSVs, so they can also be made in the build container from what gpurun merged back.)
set -e
tag=${1:-r04}
ONLY=${ONLY:-bench small extras summary}
has() { case " $ONLY " in *" $1 "*) return 0;; esac; return 1; }
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
args="--steps 100 --warmup 20 --no-cpu-baseline --pipeline-utts 0 --trace-run"
if has bench; then
sha1sum $root/abnet3_amd/lib/libabnet3_hip.so | cut -d' ' -f1 > $out/lib_digest.txt      # (which build the trace is of: bench.py compares)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py $args > $out/stats.log 2>&1
echo stats done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/traffic/fetch -- python3 $root/bench.py $args > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/traffic/write -- python3 $root/bench.py $args > $out/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $out/sq -- python3 $root/bench.py $args > $out/sq.log 2>&1
# filterbank leg (bench runs it only with the CPU baselines): its own passes
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/traffic_fb/fetch -- python3 $root/tools/fbank_time.py 3000 > $out/fb_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/traffic_fb/write -- python3 $root/tools/fbank_time.py 3000 > $out/fb_write.log 2>&1
fi
if has small; then
export MODE=plan PAIRS=485
rocprofv3 --kernel-trace --stats --output-format csv -d $out/small_stats -- python3 $root/tools/small_batch_probe.py > $out/small_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/traffic_small/fetch -- python3 $root/tools/small_batch_probe.py > $out/small_fetch.log 2>&1 || echo "small fetch pass failed (profiler): see small_fetch.log"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/traffic_small/write -- python3 $root/tools/small_batch_probe.py > $out/small_write.log 2>&1 || echo "small write pass failed (profiler): see small_write.log"
echo small done
fi
cd $root
if has summary; then
python3 tools/prof_summary.py $out/small_stats 12 > $out/small_batch_kernel_stats.txt
python3 tools/traffic_summary.py $out/traffic_small > $out/traffic_small.txt
python3 tools/prof_summary.py $out/stats 20 > $out/kernel_stats.txt
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
python3 tools/traffic_summary.py $out/traffic > $out/traffic.txt
python3 tools/traffic_summary.py $out/traffic_fb > $out/traffic_fb.txt
python3 tools/pmc_summary.py $out/sq abn:: > $out/sq_counters.txt
grep -h "^{\"metric\"" $out/stats.log | tail -1 > $out/bench_line_under_rocprof.json
echo collected $out
fi
if has extras; then
# round 5: the BatchNorm step (resident tower: one launch per direction) and the filterbank's LDS counters
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bn_stats -- python3 $root/tools/step_prof_bn.py > $out/bn_stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --output-format csv -d $out/fbank_sq -- python3 $root/tools/fbank_time.py 3000 > $out/fbank_sq.log 2>&1
fi
cd $root
if has summary; then
python3 tools/prof_summary.py $out/bn_stats 10 > $out/bn_step_kernel_stats.txt
python3 tools/pmc_summary.py $out/fbank_sq abn:: > $out/fbank_counters.txt
echo collected round-5 extras
fi
