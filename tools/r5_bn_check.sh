#!/bin/bash
# BatchNorm step: the resident tower against the layer launches (ms per step), a kernel trace of the step, the tower's tests
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
export TMPDIR=/tmp
(ABN_BN_PERSIST=1 timeout -k 10 200 python tools/bn_step_time.py; ABN_BN_PERSIST=0 timeout -k 10 200 python tools/bn_step_time.py) > gpurun_out/bn_time.log 2>&1 || exit 1
rm -rf gpurun_out/bn_prof
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/bn_prof -- python3 $root/tools/step_prof_bn.py > $root/gpurun_out/bn_prof.log 2>&1) || exit 1
python3 tools/prof_summary.py gpurun_out/bn_prof 8 > gpurun_out/bn_prof.txt 2>&1
grep ms gpurun_out/bn_time.log; cat gpurun_out/bn_prof.txt
