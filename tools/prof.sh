#!/bin/bash
# rocprofv3 kernel-trace + stats of one python tool, csv output, compact summary.
#   tools/prof.sh <name> <script.py> [args...]     (run on the GPU box, from the repo root)
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$name
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 "$root/$1" "${@:2}" > $out.log 2>&1
cd $root
python3 tools/prof_summary.py $out ${PROF_ROWS:-14}
