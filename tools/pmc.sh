#!/bin/bash
# One rocprofv3 --pmc pass (kernel-trace only, as gpurun requires) over a python tool; prints
# per-kernel sums of the requested counters.   tools/pmc.sh <name> "<CTR1 CTR2 ...>" <script.py> [args]
name=$1; ctrs=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_$name
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out -- python3 "$root/$1" "${@:2}" > $out.log 2>&1
cd $root
python3 - "$out" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for path in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name'][:60]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        calls[(k, r['Counter_Name'])] += 1
for k, d in agg.items():
    if 'abn::' not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print('   %-28s %16.0f per launch' % (c, v / max(calls[(k, c)], 1)))
PY
