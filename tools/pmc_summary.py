#!/usr/bin/env python3
"""Per-kernel mean of rocprofv3 --pmc counters (counter_collection.csv)."""
import csv, glob, sys, collections
pat = sys.argv[2] if len(sys.argv) > 2 else ''
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        if pat in r['Kernel_Name']:
            acc[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print('   %-32s n=%4d mean=%.4g' % (c, len(v), sum(v) / len(v)))
