#!/usr/bin/env python3
"""Prints a compact per-kernel table from a rocprofv3 --stats kernel_stats.csv."""
import csv, glob, sys
for path in sorted(glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)):
    rows = list(csv.DictReader(open(path)))
    print(path)
    for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
        print('  %-95s calls=%5s avg_us=%8.2f min_us=%8.2f pct=%s' % (r['Name'][:95], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, r['Percentage']))
