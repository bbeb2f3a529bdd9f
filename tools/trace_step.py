"""Kernel sequence of ONE train step (between two optimizer launches) in a rocprofv3
kernel trace, with start offsets, durations and the idle gap before each kernel.
  python tools/trace_step.py <dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
opt = [i for i, r in enumerate(rows) if 'optimizer_kernel' in r['Kernel_Name']]
a, b = opt[len(opt) // 2], opt[len(opt) // 2 + 1]
t0 = int(rows[a]['End_Timestamp'])
prev = t0
busy = 0
for r in rows[a + 1:b + 1]:
    n = r['Kernel_Name']
    n = n.replace('void ', '')[:60]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    busy += e - s
    print('%-62s +%7.1f us  dur %6.1f  gap %5.1f' % (n, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3))
    prev = e
print('step %.1f us, kernels busy %.1f us, gaps %.1f us, %d kernels' % ((prev - t0) / 1e3, busy / 1e3, (prev - t0 - busy) / 1e3, b - a))
