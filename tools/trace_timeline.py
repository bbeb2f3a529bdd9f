"""Kernel timeline of the LAST batched DTW call in a rocprofv3 kernel trace.
  python tools/trace_timeline.py <dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'expand_tiles' in r['Kernel_Name']]
last = rows[starts[-1]:]
t0 = int(last[0]['Start_Timestamp'])
for r in last:
    n = r['Kernel_Name']
    n = n[n.index('abn::') + 5:][:22] if 'abn::' in n else n[:22]
    print('%-24s start %7.0f us  end %7.0f us  (%6.0f us)  queue %s' % (
        n, (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3,
        (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Queue_Id', '?')))
