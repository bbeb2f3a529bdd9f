"""Per-call GPU timeline from a rocprofv3 kernel trace: for each dist_kernel launch,
the span to the end of the last dp_kernel that follows it.
  python tools/trace_span.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
calls = []
for r in rows:
    n = r['Kernel_Name']
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if 'dist_kernel' in n:
        calls.append({'start': s, 'dist_end': e, 'end': e, 'dp': []})
    elif 'dp_kernel' in n and calls:
        calls[-1]['end'] = max(calls[-1]['end'], e)
        calls[-1]['dp'].append((n[n.index('<'):n.index('>') + 1], (s - calls[-1]['start']) / 1e3, (e - calls[-1]['start']) / 1e3))
for c in calls[-2:]:
    print('dist %.0f us, dp phase %.0f us, total GPU span %.0f us' % ((c['dist_end'] - c['start']) / 1e3, (c['end'] - c['dist_end']) / 1e3, (c['end'] - c['start']) / 1e3))
    for d in c['dp']:
        print('   dp%s  start %.0f  end %.0f' % d)
