"""Timeline of the kernels of ONE encode launch (the last but one of a rocprofv3 --kernel-trace run of bench.py): start and end of
every kernel relative to the launch's first kernel.  usage: python tools/exp/enc_timeline.py <trace dir>"""
import csv, sys, glob
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:52]))
rows.sort()
starts = [i for i, r in enumerate(rows) if 'fg_pipe_begin' in r[2]]
i0 = starts[-2] if len(starts) > 1 else starts[-1]
t0 = rows[i0][0]
for a, b, n in rows[i0:]:
    if 'fg_pipe_begin' in n and a != t0: break
    if 'fg_dec' in n: break
    print('%-54s %8.1f .. %8.1f us' % (n, (a - t0) / 1e3, (b - t0) / 1e3))
