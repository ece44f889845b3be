"""Kernel timeline of a rocprofv3 --kernel-trace run: start (relative), duration and the idle gap before every kernel of
the last `n` dispatches -- shows where the GPU waits for the host.  usage: python tools/trace_timeline.py <kernel_trace.csv> [n] [skip]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[:len(rows) - skip] if skip else rows
t0 = prev = None
for r in rows[-n:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if t0 is None:
        t0 = s
    name = r['Kernel_Name'].split('(anonymous namespace)::')[-1].split('(')[0][:44]
    print('%9.1f %8.1f  gap %7.1f  q%s  %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, r.get('Queue_Id', '?'), name))
    prev = max(e, prev or 0)
