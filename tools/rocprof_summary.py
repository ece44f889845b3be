"""Summarise rocprofv3 output directories produced by tools/rocprof_run.sh into a small text report."""
import csv, glob, os, sys
from collections import defaultdict

root = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(root, '**', pattern), recursive=True))


print('== kernel stats (rocprofv3 --kernel-trace --stats) ==')
for f in find('*kernel_stats.csv'):
    with open(f) as fh:
        for i, row in enumerate(csv.reader(fh)):
            if i < 12:
                print(','.join(row[:9]))
print()
print('== per-kernel counter means (rocprofv3 --pmc; one dispatch = one launch) ==')
for f in find('*counter_collection.csv'):
    acc = defaultdict(lambda: defaultdict(list))
    with open(f) as fh:
        rd = csv.DictReader(fh)
        for row in rd:
            k = row.get('Kernel_Name', '?')
            acc[k][row.get('Counter_Name', '?')].append(float(row.get('Counter_Value', 0)))
    print(os.path.relpath(f, root))
    for k, d in acc.items():
        if 'fg_' not in k:
            continue
        print('  ', k[:90])
        for c, v in sorted(d.items()):
            print('      %-24s n=%-4d mean=%.4g' % (c, len(v), sum(v) / len(v)))
