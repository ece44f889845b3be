import csv, sys, glob
rows = []
for f in glob.glob(sys.argv[1] + '/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].split('<')[0][:40]))
rows.sort()
rows = rows[-7 * 20:]          # the last 20 calls
from collections import defaultdict
dur = defaultdict(list); gap = defaultdict(list)
for i, (a, b, n) in enumerate(rows):
    dur[n].append(b - a)
    if i: gap[n].append(a - rows[i - 1][1])
for n in dur:
    print('%-42s dur %7.1f us   gap before %7.1f us' % (n, sum(dur[n]) / len(dur[n]) / 1e3, (sum(gap[n]) / max(1, len(gap[n]))) / 1e3))
