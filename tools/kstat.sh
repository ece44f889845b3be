#!/bin/bash
# kernel-trace stats of the headline bench loop: bash tools/kstat.sh <tag> [extra bench args]; prints name, calls, average ns
export TMPDIR=/tmp
T=$1; shift
OUT=$PWD/gpurun_out/kstat_$T
mkdir -p $OUT
cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-configs --no-passes "$@" > $OUT/log.txt 2>&1
cd $GRAFT_REPO_ROOT
python3 - "$OUT" <<'P'
import csv, glob, sys, os
for f in glob.glob(os.path.join(sys.argv[1], '**', '*kernel_stats.csv'), recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = 0
    for r in rows:
        n = r['Name']
        if 'fg_' not in n: continue
        short = n.split('(')[0].replace('void ', '').replace('(anonymous namespace)::', '')
        if '<' in n: short = n[:n.index('>') + 1].replace('void ', '').replace('(anonymous namespace)::', '')
        print('%-70s calls %4s  avg %9.1f us' % (short[:70], r['Calls'], float(r['AverageNs']) / 1000))
P
