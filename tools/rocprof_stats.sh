#!/bin/bash
# kernel-trace + stats of the bench workload; usage: bash tools/rocprof_stats.sh <tag> [seconds]
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/stats_$1
mkdir -p $OUT
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT -o p -- python3 bench.py --seconds ${2:-600} --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-configs --no-passes > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt | cut -c1-400
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'EOF'
import csv, sys
for i, row in enumerate(csv.reader(open(sys.argv[1]))):
    if i == 0 or 'fg_' in row[0]:
        name = row[0].replace('(anonymous namespace)::', '')
        print('%-60s calls %-4s avg_ns %-12s pct %s' % (name[:60], row[1], row[3], row[4]))
EOF
