#!/bin/bash
# one PMC pass on the bench workload; usage: bash tools/rocprof_quick.sh <tag> "<counters>" [seconds]
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/quick_$1
mkdir -p $OUT
rocprofv3 --output-format csv --kernel-trace --pmc $2 -d $OUT -o p -- python3 bench.py --seconds ${3:-600} --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-configs --no-passes > $OUT/log.txt 2>&1
python3 tools/rocprof_summary.py $OUT | grep -A14 "fg_encode_fast\|fg_dec_rice\|fg_dec_restore" | cut -c1-110
