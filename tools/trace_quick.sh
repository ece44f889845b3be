#!/bin/bash
# kernel trace of a few bench steps (timed loop only: --no-passes skips the event passes), then the timeline of the last steps
# usage: bash tools/trace_quick.sh <tag> [extra bench args]
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/tq_$TAG
mkdir -p $OUT
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT -o t -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-e2e --no-passes $* > $OUT/bench.log 2>&1
python3 tools/trace_timeline.py $OUT/t_kernel_trace.csv 36 > $OUT/timeline.txt
cat $OUT/timeline.txt
