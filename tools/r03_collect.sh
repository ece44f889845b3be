#!/bin/bash
# Round-3 evidence run (GPU box): the headline bench line, rocprofv3 kernel stats + PMC passes of the same command, the other
# configurations with kernel stats, the shapes outside the headline.  Everything lands under gpurun_out/r03<tag>/.
export TMPDIR=/tmp
TAG=${1:-a}
OUT=gpurun_out/r03$TAG
mkdir -p $OUT
timeout 900 python3 bench.py > $OUT/bench_headline.log 2> $OUT/bench_headline.err; grep -v "^[WEI]2026" $OUT/bench_headline.log | tail -1 > $OUT/bench_headline.json
bash tools/rocprof_run.sh r03$TAG 600 5 > $OUT/rocprof_run.log 2>&1
bash tools/prof_configs.sh > $OUT/prof_configs.log 2>&1
bash tools/cliffs.sh > $OUT/cliffs.txt 2>&1
cat $OUT/cliffs.txt
tail -c 1500 $OUT/bench_headline.json
