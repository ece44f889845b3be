#!/bin/bash
# copy the summaries of a tools/r04_collect.sh run into profiles/ (tracked): bash tools/r04_profiles.sh <tag of the collect run> <letter for the profile names>
TAG=${1:-a}; L=${2:-b}
P=profiles
cp gpurun_out/r04$TAG/bench_headline.json $P/r04_${L}_bench.json
cp gpurun_out/prof_r04$TAG/trace/*kernel_stats.csv $P/r04_${L}_kernel_stats.csv
cp gpurun_out/prof_r04$TAG/summary.txt $P/r04_${L}_summary.txt
cp gpurun_out/r04$TAG/other_shapes.txt $P/r04_${L}_other_shapes.txt
cp gpurun_out/r04$TAG/small_calls.txt $P/r04_${L}_small_calls.txt
python3 tools/rocprof_pmc.py gpurun_out/prof_r04$TAG stream16 7032 5 r04 > /dev/null
for spec in "stream24 7032 8" "batch 90112 5" "wasted 7032 5" "stream32 7032 5"; do
  set -- $spec
  cp gpurun_out/cfg_$1/bench.json $P/r04_cfg_$1_bench.json
  cp gpurun_out/cfg_$1/t_kernel_stats.csv $P/r04_cfg_$1_kernel_stats.csv
  python3 tools/rocprof_pmc.py gpurun_out/pmc_cfg_$1 $1 $2 $3 r04 > /dev/null
done
ls -la $P/r04_*
