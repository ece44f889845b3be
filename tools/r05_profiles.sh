#!/bin/bash
# copy the summaries of a tools/r05_collect.sh run into profiles/ (tracked): bash tools/r05_profiles.sh <tag of the collect run> <letter for the profile names>
TAG=${1:-a}; L=${2:-a}
P=profiles
cp gpurun_out/r05$TAG/bench_headline.json $P/r05_${L}_bench.json
cp gpurun_out/prof_r05$TAG/trace/*kernel_stats.csv $P/r05_${L}_kernel_stats.csv
cp gpurun_out/prof_r05$TAG/summary.txt $P/r05_${L}_summary.txt
cp gpurun_out/r05$TAG/small_calls.txt $P/r05_small_calls.txt
python3 tools/rocprof_pmc.py gpurun_out/prof_r05$TAG stream16 7032 5 r05 gpurun_out/r05$TAG/build_id.txt > /dev/null
for spec in "stream24 7032 8" "batch 90112 5"; do
  set -- $spec
  cp gpurun_out/cfg5_$1/bench.json $P/r05_cfg_$1_bench.json
  cp gpurun_out/cfg5_$1/t_kernel_stats.csv $P/r05_cfg_$1_kernel_stats.csv
  python3 tools/rocprof_pmc.py gpurun_out/pmc5_$1 $1 $2 $3 r05 gpurun_out/r05$TAG/build_id.txt > /dev/null
done
# the build the passes ran on (the collect run recorded it; the library here must be the same sources)
echo "build id of the collect run: $(cat gpurun_out/r05$TAG/build_id.txt); this tree: $(python3 -c 'from pyflac_amd import _lib; print(_lib.lib().flacgpu_build_id().decode())')"
ls -la $P/r05_*
