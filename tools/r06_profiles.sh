#!/bin/bash
# copy the summaries of a tools/r06_collect.sh run into profiles/ (tracked): bash tools/r06_profiles.sh <tag of the collect run> <letter for the profile names>
TAG=${1:-a}; L=${2:-a}
P=profiles
O=gpurun_out/r06$TAG
cp $O/gpu_tests.log $P/r06_${L}_gpu_tests.log 2>/dev/null
cp $O/smoke.log $P/r06_${L}_smoke.log 2>/dev/null
cp $O/bench_headline.json $P/r06_${L}_bench.json 2>/dev/null
cp $O/bench_2ranks.json $P/r06_${L}_bench_2ranks_one_gpu.json 2>/dev/null
for f in fuzz_gpu fuzz_api fuzz_damage fuzz_dec_stream fuzz_batch fuzz32; do [ -f $O/$f.log ] && echo "$f: $(tail -1 $O/$f.log)"; done > $P/r06_${L}_fuzz.txt
cp gpurun_out/prof_r06$TAG/trace/*kernel_stats.csv $P/r06_${L}_kernel_stats.csv 2>/dev/null
cp gpurun_out/prof_r06$TAG/summary.txt $P/r06_${L}_summary.txt 2>/dev/null
cp $O/small_calls.txt $P/r06_small_calls.txt 2>/dev/null
# configs[3]: the direct packing form against the chunk form (bench.py --workload stream24 [--no-direct], two runs each)
for f in $O/bench_stream24_*.json; do [ -f $f ] && echo "$(basename $f .json): $(python3 -c "import json,sys; d=json.load(open('$f')); print('ms_per_step', d.get('ms_per_step'), 'encode_gpu_ms', d.get('encode_gpu_ms'), 'value', d.get('value'), 'kernel_id', d.get('kernel_id'))")"; done > $P/r06_${L}_stream24_direct_vs_chunk.txt
cp $O/build_id.txt $P/r06_${L}_build_id.txt
[ -d gpurun_out/prof_r06$TAG/pmc3 ] && python3 tools/rocprof_pmc.py gpurun_out/prof_r06$TAG stream16 7032 5 r06 $O/build_id.txt > /dev/null
for spec in "stream24 7032 8" "batch 90112 5"; do
  set -- $spec
  cp gpurun_out/cfg6_$1/bench.json $P/r06_cfg_$1_bench.json 2>/dev/null
  cp gpurun_out/cfg6_$1/t_kernel_stats.csv $P/r06_cfg_$1_kernel_stats.csv 2>/dev/null
  [ -d gpurun_out/pmc6_$1/pmc3 ] && python3 tools/rocprof_pmc.py gpurun_out/pmc6_$1 $1 $2 $3 r06 $O/build_id.txt > /dev/null
done
echo "ids of the collect run (build kernel host): $(cat $O/build_id.txt); this tree: $(python3 -c 'from pyflac_amd import _lib; L = _lib.lib(); print(L.flacgpu_build_id().decode(), L.flacgpu_kernel_id().decode(), L.flacgpu_host_id().decode())')"
ls -la $P/r06_*
