#!/bin/bash
# Round-5 evidence run (GPU box): the driver-shaped bench line, rocprofv3 kernel stats + PMC passes of the headline command, kernel
# stats and PMC passes of configs[3] / configs[4], the small-call probe.  Everything lands under gpurun_out/r05<tag>/ and
# gpurun_out/{prof_r05<tag>,cfg5_*,pmc5_*}; tools/r05_profiles.sh copies the summaries into profiles/ (run it on the build box).
export TMPDIR=/tmp
TAG=${1:-a}
OUT=gpurun_out/r05$TAG
mkdir -p $OUT
timeout 1500 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_headline.log 2> $OUT/bench_headline.err; grep -v "^[WEI]2026" $OUT/bench_headline.log | tail -1 > $OUT/bench_headline.json
bash tools/rocprof_run.sh r05$TAG 600 5 stream16 --no-passes > $OUT/rocprof_run.log 2>&1      # (no event passes: they run the stages one by one, i.e. unfused)
for spec in "stream24 300 8" "batch 60 5"; do
  set -- $spec
  O2=$PWD/gpurun_out/cfg5_$1
  mkdir -p $O2
  rocprofv3 --output-format csv --kernel-trace --stats -d $O2 -o t -- python3 bench.py --workload $1 --seconds $2 --level $3 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-configs --no-passes > $O2/bench.log 2>&1
  grep -v "^[WEI]2026" $O2/bench.log | tail -1 > $O2/bench.json
  O3=$PWD/gpurun_out/pmc5_$1
  mkdir -p $O3
  CMD="python3 bench.py --workload $1 --seconds $2 --level $3 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-configs --no-passes"
  rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES -d $O3/pmc1 -o pmc1 -- $CMD > $O3/bench_pmc1.log 2>&1
  rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O3/pmc3 -o pmc3 -- $CMD > $O3/bench_pmc3.log 2>&1
  rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $O3/pmc4 -o pmc4 -- $CMD > $O3/bench_pmc4.log 2>&1
done
(python3 tools/exp/process_call_probe.py 4096; python3 tools/exp/process_call_probe.py 1024; python3 tools/exp/process_call_probe.py 16384; python3 tools/exp/small_call_probe.py) > $OUT/small_calls.txt 2>&1
python3 -c "from pyflac_amd import _lib; print(_lib.lib().flacgpu_build_id().decode())" > $OUT/build_id.txt 2>/dev/null
tail -c 800 $OUT/bench_headline.json
