#!/bin/bash
# Round-4 evidence run (GPU box): the driver-shaped bench line (headline + configs[3] + configs[4]), rocprofv3 kernel stats + PMC
# passes of the headline command, kernel stats and PMC passes of the other configurations, the shapes outside the headline.
# Everything lands under gpurun_out/r04<tag>/ and gpurun_out/{prof_r04<tag>,cfg_*,pmc_cfg_*}; tools/r04_profiles.sh copies the
# summaries into profiles/.
export TMPDIR=/tmp
TAG=${1:-a}
OUT=gpurun_out/r04$TAG
mkdir -p $OUT
timeout 1200 python3 bench.py > $OUT/bench_headline.log 2> $OUT/bench_headline.err; grep -v "^[WEI]2026" $OUT/bench_headline.log | tail -1 > $OUT/bench_headline.json
bash tools/rocprof_run.sh r04$TAG 600 5 > $OUT/rocprof_run.log 2>&1
for spec in "stream24 300 8" "batch 60 5" "wasted 600 5" "stream32 600 5"; do
  set -- $spec
  O2=gpurun_out/cfg_$1
  mkdir -p $O2
  rocprofv3 --output-format csv --kernel-trace --stats -d $O2 -o t -- python3 bench.py --workload $1 --seconds $2 --level $3 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-configs > $O2/bench.log 2>&1
  grep -v "^[WEI]2026" $O2/bench.log | tail -1 > $O2/bench.json
  O3=$PWD/gpurun_out/pmc_cfg_$1
  mkdir -p $O3
  CMD="python3 bench.py --workload $1 --seconds $2 --level $3 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-configs"
  rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES -d $O3/pmc1 -o pmc1 -- $CMD > $O3/bench_pmc1.log 2>&1
  rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O3/pmc3 -o pmc3 -- $CMD > $O3/bench_pmc3.log 2>&1
  rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $O3/pmc4 -o pmc4 -- $CMD > $O3/bench_pmc4.log 2>&1
done
run() { timeout 900 python3 bench.py "$@" --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-configs 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$*: value %.1f ms/step %.3f enc_gpu %.3f dec_gpu %.3f' % (d['value'], d['ms_per_step'], d.get('encode_gpu_ms',0), d.get('decode_gpu_ms',0)))"; }
(run --workload stream16
run --workload stream16 --seconds 600.0162
run --workload stream32 --seconds 120
run --workload stream32 --seconds 600
run --workload stream32w --seconds 300
run --workload surround6 --seconds 120
run --workload wasted) > $OUT/other_shapes.txt 2>&1
cat $OUT/other_shapes.txt
python3 tools/exp/small_call_probe.py > $OUT/small_calls.txt 2>&1
tail -c 1200 $OUT/bench_headline.json
