#!/bin/bash
# kernel-trace + stats of the BASELINE configurations other than the headline (configs[3], [4], wasted-bits variant)
export TMPDIR=/tmp
for spec in "stream24 300 8" "batch 60 5" "wasted 600 5"; do
  set -- $spec
  OUT=gpurun_out/cfg_$1
  mkdir -p $OUT
  rocprofv3 --output-format csv --kernel-trace --stats -d $OUT -o t -- python3 bench.py --workload $1 --seconds $2 --level $3 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > $OUT/bench.log 2>&1
  grep -v "^[WEI]2026" $OUT/bench.log | tail -1 > $OUT/bench.json
done
