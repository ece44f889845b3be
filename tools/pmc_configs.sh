#!/bin/bash
# PMC passes (VALU instructions, FETCH_SIZE, WRITE_SIZE; separate runs) of the configurations other than the headline, for
# profiles/<round>_pmc_<workload>.json (tools/rocprof_pmc.py).  GPU box.
export TMPDIR=/tmp
for spec in "stream24 300 8" "batch 60 5" "wasted 600 5"; do
  set -- $spec
  OUT=$PWD/gpurun_out/pmc_cfg_$1
  mkdir -p $OUT
  CMD="python3 bench.py --workload $1 --seconds $2 --level $3 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e"
  rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES -d $OUT/pmc1 -o pmc1 -- $CMD > $OUT/bench_pmc1.log 2>&1
  rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc3 -o pmc3 -- $CMD > $OUT/bench_pmc3.log 2>&1
  rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc4 -o pmc4 -- $CMD > $OUT/bench_pmc4.log 2>&1
  grep -v "^[WEI]2026" $OUT/bench_pmc1.log | tail -1 | cut -c1-200
done
