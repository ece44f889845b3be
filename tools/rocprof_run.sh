#!/bin/bash
# Profile bench.py's GPU work with rocprofv3: kernel trace + stats, then PMC passes (separate runs).
# Usage (on the GPU box): bash tools/rocprof_run.sh <tag> [seconds] [level]
set -u
TAG=${1:-r1}
SECS=${2:-120}
LEVEL=${3:-5}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
WORK=${4:-stream16}
EXTRA=${5:-}
CMD="python3 bench.py --workload $WORK --seconds $SECS --steps 5 --warmup 1 --level $LEVEL --no-cpu-baseline --no-e2e --no-configs $EXTRA"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/bench_trace.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc1 -o pmc1 -- $CMD > $OUT/bench_pmc1.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM -d $OUT/pmc2 -o pmc2 -- $CMD > $OUT/bench_pmc2.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc3 -o pmc3 -- $CMD > $OUT/bench_pmc3.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc4 -o pmc4 -- $CMD > $OUT/bench_pmc4.log 2>&1
find $OUT -name "*.csv" | head -40
python3 tools/rocprof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
