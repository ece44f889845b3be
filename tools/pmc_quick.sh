#!/bin/bash
# PMC passes of bench.py (two SQ passes); usage: bash tools/pmc_quick.sh <tag> [extra bench args]
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-configs --no-passes $*"
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc1 -o pmc1 -- $CMD > $OUT/bench_pmc1.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM -d $OUT/pmc2 -o pmc2 -- $CMD > $OUT/bench_pmc2.log 2>&1
python3 tools/rocprof_summary.py $OUT > $OUT/summary.txt 2>&1
