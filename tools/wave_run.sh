#!/bin/bash
# GPU run of the wave-parallel decoder: parity on the decode tests, then timing (optionally with experiment switches)
export TMPDIR=/tmp
mkdir -p gpurun_out/wave
echo "== wave_check (WAVE=2)"; FLACGPU_DEC_WAVE=2 timeout 300 python3 tools/wave_check.py 600 3 2>&1 | tail -6
echo "== decode tests with FLACGPU_DEC_WAVE=1"
FLACGPU_DEC_WAVE=1 timeout 900 python3 -m pytest tests/test_gpu_decode.py -x -q -m gpu 2>&1 | tail -4
export FLACGPU_DEC_WAVE=1
for skip in ${SKIPS:-0}; do
  echo "== kernel stats, wave path, FLACGPU_DEC_SKIP=$skip"
  rm -rf /tmp/dks; FLACGPU_DEC_SKIP=$skip timeout 200 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/dks -o x -- python3 tools/wave_check.py 600 6 > /dev/null 2>&1
  grep -h "fg_dec_w\|fg_dec_crc" /tmp/dks/*kernel_stats.csv | awk -F'",' '{print substr($1,1,60), $2, $4}' | tr -d '"'
  [ "$skip" = "0" ] && cp /tmp/dks/*kernel_stats.csv gpurun_out/wave/wave_kernel_stats.csv 2>/dev/null
done
if [ -n "$PMC" ]; then
  for pass in 1 2; do
    if [ $pass = 1 ]; then CTR="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
    else CTR="SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM"; fi
    rm -rf gpurun_out/wave/pmc$pass; timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc $CTR -d gpurun_out/wave/pmc$pass -o pmc$pass -- python3 tools/wave_check.py 600 4 > /dev/null 2>&1
  done
  python3 tools/rocprof_summary.py gpurun_out/wave 2>/dev/null | grep -A9 "fg_dec_wparse\|fg_dec_wrestore" | cut -c1-100
fi
