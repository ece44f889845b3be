#!/bin/bash
# per-kernel averages of the headline loop for several library variants, one group (kernels do not overlap): bash tools/kstat_var.sh v1 v2 ...
for v in "$@"; do
  echo "== $v"
  if [ "$v" = "head" ]; then unset FLACGPU_LIBRARY; else export FLACGPU_ALLOW_LIBRARY_OVERRIDE=1 FLACGPU_LIBRARY=$PWD/gpurun_exp/libflacgpu_$v.so; fi
  FLACGPU_GROUPS=1 bash tools/kstat.sh $v 2>&1 | grep -i "pack\|eval\|autoc\|levin\|assem\|scan_sizes"
  FLACGPU_GROUPS=1 python3 tools/exp/direct_time.py 600 40 1 | grep -v amdgpu
  python3 tools/exp/direct_time.py 600 40 1 | grep -v amdgpu
done
