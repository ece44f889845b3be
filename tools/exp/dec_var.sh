#!/bin/bash
# decode launch of the headline stream for several library variants, alternating: bash tools/exp/dec_var.sh head v1 v2 ...
for i in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = "head" ]; then unset FLACGPU_LIBRARY; else export FLACGPU_ALLOW_LIBRARY_OVERRIDE=1 FLACGPU_LIBRARY=$PWD/gpurun_exp/libflacgpu_$v.so; fi
    echo -n "$v  "; python3 tools/exp/dec_time.py 600 30 2>/dev/null
  done
done
