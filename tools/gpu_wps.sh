#!/bin/bash
# decode launch-shape experiment: waves per SIMD targeted by the frames-per-wave heuristic
for w in 1 2 3 4 6; do
  echo "WPS $w: $(FLACGPU_DEC_WPS=$w FLACGPU_DEC_PROF=1 python bench.py --seconds 600 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -o 'kernel 0.*\|decode_kernel_ms[^,]*' | tail -3 | tr '\n' ' ')"
done
