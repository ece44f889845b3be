#!/bin/bash
# decode launch-shape sweep: frames per wave of the parse kernel (G1) and of the restore kernel (G2)
for g in ${G1LIST:-12 14 16 20}; do
  echo "G1 $g: $(FLACGPU_DEC_G1=$g python bench.py --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*\|decode_kernel_ms[^,]*' | tr '\n' ' ')"
done
for g in ${G2LIST:-4 5 6 7 8 10 14 16}; do
  echo "G2 $g: $(FLACGPU_DEC_G2=$g python bench.py --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*\|decode_kernel_ms[^,]*' | tr '\n' ' ')"
done
