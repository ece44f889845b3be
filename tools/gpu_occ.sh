#!/bin/bash
# encode occupancy sweep: pad the fast kernel's LDS request to force fewer blocks per CU (22320 B = 7 per CU)
for pad in 0 1000 5000 10000 19000 32000; do
  echo "pad $pad: $(FLACGPU_LDS_PAD=$pad python bench.py --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*\|encode_kernel_ms[^,]*' | tr '\n' ' ')"
done
