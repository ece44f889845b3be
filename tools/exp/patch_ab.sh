#!/bin/bash
# A/B of the prepared patch variants (tools/exp/patches; built as gpurun_exp/libflacgpu_<name>.so) against the tree's library on one box:
# decode tests with the variant loaded, then alternating timings.  bash tools/exp/patch_ab.sh dcp rd1
for v in "$@"; do
  echo "== $v: tests"
  FLACGPU_ALLOW_LIBRARY_OVERRIDE=1 FLACGPU_LIBRARY=$PWD/gpurun_exp/libflacgpu_$v.so timeout 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_api.py -q -x 2>&1 | tail -2
done
bash tools/exp/dec_var.sh head "$@"
for v in "$@"; do
  echo -n "$v batch: "; FLACGPU_ALLOW_LIBRARY_OVERRIDE=1 FLACGPU_LIBRARY=$PWD/gpurun_exp/libflacgpu_$v.so python3 tools/exp/batch_time.py 2>&1 | tail -1
done
echo -n "head batch: "; python3 tools/exp/batch_time.py 2>&1 | tail -1
