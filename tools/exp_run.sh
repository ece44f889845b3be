#!/bin/bash
# On the GPU box: run a command once per experiment library (gpurun_exp/libflacgpu_x*.so copied over the product library of
# this scratch copy).  usage: bash tools/exp_run.sh "<command>"
cp pyflac_amd/libflacgpu.so /tmp/libflacgpu_base.so
for f in gpurun_exp/libflacgpu_x*.so; do
  cp $f pyflac_amd/libflacgpu.so
  echo "== $f"
  bash -c "$1"
done
cp /tmp/libflacgpu_base.so pyflac_amd/libflacgpu.so
echo "== base"
bash -c "$1"
