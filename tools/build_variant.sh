#!/bin/bash
# Build the working tree's library as a named variant beside the real one: bash tools/build_variant.sh <name>
# -> gpurun_exp/libflacgpu_<name>.so (travels to the GPU box; select it with FLACGPU_ALLOW_LIBRARY_OVERRIDE=1 FLACGPU_LIBRARY=gpurun_exp/libflacgpu_<name>.so)
set -e
mkdir -p gpurun_exp
make -s -C pyflac_amd/csrc -j8
cp pyflac_amd/libflacgpu.so gpurun_exp/libflacgpu_$1.so
echo built gpurun_exp/libflacgpu_$1.so
