#!/bin/bash
# Build a named variant of the library with extra compiler flags, out of tree: bash tools/build_variant2.sh <name> "<flags>" [TUNING=1]
# -> gpurun_exp/libflacgpu_<name>.so (select it with FLACGPU_ALLOW_LIBRARY_OVERRIDE=1 FLACGPU_LIBRARY=$PWD/gpurun_exp/libflacgpu_<name>.so)
set -e
N=$1; F=$2; shift; shift
D=/tmp/fgvar_$N
mkdir -p $D /root/repo/gpurun_exp
cp /root/repo/pyflac_amd/csrc/*.hip /root/repo/pyflac_amd/csrc/*.cpp /root/repo/pyflac_amd/csrc/*.h /root/repo/pyflac_amd/csrc/*.inc /root/repo/pyflac_amd/csrc/Makefile $D/
cd $D
sed -i "s#OUT = ../libflacgpu.so#OUT = /root/repo/gpurun_exp/libflacgpu_$N.so#; s#-I../../include#-I/root/repo/include $F#; s#\.\./\.\./include/flacgpu.h#/root/repo/include/flacgpu.h#" Makefile
make -s -j8 "$@" 2>&1 | grep -E 'error|Error' || true
ls -la /root/repo/gpurun_exp/libflacgpu_$N.so
