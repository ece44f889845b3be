#!/bin/bash
# Build experiment variants of one translation unit: tools/exp_build.sh <file.hip> <n1> <n2> ...  ->  gpurun_exp/libflacgpu_x<n>.so
# (compiled with -DFG_EXP=<n>; the other objects are the current build's).  Run them on the GPU box with tools/exp_run.sh.
set -e
cd "$(dirname "$0")/../pyflac_amd/csrc"
SRC=$1; shift
mkdir -p ../../gpurun_exp
OBJS="flac_enc_kernels.o flac_enc_fast.o fast_ms_o8.o fast_ms_o12.o fast_st_o8.o fast_st_o12.o fast_mono_o8.o fast_mono_o12.o pipe_ms_o8.o pipe_ms_o12.o pipe_st_o8.o pipe_st_o12.o pipe_mono_o8.o pipe_mono_o12.o flac_enc_pipe.o flac_dec_kernels.o flac_dec_fast.o fg_ctx.o flacgpu_enc_api.o flacgpu_dec_api.o"
for n in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -DFG_EXP=$n -c $SRC -o /tmp/exp_$n.o
    L=$(echo $OBJS | sed "s#${SRC%.hip}.o#/tmp/exp_$n.o#")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_exp/libflacgpu_x$n.so $L -lpthread ) &
done
wait
ls -la ../../gpurun_exp
