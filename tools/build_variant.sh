#!/bin/bash
# Build a variant of libslamhip.so with extra -D flags for timing ablations of one source file
# (SRC=pair_loss by default, SRC=encoder for the convolution kernels):
#   tools/build_variant.sh NAME -DSLAMHIP_ABL_NO_PHASE2 ...   -> build/variants/NAME.so
# (select it at run time with SLAMHIP_LIB=build/variants/NAME.so; ablated builds give wrong results)
set -e
name=$1; shift
src=${SRC:-pair_loss}
cd /root/repo/slam_amd/csrc
mkdir -p ../../build/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize "$@" -c $src.hip -o ../../build/variants/$name.o
objs=$(ls ../_lib/*.o | grep -v $src.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/variants/$name.so ../../build/variants/$name.o $objs
echo built build/variants/$name.so
