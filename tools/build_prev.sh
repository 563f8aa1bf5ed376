#!/bin/bash
# build/variants/NAME.so from the version of one source file at a git revision (default HEAD), linked with the
# in-tree objects of the other files: the baseline of an A/B timing (tools/ab_layers.sh, tools/ab_variants.sh)
#   tools/build_prev.sh NAME [SRC=encoder] [REV=HEAD]
set -e
name=$1; src=${2:-encoder}; rev=${3:-HEAD}
cd "$(dirname "$0")/../slam_amd/csrc"
mkdir -p ../../build/variants
git show $rev:slam_amd/csrc/$src.hip > _prev_$src.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -c _prev_$src.hip -o ../../build/variants/$name.o
rm -f _prev_$src.hip
objs=$(ls ../_lib/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/variants/$name.so ../../build/variants/$name.o $objs
echo built build/variants/$name.so
