#!/bin/bash
# The audit build of the library (-DSLAMHIP_AUDIT on pair_loss.hip and overlap_fused.hip): build/variants/audit.so
set -e
cd "$(dirname "$0")/../slam_amd/csrc"
make -s
mkdir -p ../../build/variants
for f in pair_loss overlap_fused; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -DSLAMHIP_AUDIT -c $f.hip -o ../../build/variants/audit_$f.o
done
objs=$(ls ../_lib/*.o | grep -v "pair_loss.o\|overlap_fused.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/variants/audit.so ../../build/variants/audit_pair_loss.o ../../build/variants/audit_overlap_fused.o $objs
echo built build/variants/audit.so
