#!/bin/bash
# Register / spill summary and assembly of one HIP source of the library (cross-compiled, no GPU needed):
#   tools/kres.sh pair_loss      -> /tmp/asm/pair_loss.s + one line per kernel
f=${1:-pair_loss}
mkdir -p /tmp/asm
cd /root/repo/slam_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -S --cuda-device-only $f.hip -o /tmp/asm/$f.s -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|ScratchSize|Spill|Occupancy" | paste - - - - - - | sed "s/$f.hip:[0-9]*:[0-9]*: remark://g; s/\[-Rpass-analysis=kernel-resource-usage\]//g; s/  */ /g"
