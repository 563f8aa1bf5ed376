#!/bin/bash
# Tuning sweep of the fused kernel's tile shapes and of the shared-anchor form over the bench workloads
# (needs build/variants/tiles.so: tools/build_variant.sh tiles -DSLAMHIP_TUNING_TILES).
for cfg in "--size 960x1280 --neighbours 4 --batch 16" "--neighbours 2 --batch 32" "--neighbours 4 --batch 16"; do
  for v in "SLAMHIP_TILE=24x12" "SLAMHIP_TILE=16x8" "SLAMHIP_SHARED_ANCHORS=1"; do
    env $v SLAMHIP_LIB=build/variants/tiles.so timeout -k 10 200 python bench.py --steps 300 --warmup 50 --cpu-seconds 2 --encoder-steps 0 --train-steps 0 $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg | $v |', round(d['value']), 'pairs/s', round(d['roofline']['launch_ms']*1e3,1), 'us', d['verified_against_oracle'])"
  done
done
