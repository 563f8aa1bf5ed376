for cfg in "--size 960x1280 --neighbours 4 --batch 16" "--neighbours 2 --batch 32" "--neighbours 4 --batch 16"; do
  for t in 32x16 24x12; do
    SLAMHIP_TILE=$t SLAMHIP_LIB=build/variants/tiles.so timeout -k 10 200 python bench.py --steps 300 --warmup 50 --cpu-seconds 2 --encoder-steps 0 --train-steps 0 $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg | $t |', round(d['value']), 'pairs/s', round(d['roofline']['launch_ms']*1e3,1), 'us', d['verified_against_oracle'])"
  done
done
