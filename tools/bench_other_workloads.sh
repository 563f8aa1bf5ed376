#!/bin/bash
# The frame-pair kernel on inputs other than the default workload, one bench.py line each (every run is checked
# against the oracle): bash tools/bench_other_workloads.sh OUT.jsonl
out=${1:-gpurun_out/other_workloads.jsonl}
: > "$out"
common="--steps 300 --warmup 50 --cpu-seconds 2 --encoder-steps 0 --train-steps 0"
for flags in "--depth noise" "--factor 5000" "--factor 5000 --depth noise" "--neighbours 2 --batch 32" \
             "--neighbours 4 --batch 16" "--size 960x1280 --neighbours 4 --batch 16" "--batch 256"; do
  timeout -k 10 300 python bench.py $common $flags >> "$out" 2>> "$out.err" || { echo "failed: $flags"; exit 1; }
done
python - "$out" <<'P'
import json, sys
for line in open(sys.argv[1]):
    r = json.loads(line)
    print("%-110s %9.0f pairs/s  %7.1f us  frac %.3f" % (r["config"]["workload"][-110:], r["value"], r["ms_per_step"] * 1e3, r["roofline"]["frac"]))
P
