#!/bin/bash
# Per-kernel time table of the encoder training step (batch 128, eager launches): rocprofv3 --kernel-trace --stats over
# tools/bench_encoder.py; usage (on the GPU box): bash tools/trace_encoder.sh gpurun_out/NAME  -> NAME.stats.csv, NAME.top.txt
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export ENC_GRAPH=0 ENC_FWD=0 ENC_STEPS=${ENC_STEPS:-5}
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out.d" -- python3 tools/bench_encoder.py > "$out.log" 2>&1 || { echo "trace run failed"; tail -5 "$out.log"; exit 1; }
f=$(find "$out.d" -name '*kernel_stats.csv' | head -1)
cp "$f" "$out.stats.csv"
python3 - "$out.stats.csv" $((ENC_STEPS + 2)) > "$out.top.txt" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])      # 2 warm-up steps + the timed ones
tot = 0.0
out = []
for r in rows:
    name = re.sub(r"\(.*", "", r["Name"])[:78]
    ns, n = float(r["TotalDurationNs"]), int(r["Calls"])
    tot += ns
    out.append((ns, n, name))
print("kernel time per step: %.3f ms (%d steps)" % (tot / steps / 1e6, steps))
for ns, n, name in sorted(out, reverse=True)[:40]:
    print("%-78s n/step=%6.1f us/step=%9.1f avg_us=%8.1f" % (name, n / steps, ns / steps / 1e3, ns / n / 1e3))
PY
cat "$out.top.txt"
