#!/bin/bash
# HBM-side traffic of the fused pair-loss kernel: FETCH_SIZE and WRITE_SIZE in separate
# rocprofv3 --pmc runs (a TCC pass holds one of them), mean per launch, written as JSON.
# usage (on the GPU box): bash tools/pmc_traffic.sh gpurun_out/traffic
set -u
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export REPS=20
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d "$out/$c" -- python3 tools/tune_fused.py > "$out.$c.log" 2>&1 || { echo "pass $c failed"; tail -5 "$out.$c.log"; exit 1; }
done
python3 - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "pair_loss_fused" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
fetch_kb = sum(acc["FETCH_SIZE"]) / len(acc["FETCH_SIZE"])
write_kb = sum(acc["WRITE_SIZE"]) / len(acc["WRITE_SIZE"])
res = {
    "kernel": "pair_loss_fused_kernel", "workload": "64 pairs 640x480 per launch, 4 rotated HBM batches",
    "launches_averaged": len(acc["FETCH_SIZE"]),
    "FETCH_SIZE_kb_raw": fetch_kb, "WRITE_SIZE_kb_raw": write_kb,
    "correction": "gfx950: FETCH_SIZE counts 128-byte read requests as 64 bytes for wide coalesced reads -> x2; WRITE_SIZE exact",
    "hbm_read_bytes_per_launch": 2.0 * fetch_kb * 1024.0, "hbm_write_bytes_per_launch": write_kb * 1024.0,
}
res["hbm_bytes_per_launch"] = res["hbm_read_bytes_per_launch"] + res["hbm_write_bytes_per_launch"]
json.dump(res, open(out + ".json", "w"), indent=1)
print(json.dumps(res))
PY
