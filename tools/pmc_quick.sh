#!/bin/bash
# one PMC pass (VALU / SALU / LDS instruction counts, busy) + timing for a library variant
set -u
out=$1; lib=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export REPS=20 SLAMHIP_LIB=$lib
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$out" -- python3 tools/tune_fused.py > "$out.log" 2>&1 || { echo failed; tail -3 "$out.log"; exit 1; }
grep "us/launch" "$out.log"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "pair_loss_fused" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("  ".join("%s=%.4g" % (k, sum(v) / len(v)) for k, v in sorted(acc.items())))
PY
