#!/bin/bash
# Collect PMC counters of the fused kernel, one rocprofv3 run per pass.
# usage: tools/pmc_passes.sh <outdir> [env assignments for tools/tune_fused.py ...]
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for kv in "$@"; do export "$kv"; done
export REPS=${REPS:-20}
i=0
while read -r counters; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $counters --output-format csv -d "$out/pass$i" -- python3 tools/tune_fused.py > "$out.pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$out.pass$i.log"; exit 1; }
done <<'LIST'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM
FETCH_SIZE
WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
LIST
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "pair_loss_fused" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print("%-24s n=%3d mean=%.6g" % (k, len(v), sum(v) / len(v)))
PY
