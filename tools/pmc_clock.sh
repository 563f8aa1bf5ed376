#!/bin/bash
# Effective shader clock of the convolution kernels: GRBM_GUI_ACTIVE (cycles) per dispatch
# against the dispatch duration of the same run.  Usage: WHICH=dgrad CONV=... bash tools/pmc_clock.sh <outdir>
set -u
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export REPS=5
timeout -s KILL 90 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$out" -- python3 tools/bench_conv.py > "$out.log" 2>&1 || { echo "pass failed"; tail -3 "$out.log"; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, sys, collections
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "igemm" in k or "wgrad" in k:
            cnt[k[:50]][row["Counter_Name"]].append(float(row["Counter_Value"]))
            if "Start_Timestamp" in row:
                cnt[k[:50]]["dur_ns_" + row["Counter_Name"]].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
dur = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "igemm" in k or "wgrad" in k:
            dur[k[:50]].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
for k, d in cnt.items():
    mean = {c: sum(v) / len(v) for c, v in d.items()}
    t = sum(dur[k]) / max(1, len(dur[k]))
    print(k, " ".join("%s=%.4g" % kv for kv in sorted(mean.items())), "dur_us=%.1f" % (t / 1e3),
          "GUI_ACTIVE/ns=%.3f" % (mean.get("GRBM_GUI_ACTIVE", 0) / max(t, 1)))
PY
