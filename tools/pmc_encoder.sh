#!/bin/bash
# Matrix-core occupancy of the encoder training step (batch 128, 224x224x4), per kernel and for the step:
#   run 1  rocprofv3 --kernel-trace --stats      free-running kernel durations (the per-kernel table)
#   run 2  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES
#                          SQ_INSTS_VALU GRBM_GUI_ACTIVE
# both over  python3 tools/bench_encoder.py  with eager launches and whole steps only.  Writes
#   <out>.txt    per kernel: launches, share of the step's kernel time, counters, MFMA-busy %
#   <out>.json   mfma_busy_frac_of_step + sha256 of the kernel source
# MFMA-busy % of a kernel = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES): the cycles the matrix pipes
# of the busy CUs were executing, over the cycles those CUs had work.  The step's figure weights the kernels by
# their free-running time (run 1) and counts kernels without matrix instructions as zero.
# usage (on the GPU box): bash tools/pmc_encoder.sh gpurun_out/enc_mfma
set -u
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export ENC_GRAPH=0 ENC_FWD=0 ENC_STEPS=3
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 tools/bench_encoder.py > "$out.trace.log" 2>&1 || { echo "trace run failed"; tail -5 "$out.trace.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc" -- python3 tools/bench_encoder.py > "$out.pmc.log" 2>&1 || { echo "counter run failed"; tail -5 "$out.pmc.log"; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, hashlib, json, re, sys, collections
out = sys.argv[1]

def short(name):
    name = re.sub(r"\(.*", "", name)
    return name if len(name) <= 70 else name[:67] + "..."

dur = collections.defaultdict(lambda: [0, 0.0])          # name -> [launches, total ns]
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        d = dur[short(row["Kernel_Name"])]
        d[0] += 1
        d[1] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
cnt = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        cnt[short(row["Kernel_Name"])][row["Counter_Name"]] += float(row["Counter_Value"])
total_ns = sum(d[1] for d in dur.values())
lines, weighted = [], 0.0
steps = 5     # ENC_STEPS=3 timed + 2 warm-up steps, all whole training steps
for name, (n, ns) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
    c = cnt.get(name, {})
    busy_cu = c.get("SQ_BUSY_CU_CYCLES", 0.0)
    mfma = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    frac = mfma / (4.0 * busy_cu) if busy_cu else 0.0
    weighted += frac * ns
    lines.append("%-70s n/step=%6.1f  us/step=%9.1f  time=%5.1f%%  MFMA_BUSY=%.4g  BUSY_CU=%.4g  MOPS_BF16=%.4g  GUI_ACTIVE=%.4g  mfma_busy=%5.1f%%"
                 % (name, n / steps, ns / steps / 1e3, 100 * ns / total_ns, mfma / steps, busy_cu / steps,
                    c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) / steps, c.get("GRBM_GUI_ACTIVE", 0.0) / steps, 100 * frac))
step_frac = weighted / total_ns
src = "slam_amd/csrc/encoder.hip"
head = ["encoder training step, batch 128, 224x224x4, eager launches; tools/pmc_encoder.sh",
        "kernel time per step (free running, rocprofv3 --kernel-trace): %.2f ms; counters per step from a separate --pmc run" % (total_ns / steps / 1e6),
        "MFMA-busy of the step (time-weighted over all kernels): %.1f%%" % (100 * step_frac), ""]
open(out + ".txt", "w").write("\n".join(head + lines) + "\n")
json.dump({"workload": "encoder training step, batch 128, 224x224x4", "kernel_ms_per_step": total_ns / steps / 1e6,
           "mfma_busy_frac_of_step": step_frac,
           "definition": "sum_k time_k * SQ_VALU_MFMA_BUSY_CYCLES_k / (4 * SQ_BUSY_CU_CYCLES_k) / sum_k time_k",
           "source": src, "source_sha256": hashlib.sha256(open(src, "rb").read()).hexdigest()},
          open(out + ".json", "w"), indent=1)
print("\n".join(head + lines[:25]))
PY
