#!/bin/bash
# HBM-side traffic of the fused pair-loss kernel, calibrated on the kernel's own access shapes.
# FETCH_SIZE on gfx950 tallies every read request of the L2's memory side at 64 bytes whether it asks for 64
# or 128, so its factor depends on the mix of whole and half lines; the requests counted by size
# (TCC_EA0_RDREQ_32B/_64B/_128B) give the bytes directly.  Both are collected (a rocprofv3 run each) over
# build/fetch_calibration (tools/microbench/fetch_calibration.hip: kernels that read a known byte count with each
# load shape of the kernel and with its mix) and over the kernel itself (tools/tune_fused.py), plus WRITE_SIZE.
#   <out>_calibration.json   per shape: known bytes, FETCH_SIZE and its factor, bytes from the sized requests
#   <out>.json               bytes per launch of the kernel, and the sha256 of the kernel source
# usage (on the GPU box): bash tools/pmc_traffic.sh gpurun_out/traffic
set -u
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export REPS=20
SIZED="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
rm -rf "$out"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/CAL" -- build/fetch_calibration > "$out.CAL.log" 2>&1 || { echo "calibration pass failed"; tail -5 "$out.CAL.log"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc $SIZED --output-format csv -d "$out/CAL_SIZED" -- build/fetch_calibration > "$out.CAL_SIZED.log" 2>&1 || { echo "calibration pass (sized requests) failed"; tail -5 "$out.CAL_SIZED.log"; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d "$out/$c" -- python3 tools/tune_fused.py > "$out.$c.log" 2>&1 || { echo "pass $c failed"; tail -5 "$out.$c.log"; exit 1; }
done
timeout -s KILL 120 rocprofv3 --pmc $SIZED --output-format csv -d "$out/SIZED" -- python3 tools/tune_fused.py > "$out.SIZED.log" 2>&1 || { echo "pass sized requests failed"; tail -5 "$out.SIZED.log"; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, hashlib, json, re, sys, collections
out = sys.argv[1]

def counters(sub, match):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(match, row["Kernel_Name"])
            if m:
                acc[m.group(0)][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc

def mean(v, skip_first=False):
    v = v[1:] if skip_first and len(v) > 1 else v
    return sum(v) / len(v)

def sized_bytes(c, skip_first=False):
    n32, n64, n128 = (mean(c["TCC_EA0_RDREQ_%s_sum" % s], skip_first) for s in ("32B", "64B", "128B"))
    return {"requests": mean(c["TCC_EA0_RDREQ_sum"], skip_first), "requests_32B": n32, "requests_64B": n64,
            "requests_128B": n128, "bytes": 32.0 * n32 + 64.0 * n64 + 128.0 * n128}

known = json.loads([l for l in open(out + ".CAL.log") if l.startswith('{"planes"')][-1])["bytes"]
cal, cal_sized = counters("CAL", r"cal_[a-z0-9_]+"), counters("CAL_SIZED", r"cal_[a-z0-9_]+")
# bytes moved in whole 128-byte lines over bytes read, where the geometry fixes it: the 2-byte lanes read whole
# lines; a window row of 320 (depth) or 480 (colour) bytes starts at a multiple of its length and shares its
# first / last line with the neighbouring workgroup, which runs on another XCD with an L2 of its own
line_ratio = {"cal_u16_lane": 1.0, "cal_x16_chunk": 384.0 / 320.0, "cal_q8x3_chunk": 18.0 / 15.0}
shapes = {}
for name, nbytes in known.items():
    # the first repetition reads buffers that hipMemset has just written (partly cache resident): use the later ones
    kb = mean(cal[name]["FETCH_SIZE"], True)
    sz = sized_bytes(cal_sized[name], True)
    shapes[name] = {"known_bytes": nbytes, "FETCH_SIZE_kb": kb, "fetch_size_factor": nbytes / (kb * 1024.0),
                    "sized_requests": sz, "sized_bytes_over_known": sz["bytes"] / nbytes,
                    "whole_line_bytes_over_known": line_ratio.get(name)}
json.dump({"tool": "rocprofv3 --pmc FETCH_SIZE | TCC_EA0_RDREQ_{,32B_,64B_,128B_}sum -- build/fetch_calibration",
           "meaning": "per load shape of pair_loss_fused_kernel: fetch_size_factor = bytes read / (FETCH_SIZE * 1024); "
                      "sized_bytes_over_known = (32 n32 + 64 n64 + 128 n128) / bytes read; whole_line_bytes_over_known = the "
                      "same ratio predicted from the 128-byte lines each workgroup touches (null where waves of one "
                      "workgroup may or may not share a line in the L2)",
           "shapes": shapes}, open(out + "_calibration.json", "w"), indent=1)

k = counters("FETCH_SIZE", "pair_loss_fused")["pair_loss_fused"]["FETCH_SIZE"]
w = counters("WRITE_SIZE", "pair_loss_fused")["pair_loss_fused"]["WRITE_SIZE"]
sz = sized_bytes(counters("SIZED", "pair_loss_fused")["pair_loss_fused"])
fetch_kb, write_kb = mean(k), mean(w)
worst_cal = max(abs(v["sized_bytes_over_known"] / v["whole_line_bytes_over_known"] - 1.0)
                for v in shapes.values() if v["whole_line_bytes_over_known"])
src = "slam_amd/csrc/pair_loss.hip"
res = {
    "kernel": "pair_loss_fused_kernel", "workload": "64 pairs 640x480 per launch, 4 rotated HBM batches",
    "launches_averaged": len(k), "FETCH_SIZE_kb_raw": fetch_kb, "WRITE_SIZE_kb_raw": write_kb,
    "sized_requests": sz,
    "fetch_size_factor": sz["bytes"] / (fetch_kb * 1024.0),
    "correction": "reads = 32 n32 + 64 n64 + 128 n128 from TCC_EA0_RDREQ_{32B,64B,128B}_sum (within %.3f%% of the bytes "
                  "moved in whole 128-byte lines on the calibration shapes whose line count the geometry fixes); "
                  "FETCH_SIZE counts each of these requests as 64 bytes; WRITE_SIZE exact" % (100 * worst_cal),
    "calibration_worst_error": worst_cal,
    "hbm_read_bytes_per_launch": sz["bytes"], "hbm_write_bytes_per_launch": write_kb * 1024.0,
    "source": src, "source_sha256": hashlib.sha256(open(src, "rb").read()).hexdigest(),
}
res["hbm_bytes_per_launch"] = res["hbm_read_bytes_per_launch"] + res["hbm_write_bytes_per_launch"]
json.dump(res, open(out + ".json", "w"), indent=1)
print(json.dumps({n: [round(f["fetch_size_factor"], 4), round(f["sized_bytes_over_known"], 4)] for n, f in shapes.items()}))
print(json.dumps(res))
PY
