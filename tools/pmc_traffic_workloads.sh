#!/bin/bash
# HBM bytes per launch of pair_loss_fused_kernel for workloads other than the default one (k > 1 neighbours, 1280x960,
# metric depth on a plane and as per-pixel noise, where taps are gathered from outside the LDS window):
# two rocprofv3 --pmc passes each (L2 read requests by size; WRITE_SIZE) over tools/tune_fused.py, as
# tools/pmc_traffic.sh does for the default workload (whose calibration of the request sizes applies here too).
#   usage (on the GPU box): bash tools/pmc_traffic_workloads.sh gpurun_out/traffic_wl   -> <out>.json
set -u
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export REPS=20
SIZED="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
rm -rf "$out"; mkdir -p "$out"
i=0
for wl in "B=64 KN=1 SIZE=480x640" "B=32 KN=2 SIZE=480x640" "B=16 KN=4 SIZE=480x640" "B=16 KN=4 SIZE=960x1280" "B=16 KN=1 SIZE=960x1280" \
          "B=64 KN=1 SIZE=480x640 DEPTH=noise FACTOR=5000" "B=64 KN=1 SIZE=480x640 DEPTH=plane FACTOR=5000"; do
  i=$((i + 1))
  echo "$wl" > "$out/wl$i.txt"
  for pass in SIZED WRITE; do
    c="WRITE_SIZE"; [ $pass = SIZED ] && c="$SIZED"
    ( export $wl; timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d "$out/wl$i.$pass" -- python3 tools/tune_fused.py > "$out/wl$i.$pass.log" 2>&1 ) || { echo "pass $pass of '$wl' failed"; tail -5 "$out/wl$i.$pass.log"; exit 1; }
  done
done
python3 - "$out" <<'PY'
import csv, glob, hashlib, json, re, sys, collections
out = sys.argv[1]
def counters(sub):
    acc = collections.defaultdict(list)
    for f in glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "pair_loss_fused" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc
mean = lambda v: sum(v) / len(v)
res = []
for f in sorted(glob.glob(out + "/wl*.txt")):
    tag = f[:-4]
    env = dict(kv.split("=") for kv in open(f).read().split())
    B, k = int(env["B"]), int(env["KN"])
    H, W = [int(v) for v in env["SIZE"].split("x")]
    c, w = counters(tag.split("/")[-1] + ".SIZED"), counters(tag.split("/")[-1] + ".WRITE")
    rd = 32.0 * mean(c["TCC_EA0_RDREQ_32B_sum"]) + 64.0 * mean(c["TCC_EA0_RDREQ_64B_sum"]) + 128.0 * mean(c["TCC_EA0_RDREQ_128B_sum"])
    wr = mean(w["WRITE_SIZE"]) * 1024.0
    alg = B * H * W * 5 * (1 + k)
    res.append({"batch": B, "neighbours": k, "height": H, "width": W, "factor": float(env.get("FACTOR", 1.0)), "depth": env.get("DEPTH", "plane"),
                "launches_averaged": len(c["TCC_EA0_RDREQ_sum"]), "hbm_read_bytes_per_launch": rd,
                "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                "algorithmic_bytes_per_launch": alg, "anchor_read_once_bytes": alg,
                "anchor_read_per_neighbour_bytes": B * H * W * 10 * k, "traffic_over_algorithmic": (rd + wr) / alg})
src = "slam_amd/csrc/pair_loss.hip"
json.dump({"kernel": "pair_loss_fused_kernel", "tool": "rocprofv3 --pmc TCC_EA0_RDREQ_{,32B_,64B_,128B_}sum | WRITE_SIZE -- python3 tools/tune_fused.py",
           "meaning": "HBM bytes per launch = 32 n32 + 64 n64 + 128 n128 read requests of the L2's memory side + WRITE_SIZE; algorithmic = "
                      "B H W 5 (1 + k): the anchor's 5 bytes per pixel once, every neighbour's once",
           "workloads": res, "source": src, "source_sha256": hashlib.sha256(open(src, "rb").read()).hexdigest()},
          open(out + ".json", "w"), indent=1)
for r in res:
    print(r["batch"], r["neighbours"], "%dx%d" % (r["width"], r["height"]), r["depth"], r["factor"], "read %.1f MB write %.1f MB algorithmic %.1f MB ratio %.3f"
          % (r["hbm_read_bytes_per_launch"] / 1e6, r["hbm_write_bytes_per_launch"] / 1e6, r["algorithmic_bytes_per_launch"] / 1e6, r["traffic_over_algorithmic"]))
PY
