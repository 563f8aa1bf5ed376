#!/bin/bash
# Every measurement the round's profiles are copied from, in one run on the GPU box (each step under its own time
# limit, stopping at the first failure): bash tools/round_profiles.sh gpurun_out/rN
#   <out>_traffic*.json        HBM bytes per launch of the frame-pair kernel (default workload, other workloads)
#   <out>_enc.*                per-kernel table of the encoder training step
#   <out>_bench_stats.csv      rocprofv3 --kernel-trace --stats of the default bench.py command; <out>_bench_*.json its lines
#   <out>_other.jsonl          bench.py on the other inputs
set -u
out=$1
bash tools/pmc_traffic.sh ${out}_traffic || exit 1
bash tools/pmc_traffic_workloads.sh ${out}_traffic_wl || exit 1
# bench.py reports the traffic only from files whose source hash matches: install this run's
cp ${out}_traffic.json profiles/r3_traffic_fused.json && cp ${out}_traffic_calibration.json profiles/r3_traffic_calibration.json && cp ${out}_traffic_wl.json profiles/r3_traffic_workloads.json || exit 1
bash tools/trace_encoder.sh ${out}_enc > /dev/null || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d ${out}_bench.d -- python3 bench.py > ${out}_bench_traced.json 2> ${out}_bench_traced.err || { echo "traced bench failed"; tail -5 ${out}_bench_traced.err; exit 1; }
cp "$(find ${out}_bench.d -name '*kernel_stats.csv' | head -1)" ${out}_bench_stats.csv || exit 1
rm -rf ${out}_bench.d ${out}_enc.d
timeout -k 10 600 python bench.py > ${out}_bench_default.json 2> ${out}_bench_default.err || { echo "default bench failed"; tail -5 ${out}_bench_default.err; exit 1; }
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > ${out}_bench_driver.json 2> ${out}_bench_driver.err || { echo "driver-args bench failed"; exit 1; }
bash tools/bench_other_workloads.sh ${out}_other.jsonl || exit 1
echo "round profiles done"
