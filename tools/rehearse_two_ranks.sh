#!/bin/bash
# two ranks of bench.py sharing one GPU over gloo, started by hand so that a stuck rank can be made to print its stack
export BENCH_REHEARSAL_BACKEND=gloo PYTHONFAULTHANDLER=1 WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541
# (frame-pair legs only: gloo stalls on the encoder's all-reduces of GPU tensor slices, see tools/gloo_cuda_allreduce_check.py)
args="--gpus 2 --steps 20 --warmup 5 --encoder-steps ${ENC_STEPS:-0} --train-steps 0 --cpu-seconds 2"
( while sleep 25; do echo hb; done ) & HB=$!
RANK=0 LOCAL_RANK=0 timeout -s ABRT -k 10 ${LIMIT:-150} python -X faulthandler bench.py $args > gpurun_out/reh_r0.out 2> gpurun_out/reh_r0.err & P0=$!
RANK=1 LOCAL_RANK=1 timeout -s ABRT -k 10 ${LIMIT:-150} python -X faulthandler bench.py $args > gpurun_out/reh_r1.out 2> gpurun_out/reh_r1.err & P1=$!
wait $P0; echo rank0 rc=$?
wait $P1; echo rank1 rc=$?
kill $HB
for r in 0 1; do echo "== rank $r"; grep -v "amdgpu.ids\|^hb" gpurun_out/reh_r$r.err | tail -45; done
