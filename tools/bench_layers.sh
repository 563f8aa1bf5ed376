#!/bin/bash
# Forward / data gradient / filter gradient of every convolution shape of the encoder at batch 128
# (tools/bench_conv.py per shape): the per-layer table behind the step's MFMA kernels.
for c in "128,224,224,64,64" "128,112,112,64,128" "128,112,112,128,128" "128,56,56,128,256" "128,56,56,256,256" "128,28,28,256,512" "128,28,28,512,512" "128,14,14,512,512"; do
  CONV=$c REPS=10 timeout -k 10 120 python tools/bench_conv.py 2>/dev/null
done
