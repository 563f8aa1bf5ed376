#!/bin/bash
# A/B of two builds of the library on one box: tools/ab_layers.sh build/variants/OLD.so [kernels] -> alternating runs
# of tools/bench_conv.py per convolution shape (WHICH selects fwd,dgrad,wgrad).
old=$1
which=${2:-fwd,dgrad,wgrad}
for c in "128,224,224,64,64" "128,112,112,64,128" "128,112,112,128,128" "128,56,56,128,256" "128,56,56,256,256" "128,28,28,256,512" "128,28,28,512,512" "128,14,14,512,512"; do
  for lib in "$old" ""; do
    echo "lib=${lib:-default}"
    SLAMHIP_LIB=$lib WHICH=$which CONV=$c REPS=20 timeout -k 10 120 python tools/bench_conv.py 2>/dev/null || exit 1
  done
done
