#!/bin/bash
# Times tools/bench_conv.py on one box for a list of library builds: tools/ab_variants.sh "SHAPE" WHICH lib1 lib2 ...
# ("default" = the in-tree library).  Ablated builds give wrong results; timing only.
shape=$1; which=$2; shift 2
for lib in "$@"; do
  l=$lib; [ "$lib" = default ] && l=""
  echo -n "$lib: "
  SLAMHIP_LIB=$l WHICH=$which CONV=$shape REPS=20 timeout -k 10 120 python tools/bench_conv.py 2>/dev/null || exit 1
done
