#!/bin/bash
# Same-box A/B of the frame-pair kernel over library builds, alternating: tools/ab_fused.sh lib1 lib2 ...
# ("default" = the in-tree library) on the default workload, metric depth on a plane, and metric noise depth.
libs=("$@")
for round in 1 2; do
  for wl in "plane 1" "plane 5000" "noise 5000"; do
    read -r depth factor <<< "$wl"
    for lib in "${libs[@]}"; do
      l=$lib; [ "$lib" = default ] && l=""
      echo -n "$depth factor $factor $lib: "
      SLAMHIP_LIB=$l DEPTH=$depth FACTOR=$factor REPS=1000 timeout -k 10 120 python tools/tune_fused.py 2>/dev/null | cut -c1-60 || exit 1
    done
  done
done
