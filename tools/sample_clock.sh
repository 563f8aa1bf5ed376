#!/bin/bash
# Shader clock and socket power sampled with rocm-smi while a command runs (usage: sample_clock.sh <log> <cmd...>)
log=$1; shift
"$@" > "$log" 2>&1 &
pid=$!
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | tr '\n' ' ' | sed 's/GPU\[0\]\s*: //g'
  echo
  sleep 0.4
done
wait $pid
