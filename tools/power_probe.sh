#!/bin/bash
# Runs on the GPU box: socket power and shader clock (rocm-smi) while one split-f16 kernel variant runs in a loop.
# usage: tools/power_probe.sh <seconds per variant>   -> stdout
SEC=${1:-4}
for mode in mfma hs2hs pair regressor; do
  tools/hs_probe.bin loop 1 relu $mode $SEC > /tmp/loop_$mode.txt 2>&1 &
  PID=$!
  sleep 1.5
  for i in 1 2 3 4; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk" | tr '\n' ' '
    echo " [$mode]"
    sleep 0.5
  done
  wait $PID
  cat /tmp/loop_$mode.txt | tail -1
done
