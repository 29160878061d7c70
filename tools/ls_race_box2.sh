#!/bin/bash
# Runs on the GPU box: does a SMALL first launch of the two-workgroups-per-CU bf16-split LS kernel (ls_kernel 7 forced at Nt = 16) take
# the rare bad first full launch away?  Phase 1 finds out whether this box shows the events at all (tools/ls_race_box.sh); only then
# phase 2 (one-packet launch of the same kernel in front) and phase 3 (the baseline once more, for the rate).
OUT=${1:-gpurun_out/ls_race_box2}
mkdir -p $OUT
rocm-smi --showuniqueid 2>/dev/null | grep "Unique ID" > $OUT/box.txt
A="--shapes 16x4x2000 --kinds pm1,q16 --device"
timeout 400 python tools/ls_race_repro.py $A --loops 150 > $OUT/phase1.txt 2>&1
tail -1 $OUT/phase1.txt
if grep -q "!!" $OUT/phase1.txt; then
  echo "EVENTS on this box"
  timeout 700 python tools/ls_race_repro.py $A --loops 300 --warm 1 > $OUT/warm.txt 2>&1; tail -1 $OUT/warm.txt
  timeout 400 python tools/ls_race_repro.py $A --loops 150 > $OUT/phase3.txt 2>&1; tail -1 $OUT/phase3.txt
fi
