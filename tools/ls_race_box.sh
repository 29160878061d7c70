#!/bin/bash
# Runs on the GPU box: is this one of the boxes on which the two-workgroups-per-CU form of the bf16-split LS kernel fails now and then?
# Phase 1: 110 fresh-context cycles of the stress sequence (forced ls_kernel 7 at Nt = 16 / 24).  Only if that shows events: the same
# with the LDS pre-filled with NaN (a read of something the workgroup has not written turns into NaN) and with one workgroup per CU.
OUT=${1:-gpurun_out/ls_race_box}
mkdir -p $OUT
rocm-smi --showuniqueid 2>/dev/null | grep "Unique ID" > $OUT/box.txt
timeout 420 python tools/ls_race_repro.py --loops 55 --device > $OUT/phase1.txt 2>&1
tail -1 $OUT/phase1.txt
if grep -q "!!" $OUT/phase1.txt; then
  echo "EVENTS on this box: running the discriminating variants"
  timeout 600 python tools/ls_race_repro.py --loops 80 --device --dbg 256 > $OUT/nanfill.txt 2>&1; tail -1 $OUT/nanfill.txt; grep -A3 "!!" $OUT/nanfill.txt | head -30
  timeout 600 python tools/ls_race_repro.py --loops 80 --device --dbg 128 > $OUT/onewg.txt 2>&1; tail -1 $OUT/onewg.txt
  timeout 600 python tools/ls_race_repro.py --loops 80 --device --dbg 64 > $OUT/drain.txt 2>&1; tail -1 $OUT/drain.txt
fi
