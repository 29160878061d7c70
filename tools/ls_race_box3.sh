#!/bin/bash
# Runs on the GPU box (round 4, late): the fast repro (tools/ls_race_fast.py) first finds out whether this box shows the rare bad first
# launch of the two-workgroups-per-CU bf16-split LS kernel at all; only then the discriminating variants, interleaved cycle by cycle.
OUT=${1:-gpurun_out/ls_race_box3}
P1=${2:-120}
P2=${3:-900}
mkdir -p $OUT
# the variant instantiations are not part of the product build
CSI_BUILD_DEFINES=CSI_LS_RACE_VARIANTS python -c "import sys; sys.path.insert(0, '.'); import dl_channel_estimation_mamimo_amd as p; p._lib.build_library(force=True)" || exit 1
rocm-smi --showuniqueid 2>/dev/null | grep "Unique ID" > $OUT/box.txt
cat $OUT/box.txt
# every variant is a correct kernel?  (a handful of cycles each; an "event" in all of them would be a bug of the variant, not the race)
timeout 120 python tools/ls_race_fast.py --loops 4 --variants 0,0x200,0x400,0x800,0x1000,128 > $OUT/sanity.txt 2>&1
tail -7 $OUT/sanity.txt | cut -c1-160
timeout $((P1 + 120)) python tools/ls_race_fast.py --loops 100000 --seconds $P1 --variants 0 > $OUT/phase1.txt 2>&1
tail -2 $OUT/phase1.txt
if grep -q "!!" $OUT/phase1.txt; then
  echo "EVENTS on this box"
  timeout $((P2 + 120)) python tools/ls_race_fast.py --loops 100000 --seconds $P2 --variants 0,0x200,0x400,0x800,0x1000,128 > $OUT/variants.txt 2>&1
  tail -8 $OUT/variants.txt
  timeout 300 python tools/ls_race_fast.py --loops 100000 --seconds 150 --variants 0 --reuse > $OUT/reuse.txt 2>&1
  tail -2 $OUT/reuse.txt
fi
