#!/bin/bash
# Runs on the GPU box: the schedule-perturbed form (ls_debug 0x800) of the two-workgroups-per-CU bf16-split LS kernel showed the rare bad
# first launch ~70 x more often than the plain form on one box (profiles/r04_ls_ringb_variants.txt).  Does it on this box?  Then the
# discriminating variants on top of it, interleaved cycle by cycle.
OUT=${1:-gpurun_out/ls_race_box4}
P1=${2:-60}
P2=${3:-600}
mkdir -p $OUT
# the variant instantiations are not part of the product build
CSI_BUILD_DEFINES=CSI_LS_RACE_VARIANTS python -c "import sys; sys.path.insert(0, '.'); import dl_channel_estimation_mamimo_amd as p; p._lib.build_library(force=True)" || exit 1
rocm-smi --showuniqueid 2>/dev/null | grep "Unique ID" > $OUT/box.txt
cat $OUT/box.txt
V="0x800,0xa00,0xc00,0x1800,0x2800,0x4800,0x8800,0x880"
timeout 120 python tools/ls_race_fast.py --kinds pm1 --loops 3 --variants 0,$V,0x2000,0x4000,0x8000 > $OUT/sanity.txt 2>&1
grep "^variant" $OUT/sanity.txt | cut -c1-60
timeout $((P1 + 120)) python tools/ls_race_fast.py --kinds pm1 --loops 100000 --seconds $P1 --variants 0x800 > $OUT/phase1.txt 2>&1
tail -2 $OUT/phase1.txt | cut -c1-200
N=$(grep -c "!!" $OUT/phase1.txt)
if [ "$N" -ge 3 ]; then
  echo "EVENTS on this box: $N"
  timeout $((P2 + 120)) python tools/ls_race_fast.py --kinds pm1 --loops 100000 --seconds $P2 --variants $V > $OUT/variants.txt 2>&1
  grep "^variant\|cycles/s" $OUT/variants.txt | cut -c1-120
  timeout 200 python tools/ls_race_fast.py --kinds pm1 --loops 100000 --seconds 60 --variants 0x800 --reuse > $OUT/reuse.txt 2>&1
  grep "^variant\|cycles/s" $OUT/reuse.txt | cut -c1-200
  timeout 200 python tools/ls_race_fast.py --kinds pm1 --loops 100000 --seconds 60 --variants 0x800 --reuse --nok6 > $OUT/reuse_nok6.txt 2>&1
  grep "^variant\|cycles/s" $OUT/reuse_nok6.txt | cut -c1-200
fi
