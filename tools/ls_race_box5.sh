#!/bin/bash
# Runs on the GPU box: packed op_sel operations with two idle cycles behind each (ls_debug 0x4000) against the SAME code with scalar
# operations in their place (0xc000: 0x8000 + 0x4000), the plain forms (0 / 0x8000) and the perturbed forms (0x800 / 0x4800), interleaved.
# A short first phase finds out whether this box shows events at all (about one box in three does).
OUT=${1:-gpurun_out/ls_race_box5}
P1=${2:-60}
P2=${3:-420}
V="0x4000,0xc000,0x4800,0x8000,0x800,0"
mkdir -p $OUT
# the variant instantiations are not part of the product build
CSI_BUILD_DEFINES=CSI_LS_RACE_VARIANTS python -c "import sys; sys.path.insert(0, '.'); import dl_channel_estimation_mamimo_amd as p; p._lib.build_library(force=True)" || exit 1
rocm-smi --showuniqueid 2>/dev/null | grep "Unique ID" > $OUT/box.txt
cat $OUT/box.txt
timeout $((P1 + 120)) python tools/ls_race_fast.py --kinds pm1 --loops 100000 --seconds $P1 --variants $V > $OUT/phase1.txt 2>&1
grep "^variant\|cycles/s" $OUT/phase1.txt | cut -c1-150
if grep -q "!!" $OUT/phase1.txt; then
  echo "EVENTS on this box"
  timeout $((P2 + 120)) python tools/ls_race_fast.py --kinds pm1 --loops 100000 --seconds $P2 --variants $V > $OUT/variants.txt 2>&1
  grep "^variant\|cycles/s" $OUT/variants.txt | cut -c1-150
fi
