#!/bin/bash
# The hunt for the rare bad first launch of the bf16-split LS kernel's two-workgroups-per-CU form (that form exists only in
# the hunt build, CSI_BUILD_DEFINES=CSI_LS_RACE_VARIANTS - the product library cannot select it).  One script, run on a GPU box:
#     tools/ls_race_box.sh <mode> [out-dir] [phase-1 seconds] [phase-2 seconds]
# Every mode first asks "does THIS box show events at all?" (about one box in four did) and runs its discriminating part only then.
#   repro      fresh-context cycles of the stress sequence (tools/ls_race_repro.py); on events: LDS pre-filled with NaN, one workgroup
#              per CU, the drain
#   warm       the same at one shape with a warm-up launch in front
#   variants   fast cycle (tools/ls_race_fast.py) over the VAR instantiations 0x200 ... 0x1000, then an engine that is reused
#   perturbed  the schedule-perturbed forms 0x800 ... 0x8800
#   scalar     packed op_sel operations against the SAME code with scalar operations in their place (0x4000 / 0xc000 ...)
#   pkadd      the instruction alone: tools/pkadd_mfma_probe.hip beside another workgroup's MFMAs, then 30 s of the LS fast cycle
#   pkadd-cold the same from prebuilt binaries (tools/prebuild_probes.sh), the probe FIRST on the cold box
MODE=${1:?mode: repro | warm | variants | perturbed | scalar | pkadd | pkadd-cold}
OUT=${2:-gpurun_out/ls_race_$MODE}
P1=${3:-60}
P2=${4:-600}
mkdir -p $OUT
export TMPDIR=/tmp
rocm-smi --showuniqueid 2>/dev/null | grep "Unique ID" > $OUT/box.txt
cat $OUT/box.txt
hunt_build() { CSI_BUILD_DEFINES=CSI_LS_RACE_VARIANTS python -c "import sys; sys.path.insert(0, '.'); import dl_channel_estimation_mamimo_amd as p; p._lib.build_library(force=True)" || exit 1; }
fast() { timeout $(($1 + 120)) python tools/ls_race_fast.py --kinds pm1 --loops 100000 --seconds $1 --variants $2 ${4:-} > $OUT/$3.txt 2>&1; grep "^variant\|cycles/s" $OUT/$3.txt | cut -c1-160; }
case $MODE in
repro)
  hunt_build
  timeout 420 python tools/ls_race_repro.py --loops 55 --device > $OUT/phase1.txt 2>&1; tail -1 $OUT/phase1.txt
  if grep -q "!!" $OUT/phase1.txt; then
    echo "EVENTS on this box"
    for v in "256 nanfill" "128 onewg" "64 drain"; do set -- $v; timeout 600 python tools/ls_race_repro.py --loops 80 --device --dbg $1 > $OUT/$2.txt 2>&1; tail -1 $OUT/$2.txt; done
  fi ;;
warm)
  hunt_build
  A="--shapes 16x4x2000 --kinds pm1,q16 --device"
  timeout 400 python tools/ls_race_repro.py $A --loops 150 > $OUT/phase1.txt 2>&1; tail -1 $OUT/phase1.txt
  if grep -q "!!" $OUT/phase1.txt; then
    echo "EVENTS on this box"
    timeout 700 python tools/ls_race_repro.py $A --loops 300 --warm 1 > $OUT/warm.txt 2>&1; tail -1 $OUT/warm.txt
  fi ;;
variants)
  hunt_build
  fast $P1 0 phase1
  if grep -q "!!" $OUT/phase1.txt; then echo "EVENTS on this box"; fast $P2 0,0x200,0x400,0x800,0x1000,128 variants; fast 150 0 reuse --reuse; fi ;;
perturbed)
  hunt_build
  fast $P1 0x800 phase1
  if [ "$(grep -c '!!' $OUT/phase1.txt)" -ge 3 ]; then echo "EVENTS on this box"; fast $P2 0x800,0xa00,0xc00,0x1800,0x2800,0x4800,0x8800,0x880 variants; fast 60 0x800 reuse --reuse; fi ;;
scalar)
  hunt_build
  V="0x4000,0xc000,0x4800,0x8000,0x800,0"
  fast $P1 $V phase1
  if grep -q "!!" $OUT/phase1.txt; then echo "EVENTS on this box"; fast $P2 $V variants; fi ;;
pkadd)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pkadd_mfma_probe.hip -o /tmp/pkadd_probe || exit 1
  for args in "0 50" "0 0" "1 50" "4 50" "2 50"; do timeout 90 /tmp/pkadd_probe 30 $args 2>&1 | tee -a $OUT/probe.txt | tail -4; done
  hunt_build
  fast 30 0x800,0 ls_fast ;;
pkadd-cold)
  timeout 40 tools/pkadd_probe.bin 10 0 0 2>&1 | tee -a $OUT/probe.txt | tail -3
  cp tools/_variants/libcsi_mamimo.so dl-channel-estimation-mamimo_amd/libcsi_mamimo.so
  fast 20 0x800,0 ls_fast
  for args in "0 50" "8 50"; do timeout 40 tools/pkadd_probe.bin 12 $args 2>&1 | tee -a $OUT/probe.txt | tail -3; done ;;
*) echo "unknown mode $MODE"; exit 2 ;;
esac
