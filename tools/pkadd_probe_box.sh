#!/bin/bash
# Runs on the GPU box: tools/pkadd_mfma_probe.hip (packed op_sel adds checked in the kernel beside another workgroup's MFMAs) in its three
# neighbour modes, then 30 s of the LS kernel's own fast repro (perturbed form) to tell whether this box shows the LS events at all.
OUT=${1:-gpurun_out/pkadd_probe}
T=${2:-30}
mkdir -p $OUT
rocm-smi --showuniqueid 2>/dev/null | grep "Unique ID" > $OUT/box.txt
cat $OUT/box.txt
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pkadd_mfma_probe.hip -o /tmp/pkadd_probe || exit 1
for args in "0 50" "0 0" "1 50" "4 50" "2 50"; do
  timeout $((T + 60)) /tmp/pkadd_probe $T $args 2>&1 | tee -a $OUT/probe.txt | tail -4
done
CSI_BUILD_DEFINES=CSI_LS_RACE_VARIANTS python -c "import sys; sys.path.insert(0, '.'); import dl_channel_estimation_mamimo_amd as p; p._lib.build_library(force=True)" || exit 1
timeout 150 python tools/ls_race_fast.py --kinds pm1 --loops 100000 --seconds 30 --variants 0x800,0 > $OUT/ls_fast.txt 2>&1
grep "^variant\|cycles/s" $OUT/ls_fast.txt | cut -c1-150
