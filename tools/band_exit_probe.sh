#!/bin/bash
# Runs on the GPU box: the fused band kernel cut off after its prologue (exit1), after prologue + epilogue without the main loop
# (exit2), the same without the output stores (exit2_nostore) - what the part outside the main loop costs at the headline size.
cd $GRAFT_REPO_ROOT
for k in csi_band8 csi_band8_exit0 csi_band8_exit1 csi_band8_exit2 csi_band8_exit2_nostore csi_band8_exit2_noguard; do
  BAND_M=$((4000 * 128)) BAND8_HSACO=tools/band8.hsaco tools/band_probe.bin loop 32 $k 1 2>&1 | grep "^loop"
done
