#!/bin/bash
# fp32 band kernel (csi_band8), launch tail: one launch of 8000 bands (both component models' rows in one grid) against two of 4000 - and the
# round structure around it (256 CUs, one band per CU at a time).  Needs tools/band_probe.bin and tools/band8.hsaco.  -> stdout
cd "$(dirname "$0")/.."
for bands in 3840 4000 4096 7680 8000 8192; do
  BAND_M=$((bands * 128)) BAND8_HSACO=tools/band8.hsaco tools/band_probe.bin loop 32 csi_band8 2 2>&1 | grep "^loop" | sed "s/^/bands $bands (rounds of 256: $(python3 -c "print(round($bands/256,3))")): /"
done
