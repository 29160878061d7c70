#!/bin/bash
# Cold-box form of tools/pkadd_probe_box_short.sh: three of the four boxes that ever showed the LS events showed them in the first
# seconds of the first process, so the op_sel probe runs FIRST (10 s, matrix neighbours), then the LS kernel's fast repro tells
# whether this box shows events, then the probe again.  Everything prebuilt (tools/pkadd_probe.bin, tools/_variants/libcsi_mamimo.so).
OUT=${1:-gpurun_out/pkadd_probe_cold}
mkdir -p $OUT
export TMPDIR=/tmp
rocm-smi --showuniqueid 2>/dev/null | grep "Unique ID" > $OUT/box.txt
cat $OUT/box.txt
timeout 40 tools/pkadd_probe.bin 10 0 0 2>&1 | tee -a $OUT/probe.txt | tail -3
cp tools/_variants/libcsi_mamimo.so dl-channel-estimation-mamimo_amd/libcsi_mamimo.so
timeout 60 python tools/ls_race_fast.py --kinds pm1 --loops 100000 --seconds 20 --variants 0x800,0 > $OUT/ls_fast.txt 2>&1
grep "^variant\|cycles/s" $OUT/ls_fast.txt | cut -c1-150
for args in "0 50" "8 50"; do
  timeout 40 tools/pkadd_probe.bin 12 $args 2>&1 | tee -a $OUT/probe.txt | tail -3
done
