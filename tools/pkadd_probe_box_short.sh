#!/bin/bash
# Short form of tools/pkadd_probe_box.sh for the last GPU minutes of a round: everything prebuilt in the build container
# by tools/prebuild_probes.sh (tools/pkadd_probe.bin = hipcc of tools/pkadd_mfma_probe.hip; tools/_variants/libcsi_mamimo.so = the library built with
# CSI_BUILD_DEFINES=CSI_LS_RACE_VARIANTS), nothing compiled on the box.
#   1. the one GPU test added after the last full suite run, and smoke()
#   2. the op_sel probe in four neighbour modes, T seconds each
#   3. T2 seconds of the LS kernel's own fast repro (perturbed form beside the plain one): does THIS box show the LS events at all?
OUT=${1:-gpurun_out/pkadd_probe}
T=${2:-15}
T2=${3:-25}
mkdir -p $OUT
export TMPDIR=/tmp
rocm-smi --showuniqueid 2>/dev/null | grep "Unique ID" > $OUT/box.txt
cat $OUT/box.txt
timeout 150 python -m pytest tests/test_gpu_round4.py -m gpu -x -q -k "engine_close or clone_weights" > $OUT/tests.txt 2>&1
tail -2 $OUT/tests.txt
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.txt
for args in "0 50" "4 50" "2 50" "1 50"; do
  timeout $((T + 30)) tools/pkadd_probe.bin $T $args 2>&1 | tee -a $OUT/probe.txt | tail -4
done
cp tools/_variants/libcsi_mamimo.so dl-channel-estimation-mamimo_amd/libcsi_mamimo.so
timeout $((T2 + 40)) python tools/ls_race_fast.py --kinds pm1 --loops 100000 --seconds $T2 --variants 0x800,0 > $OUT/ls_fast.txt 2>&1
grep "^variant\|cycles/s" $OUT/ls_fast.txt | cut -c1-150
