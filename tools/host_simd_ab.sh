#!/bin/bash
# GPU box, ~1 minute: the host staging loops' SIMD choice (CSI_HOST_SIMD = 2 AVX2 / 5 AVX-512, read once per process) and the pinned
# result arrays of csi_estimate_c128, alternating processes on one box.  Not yet run on the Zen 5 hosts of the pool (the AVX-512 forms
# and the 16-thread input staging of the pinned-result path went in after the round's GPU minutes were spent; bits are validated).
#   bash tools/host_simd_ab.sh [rounds] > gpurun_out/host_simd_ab.txt
R=${1:-3}
export TMPDIR=/tmp
rocm-smi --showuniqueid 2>/dev/null | grep "Unique ID"
grep -m1 "model name" /proc/cpuinfo; cat /sys/fs/cgroup/cpu.max 2>/dev/null
for r in $(seq 1 $R); do
  for simd in 2 5; do
    echo "## round $r CSI_HOST_SIMD=$simd  DNN only"
    CSI_HOST_SIMD=$simd timeout 60 python tools/pinned_result_probe.py --reps 4 | grep -v "^setup\|^link"
    echo "## round $r CSI_HOST_SIMD=$simd  DNN + LS"
    CSI_HOST_SIMD=$simd timeout 60 python tools/pinned_result_probe.py --reps 4 --ls | grep -v "^setup"
  done
done
