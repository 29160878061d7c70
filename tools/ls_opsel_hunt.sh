#!/bin/bash
# ls_opsel_hunt.sh - builds the library once per form of the LS transform's packed operations (CSI_LS_VAR_DEFAULT, ls_estimate.hip.h) into
# build_variants/libcsi_v<N>.so (CPU side, hipcc cross-compiles); `tools/ls_opsel_hunt.sh run` on the GPU box counts the wrong calls of each
# with tools/ls_opsel_count.py on the reproducible case (bf16 context, second stream forked in front of the LS kernel).
# usage: tools/ls_opsel_hunt.sh build 0 2 16 32 64 | tools/ls_opsel_hunt.sh run 0 2 16 32 64
set -u
R=$(cd $(dirname $0)/.. && pwd)
MODE=$1; shift
if [ "$MODE" = build ]; then
  python -c "import sys; sys.path.insert(0,'$R'); import dl_channel_estimation_mamimo_amd as p; p._lib.build_band_kernel(False)" || exit 1
  for v in "$@"; do
    ( /opt/rocm/bin/hipcc -DCSI_LS_VAR_DEFAULT=$v --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value \
        $R/dl-channel-estimation-mamimo_amd/csrc/csi_mamimo.hip -o $R/build_variants/libcsi_v$v.so 2>&1 | grep -E "error" ; echo "built v$v" ) &
  done
  wait
elif [ "$MODE" = time ]; then
  for v in "$@"; do
    echo "== variant $v"
    CSI_DEBUG_HOOKS=1 CSI_LIBRARY_PATH=$R/build_variants/libcsi_v$v.so timeout 100 python $R/tools/ls_opsel_time.py
  done
else
  for v in "$@"; do
    echo "== variant $v"
    CSI_DEBUG_HOOKS=1 CSI_LIBRARY_PATH=$R/build_variants/libcsi_v$v.so timeout 300 python $R/tools/ls_opsel_count.py ${CALLS:-20}
  done
fi
