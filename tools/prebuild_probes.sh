#!/bin/bash
# Build container (no GPU): what tools/ls_race_box.sh pkadd-cold expects to find prebuilt, so that the GPU box compiles nothing.
#   tools/pkadd_probe.bin               hipcc of tools/pkadd_mfma_probe.hip
#   tools/_variants/libcsi_mamimo.so    the library with the race-hunt instantiations of the LS kernel (CSI_BUILD_DEFINES=CSI_LS_RACE_VARIANTS);
#                                       the in-tree product library is put back afterwards, untouched
# Both are git-ignored (*.bin, *.so) and travel with the gpurun snapshot.
set -e
cd "$(dirname "$0")/.."
SO=dl-channel-estimation-mamimo_amd/libcsi_mamimo.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pkadd_mfma_probe.hip -o tools/pkadd_probe.bin
mkdir -p tools/_variants
[ -f $SO ] && cp -p $SO /tmp/libcsi_product.so
CSI_BUILD_DEFINES=CSI_LS_RACE_VARIANTS python -c "import sys; sys.path.insert(0, '.'); import dl_channel_estimation_mamimo_amd as p; p._lib.build_library(force=True)"
mv $SO tools/_variants/libcsi_mamimo.so
if [ -f /tmp/libcsi_product.so ]; then cp -p /tmp/libcsi_product.so $SO; touch $SO; else python -c "import __graft_entry__ as g; g.build()"; fi
ls -la tools/pkadd_probe.bin tools/_variants/libcsi_mamimo.so $SO
