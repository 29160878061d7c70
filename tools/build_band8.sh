#!/bin/bash
# generate + assemble + link the band8 code object (csrc/band_kernel_gen.py) -> $1 (default tools/band8.hsaco)
set -e
HERE=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$HERE/tools/band8.hsaco}
TMP=$(mktemp -d)
python3 "$HERE/dl-channel-estimation-mamimo_amd/csrc/band4_kernel_gen.py" "$TMP/band8.s"      # every variant of band_kernel_gen.py and of band4_kernel_gen.py
/opt/rocm/lib/llvm/bin/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$TMP/band8.s" -o "$TMP/band8.o"
/opt/rocm/lib/llvm/bin/ld.lld -shared "$TMP/band8.o" -o "$OUT"
rm -rf "$TMP"
echo "$OUT"
