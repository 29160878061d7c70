#!/bin/bash
# ablation / schedule variants of the register-blocked bf16 band kernel through the library hooks at configs[2] (timing only)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/band4_variants.txt
: > $OUT
B="python bench.py --dtype bf16 --nt 64 --nr 4 --packets 5000 --steps 5 --warmup 2 --check 0 --no-cpu-baseline --no-latency --no-other-configs --no-regimes --no-next-rows --host-path 0 --full-line"
export CSI_DEBUG_HOOKS=1 CSI_BAND8_HSACO=tools/band8.hsaco
for v in "$@"; do
  CSI_BAND8_BF16_NAME=$v timeout 300 $B --option band4=0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v ms/step', round(d['ms_per_step'],3), 'band ms', round(d['roofline']['avg_launch_ms'],4))" >> $OUT 2>&1
done
cat $OUT
