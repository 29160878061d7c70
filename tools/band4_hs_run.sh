#!/bin/bash
# the register-blocked split-f16 band kernel at the headline workload: band4 = 1 / 0 alternating, then variants given as arguments through the hooks
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/band4_hs.txt
: > $OUT
B="python bench.py --steps 10 --warmup 3 --check 4 --no-cpu-baseline --no-latency --no-other-configs --no-regimes --no-next-rows --host-path 0 --full-line"
for o in 1 0 1 0; do
  timeout 300 $B --option band4=$o 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('band4=$o ms/step', round(d['ms_per_step'],3), 'M pairs/s', round(d['value']/1e6,2), 'band ms', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],4), d['parity_check'].get('dnn_rel_err'), d['split_engine_range_guard'])" >> $OUT 2>&1
done
export CSI_DEBUG_HOOKS=1 CSI_BAND8_HSACO=tools/band8.hsaco
for v in "$@"; do
  CSI_BAND8_NAME=$v timeout 300 $B --check 0 --option band4=0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v ms/step', round(d['ms_per_step'],3), 'band ms', round(d['roofline']['avg_launch_ms'],4))" >> $OUT 2>&1
done
cat $OUT
