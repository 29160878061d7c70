#!/bin/bash
# numerics of the register-blocked bf16 band kernel, then configs[2] with band4 = 1 / 0, then variants given as arguments through the hooks
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 300 python tools/band4_check.py > gpurun_out/band4_check.txt 2>&1; echo "check rc=$?" >> gpurun_out/band4_check.txt
B="python bench.py --dtype bf16 --nt 64 --nr 4 --packets 5000 --steps 5 --warmup 2 --check 4 --no-cpu-baseline --no-latency --no-other-configs --no-regimes --no-next-rows --host-path 0 --full-line"
for o in 1 0 1 0; do
  timeout 300 $B --option band4=$o 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('band4=$o ms/step', round(d['ms_per_step'],3), 'band ms', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],4), d['parity_check'].get('dnn_rel_err_vs_bf16_emulation'))" >> gpurun_out/band4_check.txt 2>&1
done
export CSI_DEBUG_HOOKS=1 CSI_BAND8_HSACO=tools/band8.hsaco
for v in "$@"; do
  CSI_BAND8_BF16_NAME=$v timeout 300 $B --check 0 --option band4=0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v ms/step', round(d['ms_per_step'],3), 'band ms', round(d['roofline']['avg_launch_ms'],4))" >> gpurun_out/band4_check.txt 2>&1
done
cat gpurun_out/band4_check.txt
