#!/bin/bash
# bf16 form of the band kernel: ablation variants of band_kernel_gen.py through the library (BASELINE configs[2] shape), timing only
# (results of the ablated variants are wrong by design).  Needs tools/band8.hsaco (tools/build_band8.sh).
cd "$(dirname "$0")/.."
export CSI_DEBUG_HOOKS=1 CSI_BAND8_HSACO=tools/band8.hsaco
for v in ${@:-csi_band8_bf16 csi_band8_bf16_noconv csi_band8_bf16_noaside csi_band8_bf16_skeleton csi_band8_bf16_nostagger}; do
  CSI_BAND8_BF16_NAME=$v python bench.py --dtype bf16 --nt 64 --nr 4 --packets 5000 --steps 3 --warmup 2 --check 0 --no-cpu-baseline --no-latency --no-other-configs --option hs_band=2 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'ms/step', round(d['ms_per_step'],3), 'pair_dense_gemm ms', d.get('kernels_ms_per_launch', d.get('kernels', {})).get('pair_dense_gemm') if isinstance(d.get('kernels_ms_per_launch', d.get('kernels', {})), dict) else None, d['roofline'].get('avg_launch_ms'))"
done
