#!/bin/bash
# Runs on the GPU box: A/B of the LS kernel in front of the DNN kernels (one stream) against LS on a CU-masked side stream
# ("ls_overlap_cus" = n compute units), alternating on one box.  usage: tools/ls_overlap_ab.sh "<n list>" [rounds]  -> stdout
NS=${1:-"0 16 32 64"}
ROUNDS=${2:-2}
for r in $(seq 1 $ROUNDS); do
  for n in $NS; do
    python bench.py --steps 20 --warmup 5 --no-other-configs --host-path 0 --no-cpu-baseline --no-latency --option ls_overlap_cus=$n $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
k=d.get('kernels',{})
print('ls_overlap_cus=%-3s round $r: %.3f ms/step  %.2f M pairs/s   ls %.3f ms  layer0 %.3f  band %.3f  parity %s' % ('$n', d['ms_per_step'], d['value']/1e6, k.get('ls_estimate',{}).get('ms_avg',0), k.get('layer0_ltf_gemm',{}).get('ms_avg',0), k.get('pair_dense_gemm',{}).get('ms_avg',0), json.dumps(d.get('parity_check'))[:160]))
"
  done
done
