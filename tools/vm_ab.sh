# A/B of the vector-memory schedules of the split-f16 layer-0 / pair kernels (bench.py --option hs_vm_cast / hs_vm_pair),
# alternating on one box:  bash tools/vm_ab.sh
for opt in "hs_vm_cast=0 hs_vm_pair=0" "hs_vm_cast=1 hs_vm_pair=1" "hs_vm_cast=2 hs_vm_pair=2" "hs_vm_cast=3 hs_vm_pair=3" "hs_vm_cast=2 hs_vm_pair=3" "hs_vm_cast=0 hs_vm_pair=0" "hs_vm_cast=1 hs_vm_pair=1" "hs_vm_cast=2 hs_vm_pair=2" "hs_vm_cast=3 hs_vm_pair=3" "hs_vm_cast=2 hs_vm_pair=3"; do
  o=""; for kv in $opt; do o="$o --option $kv"; done
  timeout 200 python bench.py --no-cpu-baseline --host-path 0 --no-latency --steps 10 $o 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$opt', round(d['value']/1e6,2), 'M pairs/s', round(d['ms_per_step'],3), 'ms  layer0', k['layer0_ltf_gemm']['ms_avg'], 'pair', k['pair_dense_gemm']['ms_avg'], 'regressor', k['regressor_gemm']['ms_avg'], 'dnn err', d['parity_check']['dnn_rel_err'])"
done
