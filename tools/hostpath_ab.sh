#!/bin/bash
# host-buffer entry points with / without the two-stream arrangement inside the pipeline's chunks, alternating on one box -> stdout
for r in 1 2; do for ov in 0 1; do
python bench.py --steps 3 --warmup 2 --no-other-configs --no-next-rows --no-cpu-baseline --no-latency --no-regimes --check 0 --option small_call_overlap=$ov 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); h = d['host_path_pcie_inclusive']
print('small_call_overlap=$ov round $r: planes pageable %.2f ms  c128->c64 dnn_only %.2f  pinned planes %.2f  c64 pinned %.2f  (pcie bound %.2f)' % (h['ms'], h['python_c128_to_c64']['dnn_only']['ms'], h['pinned_planes_dnn']['ms'], h['c64_pinned_in_and_out']['dnn_only']['ms'], h['pinned_planes_dnn'].get('pcie_bound_ms', 0)))"
done; done
