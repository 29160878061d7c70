#!/bin/bash
# config-2 step with the two component models one after the other (small_call_overlap=1 beyond 98 304 pair rows) against side by side on
# two streams at any size (=2), alternating on one box -> stdout
for r in 1 2 3; do for ov in 1 2; do
python bench.py --steps 20 --warmup 5 --no-other-configs --no-next-rows --no-cpu-baseline --no-latency --no-regimes --host-path 0 --check 2 --option small_call_overlap=$ov 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('small_call_overlap=$ov round $r: %.3f ms/step  %.2f M pairs/s  no-events %.3f ms  parity %s' % (d['ms_per_step'], d['value'] / 1e6, d['timed_region_events']['ms_per_step_without'], d['parity_check'].get('dnn_rel_err')))"
done; done
