for r in 1 2; do
for opt in "" "--option small_call_overlap=2" "--graph"; do
python bench.py --steps 20 --warmup 5 --no-other-configs --no-next-rows --host-path 0 --no-cpu-baseline --no-latency $opt 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-34s round $r: %.3f ms/step  %.2f M pairs/s  dnn err %.2e' % ('$opt', d['ms_per_step'], d['value']/1e6, d['parity_check']['dnn_rel_err']))
"
done; done
