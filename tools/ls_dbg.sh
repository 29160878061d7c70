for d in 0 1 2 3 4 7 15; do CSI_LS_DEBUG=$d CSI_LS_KERNEL=2 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --check 0 "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('dbg', $d, d['kernels']['ls_estimate']['ms_avg'])"; done
