cd /root/repo
B="python bench.py --steps 10 --warmup 3 --check 0 --no-cpu-baseline --no-latency --no-other-configs --no-regimes --no-next-rows --host-path 0 --full-line --option band4=0"
export CSI_DEBUG_HOOKS=1 CSI_BAND8_HSACO=tools/band8.hsaco
for i in 1 2 3; do for v in csi_band4 csi_band4_roleslabs; do CSI_BAND8_NAME=$v $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['roofline']['avg_launch_ms'],4))"; done; done
B2="$B --dtype bf16 --nt 64 --nr 4 --packets 5000 --steps 5 --warmup 2"
for i in 1 2 3; do for v in csi_band4_bf16 csi_band4_bf16_roleslabs; do CSI_BAND8_BF16_NAME=$v $B2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['roofline']['avg_launch_ms'],4))"; done; done
