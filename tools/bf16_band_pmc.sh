#!/bin/bash
# bf16 band kernel (BASELINE configs[2] shape): every ablation variant of band_kernel_gen.py through the library under ONE rocprofv3
# counter pass each - time per launch, shader clock, matrix-pipe utilisation, LDS-array activity.  Results of ablated variants are
# wrong by design.  Needs tools/band8.hsaco (tools/build_band8.sh).   usage: tools/bf16_band_pmc.sh [variant ...]  -> stdout
cd "$(dirname "$0")/.."
R=$(pwd)
export CSI_DEBUG_HOOKS=1 CSI_BAND8_HSACO=tools/band8.hsaco
cd /tmp && export TMPDIR=/tmp && cd $R
CTR="${CTR:-SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT}"
for v in ${@:-csi_band8_bf16 csi_band8_bf16_noconv csi_band8_bf16_noaside csi_band8_bf16_skeleton}; do
  D=gpurun_out/bf16pmc/$v
  rm -rf $D; mkdir -p $D
  CSI_BAND8_BF16_NAME=$v rocprofv3 --pmc $CTR --output-format csv -d $D -o pmc -- python bench.py --dtype bf16 --nt 64 --nr 4 --packets 5000 --steps 3 --warmup 2 --check 0 --no-cpu-baseline --no-latency --no-other-configs --no-regimes --host-path 0 --option hs_band=2 --option band4=0 --no-next-rows --full-line > $D/bench.json 2> $D/err.txt
  python - $v $D <<'PY'
import csv, sys, collections, glob, json
v, d = sys.argv[1], sys.argv[2]
f = glob.glob(d + '/**/pmc_counter_collection.csv', recursive=True)
if not f:
    print(v, 'no counter file;', open(d + '/err.txt').read()[-300:]); sys.exit(0)
agg, dur, seen = collections.defaultdict(list), [], set()
for r in csv.DictReader(open(f[0])):
    if 'csi_band' not in r['Kernel_Name']:
        continue
    agg[r['Counter_Name']].append(float(r['Counter_Value']))
    if r['Dispatch_Id'] not in seen:
        seen.add(r['Dispatch_Id']); dur.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
m = lambda k: sum(agg[k]) / max(len(agg[k]), 1)
us = sum(dur) / max(len(dur), 1) / 1e3
bz = m('SQ_BUSY_CYCLES') / 32.0
try:
    b = json.loads(open(d + '/bench.json').read().strip().splitlines()[-1]); ev = b['roofline']['avg_launch_ms']
except Exception:
    ev = float('nan')
print('%-36s %8.1f us (events %.3f ms) clock %.2f GHz  mfma_util %.3f  wait_any %.2f  wait_inst %.2f (lds %.2f)  lds_active/cu-cycle %.3f  conflicts %.0f' % (
    v, us, ev, bz / max(us, 1e-9) / 1e3, (m('SQ_VALU_MFMA_BUSY_CYCLES') / 1024.0) / max(bz, 1), m('SQ_WAIT_ANY') / max(m('SQ_WAVE_CYCLES'), 1),
    m('SQ_WAIT_INST_ANY') / max(m('SQ_WAVE_CYCLES'), 1), m('SQ_WAIT_INST_LDS') / max(m('SQ_WAVE_CYCLES'), 1), (m('SQ_LDS_IDX_ACTIVE') / 256.0) / max(bz, 1), m('SQ_LDS_BANK_CONFLICT')))
PY
  rm -rf $D/*/ 2>/dev/null
done
