#!/bin/bash
# Runs on the GPU box: kernel timeline of ONE mid-size csi_estimate_device call (the last traced one): every kernel's queue, start
# offset and duration.  usage: tools/call_timeline.sh <packets> ["opt=v,opt=v"]   -> stdout
cd "$(dirname "$0")/.."; R=$(pwd); cd /tmp && export TMPDIR=/tmp && cd $R
N=${1:-64}; OPTS=${2:-band_split=-1}
D=gpurun_out/timeline_$N; rm -rf $D; mkdir -p $D
SIZES=$N rocprofv3 --kernel-trace --output-format csv -d $D -o kt -- python tools/regime_probe.py "$OPTS" 2> $D/err.txt | tail -1
python - $D <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/kt_kernel_trace.csv', recursive=True)[0]
rows = sorted(({'k': r['Kernel_Name'].split('(')[0].replace('void csi::', '').replace('void ', '')[:44], 's': int(r['Start_Timestamp']), 'e': int(r['End_Timestamp']),
                'q': r.get('Queue_Id', '?'), 'g': r.get('Grid_Size', '?'), 'w': r.get('Workgroup_Size', '?')} for r in csv.DictReader(open(f))), key=lambda r: r['s'])
ls = [i for i, r in enumerate(rows) if r['k'].startswith('ls_') or r['k'].startswith('small_l0_ls')]      # (a one-packet call starts with the fused LS + layer-0 launch)
first = ls[-2] if len(ls) >= 2 else 0          # the call before the last one: complete
last = ls[-1]
t0 = rows[first]['s']
for r in rows[first:last]:
    print('   q%-3s %-46s grid %9s  start %8.1f us  dur %7.1f us  end %8.1f' % (r['q'], r['k'], r['g'], (r['s'] - t0) / 1e3, (r['e'] - r['s']) / 1e3, (r['e'] - t0) / 1e3))
PY
rm -rf $D/*/
