#!/bin/bash
# kernel trace of one-packet calls: per-kernel device time and the gaps between the kernels of a call.  -> stdout
cd "$(dirname "$0")/.."; R=$(pwd); cd /tmp && export TMPDIR=/tmp && cd $R
for fused in 1 0; do
  D=gpurun_out/small_trace_$fused; rm -rf $D; mkdir -p $D
  rocprofv3 --kernel-trace --output-format csv -d $D -o kt -- python tools/small_call_probe.py 200 $fused ${1:-estimate} 2> $D/err.txt | tail -1
  python - $D <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/kt_kernel_trace.csv', recursive=True)[0]
rows = sorted(({'k': r['Kernel_Name'].split('(')[0].replace('void csi::', '')[:50], 's': int(r['Start_Timestamp']), 'e': int(r['End_Timestamp'])} for r in csv.DictReader(open(f))), key=lambda r: r['s'])
rows = rows[len(rows) // 2:]                       # the second half: the timed loops
dur = collections.defaultdict(list)
for r in rows: dur[r['k']].append(r['e'] - r['s'])
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])): print('   %-52s n=%5d  avg %7.2f us' % (k, len(v), sum(v) / len(v) / 1e3))
# a "call" = a run of kernels; span from first start to last end per group of kernels separated by > 20 us... report the busy fraction instead
tot = rows[-1]['e'] - rows[0]['s']; busy = 0; cur_e = rows[0]['s']
for r in rows:
    busy += max(0, r['e'] - max(r['s'], cur_e)); cur_e = max(cur_e, r['e'])
print('   device busy %.1f %% of the traced span (%.1f ms, %d kernels)' % (100.0 * busy / tot, tot / 1e6, len(rows)))
PY
  rm -rf $D/*/
done
