#!/bin/bash
# ls_overlap_trace.sh - kernel trace of the two-stream bf16 repro: which kernels run WHILE the LS kernel runs (start / end, LDS per workgroup)
# usage (GPU box): tools/ls_overlap_trace.sh <lds_pad_bytes> ; library from CSI_LIBRARY_PATH
R=${GRAFT_REPO_ROOT:-$(pwd)}
PAD=${1:-0}
OUT=$R/gpurun_out/ls_overlap_$PAD
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp; cd $R
CSI_DEBUG_HOOKS=1 CSI_LS_LDS_PAD=$PAD rocprofv3 --kernel-trace --output-format csv -d $OUT -o kt -- python tools/ls_opsel_log.py 3 > $OUT/log.txt 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print('columns:', list(rows[0].keys()))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ls = [r for r in rows if 'ls_estimate' in r['Kernel_Name']]
for l in ls[-3:]:
    s, e = int(l['Start_Timestamp']), int(l['End_Timestamp'])
    print('LS launch %.1f us, LDS %s, grid %s, queue %s' % ((e - s) / 1e3, l.get('LDS_Block_Size'), l.get('Grid_Size'), l.get('Queue_Id')))
    for r in rows:
        rs, re_ = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if r is not l and rs < e and re_ > s:
            print('    overlaps: %-60s start %+8.1f us end %+8.1f us  LDS %s  wg %s grid %s queue %s' % (r['Kernel_Name'][:60], (rs - s) / 1e3, (re_ - s) / 1e3, r.get('LDS_Block_Size'), r.get('Workgroup_Size'), r.get('Grid_Size'), r.get('Queue_Id')))
PY
