#!/bin/bash
# PMC passes over tools/gemm_probe (GPU box): usage tools/probe_pmc.sh <probe args...>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/probe_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o pmc -- tools/gemm_probe "$@" > $OUT/run_sq.txt 2> $OUT/sq.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- tools/gemm_probe "$@" > $OUT/run_kt.txt 2> $OUT/kt.err
python3 - <<PY
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/sq/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r['Kernel_Name'][:90]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in rows.items():
    print(k)
    print('   ' + '  '.join(f"{n}={sum(v)/len(v):.4g}" for n, v in sorted(c.items())))
for f in glob.glob('$OUT/kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print(r['Name'][:90], r['Calls'], r['AverageNs'])
PY
