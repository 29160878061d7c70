#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + separate PMC passes of ANY command (tools/profile_config.sh does the same for bench.py
# configurations).  usage: tools/profile_cmd.sh <tag> -- <command ...>
#   e.g. SIZES=24 tools/profile_cmd.sh r05_midsize_24pkt -- python tools/regime_probe.py band_split=-1
# -> gpurun_out/prof_<tag>/{kt,pmc_fetch,pmc_write,pmc_sq,pmc_l2}; summarise with tools/profile_summarize.py <dir> <tag> profiles "<command>".
# PMC passes never combine with trace domains other than --kernel-trace (node-safety rule).
set -u
TAG=$1; shift
[ "${1:-}" = "--" ] && shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
echo "$*" > $OUT/cmd.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- "$@" > $OUT/out_kt.txt 2> $OUT/kt.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- "$@" > $OUT/out_fetch.txt 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- "$@" > $OUT/out_write.txt 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -o pmc -- "$@" > $OUT/out_sq.txt 2> $OUT/pmc_sq.err
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -o pmc -- "$@" > $OUT/out_l2.txt 2> $OUT/pmc_l2.err
# the summariser wants <pass>/<name>.csv directly under the pass directory: rocprofv3 nests them under a host directory
for p in kt pmc_fetch pmc_write pmc_sq pmc_l2; do
  for f in $(find $OUT/$p -name "*.csv"); do mv -n $f $OUT/$p/ 2>/dev/null; done
done
find $OUT -name "*.csv" ! -name "*kernel_trace.csv" ! -name "*counter_collection.csv" ! -name "*kernel_stats.csv" -delete 2>/dev/null
find $OUT -name "*.csv" | head -30
