#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + separate PMC passes of bench.py on ANY configuration.
# usage: tools/profile_config.sh <tag> [kt-only] -- <bench.py flags of the configuration>
#   e.g. tools/profile_config.sh r05_bf16_config3 -- --dtype bf16 --nt 64 --nr 4 --packets 5000
# -> gpurun_out/prof_<tag>/{kt,pmc_fetch,pmc_write,pmc_sq,pmc_l2}; summarise with tools/profile_summarize.py.
# PMC passes never combine with trace domains other than --kernel-trace (node-safety rule).
set -u
TAG=$1; shift
KT_ONLY=0
if [ "${1:-}" = "kt-only" ]; then KT_ONLY=1; shift; fi
[ "${1:-}" = "--" ] && shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
# NEXT_ROWS=1 keeps the "next_rows" legs (LMMSE, training step, LS on the 802.11 pilot) in the profiled command
NR_FLAG="--no-next-rows"; [ "${NEXT_ROWS:-0}" = 1 ] && NR_FLAG=""
COMMON="--no-cpu-baseline --check 0 --no-latency --host-path 0 --no-other-configs $NR_FLAG --no-regimes"
BENCH="python bench.py --steps 3 --warmup 1 $COMMON $*"
BENCH_KT="python bench.py --steps ${KT_STEPS:-20} --warmup ${KT_WARMUP:-5} $COMMON $*"
echo "$BENCH_KT" > $OUT/cmd.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $BENCH_KT > $OUT/bench_kt.json 2> $OUT/kt.err
[ $KT_ONLY = 1 ] && exit 0
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/bench_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/bench_write.json 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -o pmc -- $BENCH > $OUT/bench_sq.json 2> $OUT/pmc_sq.err
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -o pmc -- $BENCH > $OUT/bench_l2.json 2> $OUT/pmc_l2.err
# keep only what the summariser reads (the merge back is capped at 64 MiB)
find $OUT -name "*.csv" ! -name "*kernel_trace.csv" ! -name "*counter_collection.csv" ! -name "*kernel_stats.csv" -delete 2>/dev/null
find $OUT -name "*.csv" | head -30
