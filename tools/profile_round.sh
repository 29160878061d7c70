#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes of the default bench.
# usage: tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/
# PMC passes never combine with trace domains other than --kernel-trace (node-safety rule).
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
BENCH="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --check 0 --no-latency --host-path 0 --no-other-configs --no-next-rows"
# the kernel trace runs the bench's default step counts (20 + 5 warmup), so that its per-kernel averages are the timed region's
BENCH_KT="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --check 0 --no-latency --host-path 0 --no-other-configs --no-next-rows"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $BENCH_KT > $OUT/bench_kt.json 2> $OUT/kt.err
[ "${2:-}" = "kt-only" ] && exit 0
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/bench_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/bench_write.json 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -o pmc -- $BENCH > $OUT/bench_sq.json 2> $OUT/pmc_sq.err
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -o pmc -- $BENCH > $OUT/bench_l2.json 2> $OUT/pmc_l2.err
find $OUT -name "*.csv" | head -30
