#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace/stats and the two HBM counter passes for
# bench.py's default workload; raw output under gpurun_out/, summaries are copied to profiles/ by hand.
# Usage: tools/collect_profiles.sh <tag> [bench args...]
set -e
TAG=${1:-r01}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
# the stats pass runs bench.py with its DEFAULT step counts, so its per-kernel average is directly comparable with bench's own
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --no-cpu-baseline "$@" > $OUT/stats.log 2>&1
# HBM traffic: FETCH_SIZE and WRITE_SIZE need separate passes (TCC slots), counters only (no trace domains)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/sq.log 2>&1
tail -1 $OUT/stats.log
find $OUT -name "*kernel_stats.csv" | head -1 | xargs head -8
for d in fetch write sq; do python3 tools/pmc_summary.py $OUT/$d rollout; done
