#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes for bench.py.
# Every profiler run is under its own `timeout`: a hung rocprofv3 must not eat the GPU budget.
# usage: tools/profile.sh <tag> [bench args...]     -> gpurun_out/prof_<tag>/{trace,fetch,write}
set -u
TAG=${1:-run}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 5 --warmup 2 --min-seconds 0 --no-extra --no-cpu-baseline --no-latency $*"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.json 2> $OUT/trace.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH > $OUT/fetch.json 2> $OUT/fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH > $OUT/write.json 2> $OUT/write.err
find $OUT -type f | head -50
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; head -12 $f; done
for f in $(find $OUT/fetch -name "*counter_collection.csv" | head -1); do echo "== $f"; head -5 $f; done
