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
# the kernel trace runs WARM (round 6): 400 timed steps behind 40 of warm-up — the first launches of a process are cold (clocks,
# caches, code objects) and 17 calls averaged them in: the kernel averages of round 5 summed to more than the step they explain.
# The two counter passes serialise the kernels anyway and stay short.
TRACE="python $REPO/bench.py --steps 200 --warmup 40 --min-seconds 0 --no-extra --no-cpu-baseline --no-latency $*"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $TRACE > $OUT/trace.json 2> $OUT/trace.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH > $OUT/fetch.json 2> $OUT/fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH > $OUT/write.json 2> $OUT/write.err
find $OUT -type f | head -50
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; head -12 $f; done
for f in $(find $OUT/fetch -name "*counter_collection.csv" | head -1); do echo "== $f"; head -5 $f; done
