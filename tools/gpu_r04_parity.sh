#!/bin/bash
# Round 4, parity first (VERDICT r3 "do this" #1 and #2) in ONE GPU call:
#   (1) the deterministic regression test on the ROUND-3 library (tools/ab/libkt_engine_r03.so, built from 40ca545): must FAIL
#   (2) the same test on this tree's library: must pass
#   (3) the litmus test of the multi-group meet -> gpurun_out/<tag>_meet_litmus.txt
#   (4) the whole -m gpu suite with KT_DEBUG_POISON=1 (every device allocation filled with 0xA5)
#   (5) the stress test with KT_STRESS_ROUNDS (default 500) fresh engines per config
#   gpurun --timeout 2400 -- 'bash tools/gpu_r04_parity.sh r04a'
set -u
TAG=${1:-r04}; ROUNDS=${2:-500}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
T=tests/test_engine_gpu.py::test_reconcile_after_a_larger_scan_of_the_same_engine
if [ -f tools/ab/libkt_engine_r03.so ]; then
  KT_ENGINE_LIB=tools/ab/libkt_engine_r03.so timeout 600 python -m pytest $T -x -q > $OUT/${TAG}_regression_r03lib.log 2>&1
  echo "(1) regression test, round-3 library: exit $? (want 1)"; grep -E "AssertionError|passed|failed" $OUT/${TAG}_regression_r03lib.log | head -4
fi
timeout 600 python -m pytest $T -x -q > $OUT/${TAG}_regression.log 2>&1; echo "(2) regression test, this tree: exit $? (want 0)"; tail -1 $OUT/${TAG}_regression.log
if [ -x tools/microbench/meet_litmus ]; then
  timeout 600 tools/microbench/meet_litmus 1000 1024 > $OUT/${TAG}_meet_litmus.txt 2>&1; echo "(3) meet litmus: exit $?"; cat $OUT/${TAG}_meet_litmus.txt
fi
KT_DEBUG_POISON=1 timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest_gpu_poison.log 2>&1; echo "(4) pytest -m gpu under KT_DEBUG_POISON=1: exit $?"; tail -4 $OUT/${TAG}_pytest_gpu_poison.log
KT_STRESS_ROUNDS=$ROUNDS timeout 1200 python -m pytest tests/test_engine_gpu.py -k stress_fresh -x -q -s > $OUT/${TAG}_stress.log 2>&1; echo "(5) stress, $ROUNDS fresh engines per config: exit $?"; grep -E "fresh-engine|passed|failed|round " $OUT/${TAG}_stress.log | head -12
