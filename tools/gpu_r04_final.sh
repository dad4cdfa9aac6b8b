#!/bin/bash
# Round 4, final call: the whole -m gpu suite (three configs[4] shards, the 500-round fresh-engine stress test with
# KT_STRESS_ROUNDS=500) on the final sources, then the round's evidence (tools/round_evidence.sh).
set -u
TAG=${1:-r04z}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 1300 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest -m gpu: exit $?"; tail -5 $OUT/${TAG}_pytest_gpu.log
KT_STRESS_ROUNDS=500 timeout 600 python -m pytest tests/test_engine_gpu.py -k stress_fresh -x -q -s > $OUT/${TAG}_stress.log 2>&1; echo "stress x500: exit $?"; grep -E "fresh-engine|passed|failed|round " $OUT/${TAG}_stress.log | head -8
bash tools/round_evidence.sh $TAG
