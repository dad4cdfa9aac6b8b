#!/bin/bash
# Round 4: the final sources once more with every device allocation poisoned (KT_DEBUG_POISON=1: 0xA5 fill — a kernel that
# reads what nobody wrote shows up as a parity failure; configs[4] shards left out: 6 min of oracle time), and 200 rounds of
# the fresh-engine reconcile stress test.
set -u
TAG=${1:-r04w}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
KT_DEBUG_POISON=1 timeout 700 python -m pytest tests -m gpu -q --deselect "tests/test_engine_gpu.py::test_config4_one_shard[0]" --deselect "tests/test_engine_gpu.py::test_config4_one_shard[3]" --deselect "tests/test_engine_gpu.py::test_config4_one_shard[7]" > $OUT/${TAG}_pytest_gpu_poison.log 2>&1; echo "poisoned pytest -m gpu: exit $?"; tail -4 $OUT/${TAG}_pytest_gpu_poison.log
KT_STRESS_ROUNDS=200 timeout 300 python -m pytest tests/test_engine_gpu.py -k stress_fresh -x -q -s > $OUT/${TAG}_stress.log 2>&1; echo "stress x200: exit $?"; grep -E "fresh-engine|passed|failed" $OUT/${TAG}_stress.log | head -6
