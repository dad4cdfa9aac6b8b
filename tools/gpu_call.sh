#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "multi_chunk or namespace_order or golden or random_small_rich or pod_events_between or event_bursts or config4_one_shard and 1]" > gpurun_out/r05v_pytest.log 2>&1; echo "pytest: exit $?"; tail -4 gpurun_out/r05v_pytest.log
bash tools/gpu_ab.sh r05v "tree" "2 4"
