#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "multi_chunk or random_small_rich or golden or lean_sweep or edge_shapes or namespace_order or few_pod or pod_events_between or incremental_event" > gpurun_out/r05k_pytest.log 2>&1; echo "pytest subset: exit $?"; tail -5 gpurun_out/r05k_pytest.log
bash tools/gpu_ab.sh r05k "dyn rng rng@KT_NO_WG_RANGES=1" "4"
bash tools/gpu_ab.sh r05k "stage rng" "2"
