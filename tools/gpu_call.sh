#!/bin/bash
# scratch driver of one GPU call (edited per experiment)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; mkdir -p gpurun_out
bash tools/gpu_ab.sh r05d "tree tree@KT_NO_VERDICT_IMAGES=1" "4"
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "multi_chunk or random_small_rich or golden or lean_sweep or edge_shapes or namespace_order or pod_events_between or few_pod" > gpurun_out/r05d_pytest.log 2>&1; echo "pytest subset: exit $?"; tail -5 gpurun_out/r05d_pytest.log
