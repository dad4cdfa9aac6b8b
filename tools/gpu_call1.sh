#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; mkdir -p gpurun_out
bash tools/gpu_ab.sh r05a "base q8 q12" "4 2"
KT_ENGINE_LIB=$REPO/tools/ab/libkt_engine_q8.so timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "multi_chunk or random_small_rich or golden or config2_full or lean_sweep or edge_shapes or namespace_order" > gpurun_out/r05a_pytest_q8.log 2>&1; echo "pytest q8 subset: exit $?"; tail -5 gpurun_out/r05a_pytest_q8.log
