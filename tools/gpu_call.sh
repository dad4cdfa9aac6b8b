#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; mkdir -p gpurun_out
bash tools/gpu_ab.sh r05c "head head@KT_FORCE_NS_ORDER=1 tree tree@KT_FORCE_NS_ORDER=1" "2"
bash tools/gpu_ab.sh r05c "head tree" "4"
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "not config4 and not full_size and not sixteen and not 100k and not stress" > gpurun_out/r05c_pytest.log 2>&1; echo "pytest subset: exit $?"; tail -5 gpurun_out/r05c_pytest.log
