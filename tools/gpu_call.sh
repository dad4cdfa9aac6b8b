#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; mkdir -p gpurun_out
bash tools/gpu_ab.sh r05w "tree n3" "4"
