#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; mkdir -p gpurun_out
bash tools/gpu_ab.sh r05z "p_nomeet tree tree@KT_NO_AGG_ONE=1" "2" "--pods-per-gpu 4000000"
