#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; mkdir -p gpurun_out
bash tools/gpu_ab.sh r05h "q12 q12@KT_CHUNK_BUDGET=110000 q12@KT_CHUNK_BUDGET=80000" "4"
for b in 163000 110000 80000; do KT_CHUNK_BUDGET=$b KT_DEBUG_LDS=1 KT_ENGINE_LIB=tools/ab/libkt_engine_q12.so python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-extra 2>&1 >/dev/null | grep -E "bitmap index" | head -1; done
