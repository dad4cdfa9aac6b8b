#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
for i in 1 2 3 4 5 6 7 8; do timeout 300 python -m pytest tests/test_engine_gpu.py -m gpu -q -k "pod_events or incremental_event" 2>&1 | tail -1; done
