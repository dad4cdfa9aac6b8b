#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
( time timeout 600 python -m pytest tests -m gpu -x -q -k "not config4" ) > $OUT/r03c_pytest_gpu.log 2>&1; echo "pytest -m gpu (without config4): exit $?"; tail -6 $OUT/r03c_pytest_gpu.log
timeout 300 python tools/latency_bench.py --config 2 > $OUT/r03c_latency_cfg2.json 2>&1; echo "latency cfg2: exit $?"; tail -c 1200 $OUT/r03c_latency_cfg2.json
timeout 300 python tools/latency_bench.py --config 4 > $OUT/r03c_latency_cfg4.json 2>&1; echo "latency cfg4: exit $?"; tail -c 1200 $OUT/r03c_latency_cfg4.json
