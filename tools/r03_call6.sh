#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
( time timeout 600 python -m pytest tests -m gpu -x -q -k "not config4" ) > $OUT/r03f_pytest_gpu.log 2>&1; echo "pytest: exit $?"; tail -15 $OUT/r03f_pytest_gpu.log
timeout 300 python tools/latency_bench.py --config 2 > $OUT/r03f_latency_cfg2.json 2>&1; echo "latency cfg2: exit $?"; tail -c 900 $OUT/r03f_latency_cfg2.json; echo
timeout 300 python tools/latency_bench.py --config 2 --pods 4000000 > $OUT/r03f_latency_cfg2_4M.json 2>&1; echo "latency cfg2 4M: exit $?"; tail -c 900 $OUT/r03f_latency_cfg2_4M.json; echo
timeout 300 python tools/latency_bench.py --config 4 > $OUT/r03f_latency_cfg4.json 2>&1; echo "latency cfg4: exit $?"; tail -c 900 $OUT/r03f_latency_cfg4.json; echo
