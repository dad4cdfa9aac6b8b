#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; mkdir -p gpurun_out
timeout 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r05q_bench_cfg2_driver.json 2> gpurun_out/r05q_bench_cfg2_driver.err; echo "exit $?"
tail -c 600 gpurun_out/r05q_bench_cfg2_driver.json
