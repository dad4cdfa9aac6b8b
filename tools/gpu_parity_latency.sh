#!/bin/bash
# One GPU call while iterating on the host side: the quick GPU parity suite + the latency legs of configs 2 and 4 (one shard).   gpurun --timeout 1200 -- "bash tools/gpu_parity_latency.sh <tag>"
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
TAG=${1:-r03k}
( time timeout 600 python -m pytest tests -m gpu -x -q -k "not config4 and not rccl" --durations=3 ) > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest: exit $?"; tail -8 $OUT/${TAG}_pytest_gpu.log
for cfg in 2 4; do
  env ${KT_DEBUG_COMPILE_ON:+KT_DEBUG_COMPILE=1} timeout 300 python tools/latency_bench.py --config $cfg > $OUT/${TAG}_latency_cfg$cfg.json 2> $OUT/${TAG}_latency_cfg$cfg.err
  echo "latency cfg$cfg: exit $?"; python - $OUT/${TAG}_latency_cfg$cfg.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])["latency"]
    for k, v in d.items(): print("   %-16s %s" % (k, v))
except Exception as ex:
    print("no latency line:", ex); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
done
