#!/bin/bash
# One GPU call of the inner development loop: GPU test-suite (stops at the first failure), then bench lines of the
# configs given as arguments (default "2 4"), each optionally under environment variants "VAR=val,VAR2=val2".
#   gpurun --timeout 900 -- 'bash tools/gpu_iter.sh <tag> "2 4" "KT_CHECK_WGS_PER_CU=1"'
set -u
TAG=${1:-it}; CFGS=${2:-"2 4"}; VARIANTS=${3:-""}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest -m gpu: exit $?"; tail -25 $OUT/${TAG}_pytest_gpu.log
fi
show() {
python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("  %.3e %s  %.4f ms/step  roofline %.1f%% (%s)  per-kernel %s" % (d["value"], d["unit"], d["ms_per_step"],
          100 * d["roofline"]["frac"], d["roofline"]["kernel"], d["roofline"]["per_kernel_ms"]))
except Exception as ex:
    print("  no bench line:", ex)
PY
}
for cfg in $CFGS; do
  for var in "" $VARIANTS; do
    name=${TAG}_bench_cfg${cfg}$(echo "$var" | tr -c 'A-Za-z0-9\n' '_')
    env $(echo "$var" | tr ',' ' ') timeout 400 python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline --verify > $OUT/$name.json 2> $OUT/$name.err
    echo "bench cfg$cfg [$var]: exit $?"; show $OUT/$name.json; tail -2 $OUT/$name.err
  done
done
