#!/bin/bash
# One GPU call that re-establishes the evidence of a round: the GPU test-suite, smoke(), the bench line of every
# BASELINE config that fits one GPU, and the rocprofv3 trace + PMC passes of the headline config.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/round_start.sh r02'
# Everything lands in gpurun_out/<tag>_*; copy what should be judged into profiles/.
set -u
TAG=${1:-rXX}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest -m gpu: exit $?"; tail -3 $OUT/${TAG}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.log 2>&1; echo "smoke: exit $?"; tail -1 $OUT/${TAG}_smoke.log
for cfg in 2 3 1 4; do
  timeout 400 python bench.py --config $cfg --steps 50 --warmup 5 > $OUT/${TAG}_bench_cfg${cfg}.json 2> $OUT/${TAG}_bench_cfg${cfg}.err
  echo "bench cfg$cfg: exit $?"; python - <<PY
import json
try:
    d = json.loads(open("$OUT/${TAG}_bench_cfg${cfg}.json").read().strip().splitlines()[-1])
    print("  %.3e %s  %.4f ms/step  roofline %.1f%% (%s)  per-kernel %s" % (d["value"], d["unit"], d["ms_per_step"],
          100 * d["roofline"]["frac"], d["roofline"]["kernel"], d["roofline"]["per_kernel_ms"]))
except Exception as ex:
    print("  no bench line:", ex)
PY
done
timeout 1000 bash tools/profile.sh ${TAG}_cfg2 --config 2 > $OUT/${TAG}_profile.log 2>&1; echo "profile: exit $?"; tail -15 $OUT/${TAG}_profile.log
