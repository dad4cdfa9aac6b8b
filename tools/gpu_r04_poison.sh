#!/bin/bash
# Round 4: the WHOLE -m gpu suite under KT_DEBUG_POISON=1 (every device allocation filled with 0xA5) on the final sources,
# then the bench lines of configs 2 (driver arguments) / 4 / 2@4M with profiles/pmc_summary.json of the same source hash in
# place (traffic and bound_by quoted).
set -u
TAG=${1:-r04p}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
KT_DEBUG_POISON=1 timeout 1300 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest_gpu_poison.log 2>&1; echo "pytest -m gpu under KT_DEBUG_POISON=1: exit $?"; tail -4 $OUT/${TAG}_pytest_gpu_poison.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_cfg2_driver.json 2> $OUT/${TAG}_bench_cfg2_driver.err; echo "bench cfg2 (driver arguments): exit $?"
timeout 300 python bench.py --config 2 --steps 2000 --warmup 20 --no-cpu-baseline --no-latency > $OUT/${TAG}_bench_cfg2.json 2> $OUT/${TAG}_bench_cfg2.err; echo "bench cfg2: exit $?"
timeout 400 python bench.py --config 4 --steps 200 --warmup 10 --no-cpu-baseline --no-latency > $OUT/${TAG}_bench_cfg4.json 2> $OUT/${TAG}_bench_cfg4.err; echo "bench cfg4: exit $?"
timeout 400 python bench.py --config 2 --pods-per-gpu 4000000 --steps 200 --warmup 10 --no-cpu-baseline --no-latency > $OUT/${TAG}_bench_cfg2_4M.json 2> $OUT/${TAG}_bench_cfg2_4M.err; echo "bench cfg2 4M: exit $?"
timeout 400 python bench.py --config 3 --steps 500 --warmup 10 --no-cpu-baseline --no-latency > $OUT/${TAG}_bench_cfg3.json 2> $OUT/${TAG}_bench_cfg3.err; echo "bench cfg3: exit $?"
for f in $OUT/${TAG}_bench_cfg*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%s: %.4f ms/step frac %.3f traffic %s | check %s" % (sys.argv[1].split("/")[-1], d["ms_per_step"], r["frac"], r.get("traffic"), r["check"].get("bound_by")))
except Exception as ex:
    print(sys.argv[1], "no bench line:", ex)
PY
done
