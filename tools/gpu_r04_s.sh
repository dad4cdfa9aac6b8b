#!/bin/bash
# Round 4: the configs[4] shard with 1/8, 1/4 and 1/2 of its throttles — time against words per pod, i.e. the fixed cost of a
# sweep (tiles, prologues) apart from the cost per visited word and match: calibrates the cost model of the anchored scan.
set -u
TAG=${1:-r04s}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
for t in 1250 2500 5000; do
  timeout 300 python bench.py --config 4 --throttles $t --steps 100 --warmup 5 --no-cpu-baseline --no-latency > $OUT/${TAG}_cfg4_T$t.json 2> $OUT/${TAG}_cfg4_T$t.err; echo "T=$t: exit $?"
done
for f in $OUT/${TAG}_cfg4_T*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%s: T=%d %.4f ms/step | %s | %s" % (sys.argv[1].split("/")[-1], d["config"]["throttles"], d["ms_per_step"], r["check"]["kernel"], r["per_kernel_ms"]))
except Exception as ex:
    print(sys.argv[1], "no bench line:", ex); print(open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
done
