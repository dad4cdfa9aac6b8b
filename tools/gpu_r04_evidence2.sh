#!/bin/bash
# Round 4: the evidence once more on the FINAL source hash (a header comment moved it), then — profiles/pmc_summary.json
# rebuilt on the box from these very passes — the bench lines with traffic / bound_by quoted.
set -u
TAG=${1:-r04v}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
if [ -z "${KT_SKIP_SUBSET:-}" ]; then
  timeout 600 python -m pytest tests -m gpu -x -q -k "golden or lean_sweep or event_sized or config2_full or few_pod or abi_flat" > $OUT/${TAG}_pytest_subset.log 2>&1; echo "pytest subset: exit $?"; tail -2 $OUT/${TAG}_pytest_subset.log
fi
bash tools/round_evidence.sh $TAG > $OUT/${TAG}_evidence.log 2>&1; tail -40 $OUT/${TAG}_evidence.log | cut -c1-300
cp profiles/pmc_summary.json /tmp/pmc_before.json
bash tools/summarise_round.sh $TAG r04tmp > /dev/null 2>&1
for c in 2 3 4; do
  timeout 400 python bench.py --config $c --steps 500 --warmup 10 --no-cpu-baseline --no-latency > $OUT/${TAG}_quoted_bench_cfg$c.json 2> $OUT/${TAG}_quoted_bench_cfg$c.err; echo "quoted bench cfg$c: exit $?"
done
timeout 400 python bench.py --config 2 --pods-per-gpu 4000000 --steps 200 --warmup 10 --no-cpu-baseline --no-latency > $OUT/${TAG}_quoted_bench_cfg2_4M.json 2> $OUT/${TAG}_quoted_bench_cfg2_4M.err; echo "quoted bench cfg2 4M: exit $?"
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_quoted_bench_cfg2_driver.json 2> $OUT/${TAG}_quoted_bench_cfg2_driver.err; echo "quoted bench cfg2 (driver arguments): exit $?"
for f in $OUT/${TAG}_quoted_bench_cfg*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%s: %.4f ms/step frac %.3f (rocprof %s) traffic %s" % (sys.argv[1].split("/")[-1], d["ms_per_step"], r["frac"], r["check"].get("frac_rocprof"), r.get("traffic")))
except Exception as ex:
    print(sys.argv[1], "no bench line:", ex)
PY
done
