#!/bin/bash
# Round 4, call B: the whole -m gpu suite on the new feed path / wide agreement / CheckRecs drain, the stress test on a
# second box, the bench line of configs[2] with the latency leg (pod event numbers), and the namespace-ordered scan of a
# single-chunk program as an A/B (KT_FORCE_NS_ORDER=1).
set -u
TAG=${1:-r04b}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest -m gpu: exit $?"; tail -6 $OUT/${TAG}_pytest_gpu.log
KT_STRESS_ROUNDS=500 timeout 600 python -m pytest tests/test_engine_gpu.py -k stress_fresh -x -q -s > $OUT/${TAG}_stress.log 2>&1; echo "stress x500: exit $?"; grep -E "fresh-engine|passed|failed|round " $OUT/${TAG}_stress.log | head -8
show() {
python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("  %.3e %s  %.4f ms/step | check %.3f aggregate %.3f reconcile %.3f step %.3f | %s" % (d["value"], d["unit"], d["ms_per_step"],
          r["check"]["frac"], r["aggregate"]["frac"], r["reconcile"]["frac"], r["step"]["frac"], r["per_kernel_ms"]))
    if d.get("latency"):
        print("   latency:", {k: v for k, v in d["latency"].items() if k.startswith(("upsert", "delete", "check1", "sweep"))})
except Exception as ex:
    print("  no bench line:", ex)
PY
}
timeout 400 python bench.py --config 2 --steps 500 --warmup 10 --no-cpu-baseline --verify > $OUT/${TAG}_bench_cfg2.json 2> $OUT/${TAG}_bench_cfg2.err; echo "bench cfg2: exit $?"; show $OUT/${TAG}_bench_cfg2.json; tail -2 $OUT/${TAG}_bench_cfg2.err
KT_FORCE_NS_ORDER=1 timeout 400 python bench.py --config 2 --steps 500 --warmup 10 --no-cpu-baseline --no-latency --verify > $OUT/${TAG}_bench_cfg2_nsorder.json 2> $OUT/${TAG}_bench_cfg2_nsorder.err; echo "bench cfg2 KT_FORCE_NS_ORDER=1: exit $?"; show $OUT/${TAG}_bench_cfg2_nsorder.json; tail -2 $OUT/${TAG}_bench_cfg2_nsorder.err
KT_FORCE_NS_ORDER=1 timeout 400 python bench.py --config 2 --pods-per-gpu 4000000 --steps 100 --warmup 5 --no-cpu-baseline --no-latency > $OUT/${TAG}_bench_cfg2_4M_nsorder.json 2> $OUT/${TAG}_bench_cfg2_4M_nsorder.err; echo "bench cfg2 4M KT_FORCE_NS_ORDER=1: exit $?"; show $OUT/${TAG}_bench_cfg2_4M_nsorder.json
timeout 400 python bench.py --config 4 --steps 50 --warmup 5 --no-cpu-baseline --no-latency > $OUT/${TAG}_bench_cfg4.json 2> $OUT/${TAG}_bench_cfg4.err; echo "bench cfg4: exit $?"; show $OUT/${TAG}_bench_cfg4.json
