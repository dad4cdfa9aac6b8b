#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
( time timeout 600 python -m pytest tests -m gpu -x -q -k "not config4" ) > $OUT/r03b_pytest_gpu.log 2>&1; echo "pytest -m gpu (without config4): exit $?"; tail -12 $OUT/r03b_pytest_gpu.log
timeout 120 tools/microbench/xcd_atomics > $OUT/r03b_xcd_atomics.txt 2>&1; echo "xcd probe: exit $?"; cat $OUT/r03b_xcd_atomics.txt
timeout 300 python bench.py --config 2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r03b_bench_cfg2.json 2> $OUT/r03b_bench_cfg2.err; echo "bench cfg2: exit $?"
KT_NO_FEW=1 timeout 300 python tools/latency_bench.py --config 2 > $OUT/r03b_latency_cfg2_nofew.json 2>&1; echo "latency (staged path): exit $?"
timeout 300 python tools/latency_bench.py --config 4 > $OUT/r03b_latency_cfg4.json 2>&1; echo "latency cfg4: exit $?"
python - <<'PY'
import json
for f in ("r03b_bench_cfg2.json",):
    try:
        d = json.loads(open("gpurun_out/" + f).read().strip().splitlines()[-1])
        print(f, "%.4f ms/step" % d["ms_per_step"], d["roofline"]["per_kernel_ms"]); print("  latency:", d["latency"])
    except Exception as ex:
        print(f, "no line", ex); print(open("gpurun_out/" + f.replace(".json", ".err")).read()[-2000:])
for f in ("r03b_latency_cfg2_nofew.json", "r03b_latency_cfg4.json"):
    print(f, open("gpurun_out/" + f).read()[-1500:])
PY
