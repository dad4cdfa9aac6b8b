#!/bin/bash
# round 3, GPU call 1: GPU test-suite with the total full-size parity, baseline bench lines, design probes
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
nproc; rocm-smi --showuse 2>/dev/null | head -8
( time timeout 1000 python -m pytest tests -m gpu -x -q --durations=12 ) > $OUT/r03a_pytest_gpu.log 2>&1; echo "pytest -m gpu: exit $?"; tail -22 $OUT/r03a_pytest_gpu.log
timeout 120 tools/microbench/design_probe > $OUT/r03a_design_probe.txt 2>&1; echo "design probe: exit $?"; cat $OUT/r03a_design_probe.txt
timeout 120 python tools/rccl_two_ranks_one_gpu.py > $OUT/r03a_rccl2.txt 2>&1; echo "rccl probe: exit $?"; tail -3 $OUT/r03a_rccl2.txt
timeout 300 python bench.py --config 2 --steps 20 --warmup 5 > $OUT/r03a_bench_cfg2_driver.json 2> $OUT/r03a_bench_cfg2_driver.err; echo "bench cfg2 (driver args): exit $?"
timeout 300 python bench.py --config 2 --steps 2000 --warmup 20 --no-cpu-baseline --no-latency > $OUT/r03a_bench_cfg2.json 2> $OUT/r03a_bench_cfg2.err; echo "bench cfg2: exit $?"
timeout 300 python bench.py --config 4 --steps 50 --warmup 5 --no-cpu-baseline --no-latency > $OUT/r03a_bench_cfg4.json 2> $OUT/r03a_bench_cfg4.err; echo "bench cfg4: exit $?"
for f in $OUT/r03a_bench_cfg*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%s: %.3e %s %.4f ms/step | check %.3f aggregate %.3f reconcile %.3f step %.3f | %s | after-event %s" % (sys.argv[1].split("/")[-1], d["value"], d["unit"], d["ms_per_step"],
          r["check"]["frac"], r["aggregate"]["frac"], r["reconcile"]["frac"], r["step"]["frac"], r["per_kernel_ms"], r.get("step_after_event")))
    if d.get("latency"): print("   latency:", {k: v for k, v in d["latency"].items()})
except Exception as ex:
    print(sys.argv[1], "no bench line:", ex); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
done
