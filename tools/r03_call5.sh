#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
( time timeout 600 python -m pytest tests -m gpu -x -q -k "not config4 and not rccl" --durations=3 ) > $OUT/r03e_pytest_gpu.log 2>&1; echo "pytest: exit $?"; tail -14 $OUT/r03e_pytest_gpu.log
show() {
python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%-34s %.4f ms/step | agg %s %.4f ms (frac %.3f) reduce %.4f finalize %.4f check %.4f | reconcile frac %.3f step frac %.3f" % (sys.argv[2], d["ms_per_step"], r["aggregate"]["kernel"], r["per_kernel_ms"]["aggregate"], r["aggregate"]["frac"], r["per_kernel_ms"]["reduce"], r["per_kernel_ms"]["finalize"], r["per_kernel_ms"]["check"], r["reconcile"]["frac"], r["step"]["frac"]))
except Exception as ex:
    print(sys.argv[2], "no bench line:", ex); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
}
for cfg in 2 3 4 1; do
  for var in "" ; do
    name=r03e_bench_cfg${cfg}_$(echo "$var" | tr -c 'A-Za-z0-9\n' '_')
    env $var timeout 300 python bench.py --config $cfg --steps 200 --warmup 10 --no-cpu-baseline --no-latency --verify > $OUT/$name.json 2> $OUT/$name.err
    echo "bench cfg$cfg [$var]: exit $?"; show $OUT/$name.json "cfg$cfg $var"
  done
done
timeout 300 python bench.py --config 2 --pods-per-gpu 4000000 --steps 50 --warmup 5 --no-cpu-baseline --no-latency --verify > $OUT/r03e_bench_cfg2_4M.json 2> $OUT/r03e_bench_cfg2_4M.err; show $OUT/r03e_bench_cfg2_4M.json "cfg2 4M pods"
