#!/bin/bash
# One GPU call while iterating on a kernel: the quick GPU parity suite + one configs[4] shard, bench lines of configs 2 / 4 / 3 / 1 and 2 with 4M pods, rocprofv3 kernel stats of configs 2 / 4.   gpurun --timeout 1500 -- "bash tools/gpu_parity_bench.sh <tag>"
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
TAG=${1:-r03f}
( time timeout 600 python -m pytest tests -m gpu -x -q -k "not config4 and not rccl" --durations=3 ) > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest: exit $?"; tail -8 $OUT/${TAG}_pytest_gpu.log
( time timeout 300 python -m pytest tests -m gpu -x -q -k "config4_one_shard and 0" ) > $OUT/${TAG}_pytest_cfg4.log 2>&1; echo "pytest cfg4 shard 0: exit $?"; tail -3 $OUT/${TAG}_pytest_cfg4.log
show() {
python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%-20s %.4f ms/step | agg %.4f ms (frac %.3f) reduce %.4f finalize %.4f check %.4f | reconcile frac %.3f step frac %.3f" % (sys.argv[2], d["ms_per_step"], r["per_kernel_ms"]["aggregate"], r["aggregate"]["frac"], r["per_kernel_ms"]["reduce"], r["per_kernel_ms"]["finalize"], r["per_kernel_ms"]["check"], r["reconcile"]["frac"], r["step"]["frac"]))
except Exception as ex:
    print(sys.argv[2], "no bench line:", ex); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
}
for cfg in 2 4 3 1; do
  name=${TAG}_bench_cfg${cfg}
  timeout 300 python bench.py --config $cfg --steps 200 --warmup 10 --no-cpu-baseline --no-latency --verify > $OUT/$name.json 2> $OUT/$name.err
  echo "bench cfg$cfg: exit $?"; show $OUT/$name.json "cfg$cfg"
done
timeout 300 python bench.py --config 2 --pods-per-gpu 4000000 --steps 50 --warmup 5 --no-cpu-baseline --no-latency --verify > $OUT/${TAG}_bench_cfg2_4M.json 2> $OUT/${TAG}_bench_cfg2_4M.err; show $OUT/${TAG}_bench_cfg2_4M.json "cfg2 4M pods"
cd /tmp && export TMPDIR=/tmp
for cfg in 2 4; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_cfg$cfg -- python $REPO/bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline --no-latency > /dev/null 2> $OUT/${TAG}_prof_cfg$cfg.err
  f=$(find $OUT/prof_${TAG}_cfg$cfg -name '*kernel_stats.csv' | head -1)
  echo "kernel stats cfg$cfg:"; python - "$f" <<'PY'
import csv, sys
for i, r in enumerate(csv.reader(open(sys.argv[1]))):
    if i and i < 8: print("   %-60s calls %4s avg %10.1f ns" % (r[0][:60], r[1], float(r[3])))
PY
done
