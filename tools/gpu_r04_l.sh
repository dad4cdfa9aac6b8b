#!/bin/bash
# Round 4, call L: kt_aggregate_bitmap PK knows at compile time that it runs over the scan view (no record hangs off the
# row list), request words of lanes that are not counted no longer masked; more VALU rates — parity (no shards), A/B
# against tools/ab/libkt_engine_prev.so (= the previous commit).
set -u
TAG=${1:-r04l}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
tools/microbench/valu_rates | tee $OUT/${TAG}_valu_rates.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "not one_shard and not stress_fresh" > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest -m gpu (no shards, no stress): exit $?"; tail -4 $OUT/${TAG}_pytest_gpu.log
show() {
python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("  %.3e %s  %.4f ms/step | check %.3f aggregate %.3f reconcile %.3f step %.3f | %s" % (d["value"], d["unit"], d["ms_per_step"],
          r["check"]["frac"], r["aggregate"]["frac"], r["reconcile"]["frac"], r["step"]["frac"], r["per_kernel_ms"]))
except Exception as ex:
    print("  no bench line:", ex)
PY
}
run() {  # name, env, bench args
  local name=$1 envs=$2; shift 2
  env $envs timeout 400 python bench.py "$@" --no-cpu-baseline --no-latency > $OUT/${TAG}_$name.json 2> $OUT/${TAG}_$name.err
  echo "bench $name [$envs]: exit $?"; show $OUT/${TAG}_$name.json; grep -v amdgpu.ids $OUT/${TAG}_$name.err | tail -2
}
OLD=KT_ENGINE_LIB=$REPO/tools/ab/libkt_engine_prev.so
for rep in 1 2; do
run bench_cfg2_$rep "KT_X=0" --config 2 --steps 1000 --warmup 10 --verify
run bench_cfg2_old_$rep "$OLD" --config 2 --steps 1000 --warmup 10 --verify
done
run bench_cfg3 "KT_X=0" --config 3 --steps 500 --warmup 10 --verify
run bench_cfg1 "KT_X=0" --config 1 --steps 500 --warmup 10 --verify
run bench_cfg1_old "$OLD" --config 1 --steps 500 --warmup 10 --verify
run bench_cfg4 "KT_X=0" --config 4 --steps 100 --warmup 5 --verify
run bench_cfg4_old "$OLD" --config 4 --steps 100 --warmup 5
run bench_cfg2_4M "KT_X=0" --config 2 --pods-per-gpu 4000000 --steps 100 --warmup 5
run bench_cfg2_4M_old "$OLD" --config 2 --pods-per-gpu 4000000 --steps 100 --warmup 5
