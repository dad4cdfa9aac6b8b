#!/bin/bash
# Round 4, call J: VALU issue rates (tools/microbench/valu_rates), the check's chunk prologue with the term numbers and
# their flags requested before the image is staged (A/B: libkt_engine_prev.so = the previous commit, libkt_engine_fte.so =
# also the first tile's records requested there, -DKT_FIRST_TILE_EARLY) — parity subset + bench lines.
set -u
TAG=${1:-r04j}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
tools/microbench/valu_rates | tee $OUT/${TAG}_valu_rates.txt
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "lean_sweep or random_small or golden or multi_chunk or config2_full or config3 or few_pod or pod_events or edge_shapes or wide_label" > $OUT/${TAG}_pytest_subset.log 2>&1; echo "pytest subset: exit $?"; tail -3 $OUT/${TAG}_pytest_subset.log
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
FTE=KT_ENGINE_LIB=$REPO/tools/ab/libkt_engine_fte.so
for rep in 1 2; do
run bench_cfg2_$rep "KT_X=0" --config 2 --steps 1000 --warmup 10 --verify
run bench_cfg2_old_$rep "$OLD" --config 2 --steps 1000 --warmup 10 --verify
run bench_cfg2_fte_$rep "$FTE" --config 2 --steps 1000 --warmup 10 --verify
done
run bench_cfg1 "KT_X=0" --config 1 --steps 500 --warmup 10 --verify
run bench_cfg1_old "$OLD" --config 1 --steps 500 --warmup 10 --verify
run bench_cfg1_fte "$FTE" --config 1 --steps 500 --warmup 10 --verify
run bench_cfg4 "KT_X=0" --config 4 --steps 100 --warmup 5 --verify
run bench_cfg4_old "$OLD" --config 4 --steps 100 --warmup 5
