#!/bin/bash
# Round 4, call M: the check's chunk prologue on the ONE instantiation (no scratch now) — term numbers + flags requested
# before the image is staged (-DKT_EARLY_TERMS), the first tile's records too (-DKT_FIRST_TILE_EARLY), both: A/B libraries
# in tools/ab/ against the default build.
set -u
TAG=${1:-r04m}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
show() {
python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("  %.4f ms/step | check %.3f | %s" % (d["ms_per_step"], r["check"]["frac"], r["per_kernel_ms"]))
except Exception as ex:
    print("  no bench line:", ex)
PY
}
run() {  # name, env, bench args
  local name=$1 envs=$2; shift 2
  env $envs timeout 400 python bench.py "$@" --no-cpu-baseline --no-latency > $OUT/${TAG}_$name.json 2> $OUT/${TAG}_$name.err
  echo "bench $name [$envs]: exit $?"; show $OUT/${TAG}_$name.json; grep -v amdgpu.ids $OUT/${TAG}_$name.err | tail -2
}
for rep in 1 2; do
  run bench_cfg2_$rep "KT_X=0" --config 2 --steps 1000 --warmup 10 --verify
  for v in ET FTE BOTH; do
    run bench_cfg2_${v}_$rep "KT_ENGINE_LIB=$REPO/tools/ab/libkt_engine_$v.so" --config 2 --steps 1000 --warmup 10 --verify
  done
done
run bench_cfg1 "KT_X=0" --config 1 --steps 500 --warmup 10 --verify
for v in ET FTE BOTH; do run bench_cfg1_$v "KT_ENGINE_LIB=$REPO/tools/ab/libkt_engine_$v.so" --config 1 --steps 500 --warmup 10 --verify; done
run bench_cfg2_4M "KT_X=0" --config 2 --pods-per-gpu 4000000 --steps 100 --warmup 5
run bench_cfg2_4M_BOTH "KT_ENGINE_LIB=$REPO/tools/ab/libkt_engine_BOTH.so" --config 2 --pods-per-gpu 4000000 --steps 100 --warmup 5
