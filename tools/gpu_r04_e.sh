#!/bin/bash
# Round 4, call E: the check's wordwise form with per-nibble `active` tables in every lean instantiation (A/B against the
# peel: tools/ab/libkt_engine_peel.so = the same sources built with -DKT_CHECK_PEEL), classes numbered larger-first,
# kt_feed_small pulling the pinned slot into device scratch first (A/B: KT_FEED_NO_STAGE=1).
set -u
TAG=${1:-r04e}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q -k "not one_shard and not stress_fresh" > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest -m gpu (no shards, no stress): exit $?"; tail -4 $OUT/${TAG}_pytest_gpu.log
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "one_shard and 0" > $OUT/${TAG}_pytest_shard0.log 2>&1; echo "configs[4] shard 0: exit $?"; tail -3 $OUT/${TAG}_pytest_shard0.log
show() {
python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("  %.3e %s  %.4f ms/step | check %.3f aggregate %.3f reconcile %.3f step %.3f | %s" % (d["value"], d["unit"], d["ms_per_step"],
          r["check"]["frac"], r["aggregate"]["frac"], r["reconcile"]["frac"], r["step"]["frac"], r["per_kernel_ms"]))
    for k, v in (d.get("latency") or {}).items():
        if k.startswith(("upsert", "delete", "check1", "sweep")): print("     ", k, v)
except Exception as ex:
    print("  no bench line:", ex)
PY
}
run() {  # name, env, bench args
  local name=$1 envs=$2; shift 2
  env $envs timeout 400 python bench.py "$@" --no-cpu-baseline > $OUT/${TAG}_$name.json 2> $OUT/${TAG}_$name.err
  echo "bench $name [$envs]: exit $?"; show $OUT/${TAG}_$name.json; grep -v amdgpu.ids $OUT/${TAG}_$name.err | tail -2
}
PEEL=KT_ENGINE_LIB=$REPO/tools/ab/libkt_engine_peel.so
run bench_cfg2 "KT_X=0" --config 2 --steps 500 --warmup 10 --verify
run bench_cfg2_peel "$PEEL" --config 2 --steps 500 --warmup 10 --no-latency --verify
run bench_cfg2_nostage "KT_FEED_NO_STAGE=1" --config 2 --steps 100 --warmup 10
run bench_cfg3 "KT_X=0" --config 3 --steps 500 --warmup 10 --no-latency --verify
run bench_cfg4 "KT_X=0" --config 4 --steps 100 --warmup 5 --no-latency
run bench_cfg4_peel "$PEEL" --config 4 --steps 100 --warmup 5 --no-latency
run bench_cfg2_4M "KT_X=0" --config 2 --pods-per-gpu 4000000 --steps 100 --warmup 5 --no-latency
run bench_cfg2_4M_peel "$PEEL" --config 2 --pods-per-gpu 4000000 --steps 100 --warmup 5 --no-latency
run bench_cfg1 "KT_X=0" --config 1 --steps 500 --warmup 10 --no-latency --verify
