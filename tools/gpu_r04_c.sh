#!/bin/bash
# Round 4, call C: the queued packed fold (kt_aggregate_bitmap PK) — parity subset, then A/B bench lines
# (KT_NO_FOLD_QUEUE=1 = the adds issued per peel step, as in round 3) on configs 2, 4 and 2 with 4M pods.
set -u
TAG=${1:-r04c}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_sharded_gpu.py tests/test_parity_extended_gpu.py -m gpu -x -q \
  -k "golden or config2_full or config3 or (one_shard and 0) or pod_events or event_bursts or larger_scan or skewed or eight_shards or uneven or unit_table or random_manifest or full_parity or wide_sums" \
  > $OUT/${TAG}_pytest_subset.log 2>&1; echo "pytest subset: exit $?"; tail -5 $OUT/${TAG}_pytest_subset.log
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
for cfg in 2 4; do
  for var in "" "KT_NO_FOLD_QUEUE=1"; do
    name=${TAG}_bench_cfg${cfg}$(echo "$var" | tr -c 'A-Za-z0-9\n' '_')
    env $var timeout 400 python bench.py --config $cfg --steps 300 --warmup 10 --no-cpu-baseline --no-latency --verify > $OUT/$name.json 2> $OUT/$name.err
    echo "bench cfg$cfg [$var]: exit $?"; show $OUT/$name.json; grep -v amdgpu.ids $OUT/$name.err | tail -2
  done
done
for var in "" "KT_NO_FOLD_QUEUE=1"; do
  name=${TAG}_bench_cfg2_4M$(echo "$var" | tr -c 'A-Za-z0-9\n' '_')
  env $var timeout 400 python bench.py --config 2 --pods-per-gpu 4000000 --steps 100 --warmup 5 --no-cpu-baseline --no-latency > $OUT/$name.json 2> $OUT/$name.err
  echo "bench cfg2 4M [$var]: exit $?"; show $OUT/$name.json
done
