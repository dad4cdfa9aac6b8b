#!/bin/bash
# Round 4, call N: kt_feed_few (an informer event proper: one wave per pod, the slot in LDS) — parity around pod events
# incl. test_event_sized_batches_match_bulk, the latency leg with KT_NO_FEED_FEW=1 as the A/B.
set -u
TAG=${1:-r04n}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q \
  -k "event_sized or pod_events or event_bursts or incremental or few_pod or concurrent or abi_flat or host_gpu or plugin or golden or namespace_order or beyond_the_compiled or metrics or paging or throttle_events or wide_label or overflow_guard" \
  > $OUT/${TAG}_pytest_subset.log 2>&1; echo "pytest subset: exit $?"; tail -4 $OUT/${TAG}_pytest_subset.log
lat() {
python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("  %.4f ms/step" % d["ms_per_step"], d["roofline"]["per_kernel_ms"])
    for k, v in (d.get("latency") or {}).items():
        if k.startswith(("upsert", "delete", "check1", "sweep")): print("     ", k, v)
except Exception as ex:
    print("  no bench line:", ex)
PY
}
for var in "KT_X=0" "KT_NO_FEED_FEW=1"; do
  name=${TAG}_lat_$(echo "$var" | tr -c 'A-Za-z0-9\n' '_')
  env $var timeout 400 python bench.py --config 2 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err
  echo "latency leg [$var]: exit $?"; lat $OUT/$name.json; grep -v amdgpu.ids $OUT/$name.err | tail -2
done
env timeout 400 python bench.py --config 4 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_lat_cfg4.json 2> $OUT/${TAG}_lat_cfg4.err; echo "latency leg cfg4: exit $?"; lat $OUT/${TAG}_lat_cfg4.json
