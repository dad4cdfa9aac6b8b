#!/bin/bash
# Round 4, call F: the event kernels signal a pinned sequence number (settle_ingest polls it instead of waiting on the
# event) — parity subset around pod events / few-pod checks / concurrency, the latency leg with its A/B switches
# (KT_INGEST_EVENT_WAIT=1: hipEventSynchronize as before; KT_FEED_NO_STAGE=1: the kernel walks the pinned slot), then the
# round's evidence (bench lines, rocprofv3 kernel stats, FETCH/WRITE passes, SQ counters) on these sources.
set -u
TAG=${1:-r04f}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q \
  -k "pod_events or event_bursts or incremental or few_pod or concurrent or abi_flat or host_gpu or plugin or golden or lean_sweep or random_small or namespace_order or beyond_the_compiled or metrics or paging or throttle_events" \
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
for var in "KT_X=0" "KT_INGEST_EVENT_WAIT=1" "KT_FEED_NO_STAGE=1"; do
  name=${TAG}_lat_$(echo "$var" | tr -c 'A-Za-z0-9\n' '_')
  env $var timeout 400 python bench.py --config 2 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err
  echo "latency leg [$var]: exit $?"; lat $OUT/$name.json; grep -v amdgpu.ids $OUT/$name.err | tail -2
done
bash tools/round_evidence.sh $TAG
