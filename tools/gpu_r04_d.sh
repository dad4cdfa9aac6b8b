#!/bin/bash
# Round 4, call D: the single-launch pod event path (kt_feed_small) — parity subset around pod events / few-pod checks /
# the flat C entry points / the C++ plugin mirror / torch-backed sharded tests, the stress test on a third box, and the
# bench line of configs[2] with the latency leg.
set -u
TAG=${1:-r04d}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q \
  -k "pod_events or event_bursts or incremental or few_pod or concurrent or abi_flat or host_gpu or plugin or golden or sharded or skewed or eight_shards or uneven or edge_shapes or wide_label or namespace_order or beyond_the_compiled or admit or metrics or paging" \
  > $OUT/${TAG}_pytest_subset.log 2>&1; echo "pytest subset: exit $?"; tail -5 $OUT/${TAG}_pytest_subset.log
KT_STRESS_ROUNDS=500 timeout 600 python -m pytest tests/test_engine_gpu.py -k stress_fresh -x -q -s > $OUT/${TAG}_stress.log 2>&1; echo "stress x500: exit $?"; grep -E "fresh-engine|passed|failed|round " $OUT/${TAG}_stress.log | head -8
timeout 400 python bench.py --config 2 --steps 500 --warmup 10 --no-cpu-baseline --verify > $OUT/${TAG}_bench_cfg2.json 2> $OUT/${TAG}_bench_cfg2.err; echo "bench cfg2: exit $?"
python - $OUT/${TAG}_bench_cfg2.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("  %.3e %s  %.4f ms/step" % (d["value"], d["unit"], d["ms_per_step"]), d["roofline"]["per_kernel_ms"])
for k, v in (d.get("latency") or {}).items():
    if k.startswith(("upsert", "delete", "check1", "sweep")): print("   ", k, v)
PY
grep -v amdgpu.ids $OUT/${TAG}_bench_cfg2.err | tail -3
