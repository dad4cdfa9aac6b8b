#!/bin/bash
# Round 4, call O: namespace-ordered scans skip, per TILE, the chunks its own namespaces have no word in (the workgroup's
# chunk loop walks the union over its tile range) — parity on the chunked paths + configs[4] shard 7, bench A/B
# (KT_NO_TILE_SKIP=1).
set -u
TAG=${1:-r04o}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q -k "multi_chunk or namespace_order or sharded or shards or eight or skewed or uneven or pod_events or event_bursts or few_pod or incremental or golden or lean_sweep or beyond_the_compiled" > $OUT/${TAG}_pytest_subset.log 2>&1; echo "pytest subset: exit $?"; tail -3 $OUT/${TAG}_pytest_subset.log
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "one_shard and 7" > $OUT/${TAG}_pytest_shard7.log 2>&1; echo "configs[4] shard 7: exit $?"; tail -3 $OUT/${TAG}_pytest_shard7.log
show() {
python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("  %.4f ms/step | %s" % (d["ms_per_step"], r["per_kernel_ms"]))
except Exception as ex:
    print("  no bench line:", ex)
PY
}
for rep in 1 2; do
for var in "KT_X=0" "KT_NO_TILE_SKIP=1"; do
  name=${TAG}_bench_cfg4_$(echo "$var" | tr -c 'A-Za-z0-9\n' '_')_$rep
  env $var timeout 400 python bench.py --config 4 --steps 100 --warmup 5 --no-cpu-baseline --no-latency --verify > $OUT/$name.json 2> $OUT/$name.err
  echo "bench cfg4 [$var]: exit $?"; show $OUT/$name.json; grep -v amdgpu.ids $OUT/$name.err | tail -2
done
done
