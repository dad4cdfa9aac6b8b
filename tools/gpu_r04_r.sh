#!/bin/bash
# Round 4: the packed fold taking a word's matches in scan_tile's post hook, two per step (kt_aggregate_bitmap PK and the
# fused sweep) — parity subset, then A/B against tools/ab/libkt_engine_prev.so (one match per peel step) on the same box.
set -u
TAG=${1:-r04r}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q -k "golden or random_small_rich or lean_sweep or edge_shapes or config1_full or config2_full or config3_overrides or multi_chunk_index or selector_errors or config4_one_shard and not 3 and not 7" > $OUT/${TAG}_pytest.log 2>&1; echo "pytest: exit $?"; tail -5 $OUT/${TAG}_pytest.log
B="--steps 1000 --warmup 20 --no-cpu-baseline --no-latency"
OLD=KT_ENGINE_LIB=$REPO/tools/ab/libkt_engine_prev.so
run() {  # name, env, args
  local name=$1 envs=$2; shift 2
  env $envs timeout 300 python bench.py "$@" > $OUT/${TAG}_$name.json 2> $OUT/${TAG}_$name.err; echo "$name: exit $?"
}
run cfg2_old "$OLD" --config 2 $B
run cfg2_new "A=1" --config 2 $B
run cfg2_old_b "$OLD" --config 2 $B
run cfg2_new_b "A=1" --config 2 $B
run cfg2_sweep_new "A=1" --config 2 $B --sweep
run cfg2_4M_old "$OLD" --config 2 --pods-per-gpu 4000000 --steps 300 --warmup 10 --no-cpu-baseline --no-latency
run cfg2_4M_new "A=1" --config 2 --pods-per-gpu 4000000 --steps 300 --warmup 10 --no-cpu-baseline --no-latency
run cfg4_old "$OLD" --config 4 --steps 100 --warmup 5 --no-cpu-baseline --no-latency
run cfg4_new "A=1" --config 4 --steps 100 --warmup 5 --no-cpu-baseline --no-latency
for f in $OUT/${TAG}_cfg*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%s: %.4f ms/step | %s | per kernel %s" % (sys.argv[1].split("/")[-1], d["ms_per_step"], d["config"]["engine_version"][-16:], r["per_kernel_ms"]))
except Exception as ex:
    print(sys.argv[1], "no bench line:", ex); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
done
