#!/bin/bash
# Round 4: kt_sweep_launch (the PreFilter sweep and the reconcile scan fused into one pass: kt_check_bitmap's AGG
# instantiation) — parity (run_full_parity and the full-size checks compare it with the oracle), then the step as one sweep
# against the step as reconcile + check on the same box.
set -u
TAG=${1:-r04q}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q -k "golden or random_small_rich or lean_sweep or edge_shapes or config1_full or config2_full or config3_overrides or multi_chunk_index or selector_errors" > $OUT/${TAG}_pytest.log 2>&1; echo "pytest: exit $?"; tail -15 $OUT/${TAG}_pytest.log
B="--steps 1000 --warmup 20 --no-cpu-baseline --no-latency"
run() {  # name, env, args
  local name=$1 envs=$2; shift 2
  env $envs timeout 300 python bench.py "$@" > $OUT/${TAG}_$name.json 2> $OUT/${TAG}_$name.err; echo "$name: exit $?"
}
run cfg2_pair "A=1" --config 2 $B
run cfg2_sweep "A=1" --config 2 $B --sweep
run cfg3_sweep "A=1" --config 3 $B --sweep
run cfg1_sweep "A=1" --config 1 $B --sweep
run cfg2_4M_sweep "A=1" --config 2 --pods-per-gpu 4000000 --steps 300 --warmup 10 --no-cpu-baseline --no-latency --sweep
run cfg4_sweep "A=1" --config 4 --steps 100 --warmup 5 --no-cpu-baseline --no-latency --sweep
for f in $OUT/${TAG}_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%s: %.4f ms/step %.3e decisions/s | %s %s | per kernel %s" % (sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], r["check"]["kernel"], r.get("fused_sweep"), r["per_kernel_ms"]))
except Exception as ex:
    print(sys.argv[1], "no bench line:", ex); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
done
