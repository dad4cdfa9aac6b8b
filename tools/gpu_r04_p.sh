#!/bin/bash
# Round 4: the step on two streams (bench.py --overlap: check(i) beside reconcile(i + 1), check(i) behind finalize(i)) against
# the serial step on the same box; the overlapped loop must leave bit-identical summaries and status behind.
set -u
TAG=${1:-r04p}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
B="--steps 1000 --warmup 20 --no-cpu-baseline --no-latency"
run() {  # name, env, args
  local name=$1 envs=$2; shift 2
  env $envs timeout 300 python bench.py "$@" > $OUT/${TAG}_$name.json 2> $OUT/${TAG}_$name.err; echo "$name: exit $?"
}
run cfg2_serial "A=1" --config 2 $B
run cfg2_overlap "A=1" --config 2 $B --overlap
run cfg2_overlap_1wg "KT_CHECK_WGS_PER_CU=1" --config 2 $B --overlap
run cfg2_serial_b "A=1" --config 2 $B
run cfg3_overlap "A=1" --config 3 $B --overlap
run cfg2_4M_overlap "A=1" --config 2 --pods-per-gpu 4000000 --steps 300 --warmup 10 --no-cpu-baseline --no-latency --overlap
run cfg4_overlap "A=1" --config 4 --steps 200 --warmup 5 --no-cpu-baseline --no-latency --overlap
run cfg1_overlap "A=1" --config 1 $B --overlap
for f in $OUT/${TAG}_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%s: %.4f ms/step %.3e decisions/s streams=%s identical=%s" % (sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], d["config"].get("streams"), d["config"].get("overlap_identical_to_serial")))
except Exception as ex:
    print(sys.argv[1], "no bench line:", ex); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
done
