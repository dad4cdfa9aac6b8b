#!/bin/bash
# Round 4, the last GPU call: word forms (groups of a class numbered by form; a word without veto bits is scanned by reading
# the `any` half of the atom rows only) — the -m gpu suite without the configs[4] shards, the full-size configs[4] shard
# step of this build against tools/ab/libkt_engine_prev.so (a build whose shard results were compared with the oracle) by
# the fingerprint of everything the step leaves behind (bench.py: results_sha1) and by time, then the round's evidence.
set -u
TAG=${1:-r04v}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 400 python -m pytest tests -m gpu -q -x --deselect "tests/test_engine_gpu.py::test_config4_one_shard[0]" --deselect "tests/test_engine_gpu.py::test_config4_one_shard[3]" --deselect "tests/test_engine_gpu.py::test_config4_one_shard[7]" > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest -m gpu: exit $?"; tail -3 $OUT/${TAG}_pytest_gpu.log
OLD=KT_ENGINE_LIB=$REPO/tools/ab/libkt_engine_prev.so
run() {  # name, env, args
  local name=$1 envs=$2; shift 2
  env $envs timeout 200 python bench.py "$@" > $OUT/${TAG}_$name.json 2> $OUT/${TAG}_$name.err; echo "$name: exit $?"
}
run ab_cfg4_old "$OLD" --config 4 --steps 100 --warmup 5 --no-cpu-baseline --no-latency
run ab_cfg4_new "A=1" --config 4 --steps 100 --warmup 5 --no-cpu-baseline --no-latency
for f in $OUT/${TAG}_ab_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%s: %.4f ms/step | results %s | %s | %s" % (sys.argv[1].split("/")[-1], d["ms_per_step"], d["config"].get("results_sha1"), d["config"]["engine_version"][-16:], r["per_kernel_ms"]))
except Exception as ex:
    print(sys.argv[1], "no bench line:", ex); print(open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
done
KT_SKIP_SUBSET=1 bash tools/gpu_r04_evidence2.sh $TAG 2>&1 | tail -12
