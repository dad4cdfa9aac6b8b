#!/bin/bash
# A/B timing of library variants (tools/build_variant.sh) in ONE GPU call: for every variant and config a short bench run;
# prints ms/step, the per-kernel HIP-event times and results_sha1 (equal fingerprints = bit-identical results of the step).
#   gpurun --timeout 900 -- 'bash tools/gpu_ab.sh <tag> "base q8" "2 4" [extra bench args]'
set -u
TAG=${1:-ab}; LIBS=${2:-base}; CFGS=${3:-"2 4"}; EXTRA=${4:-}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
for cfg in $CFGS; do
  steps=200; [ "$cfg" == "2" ] && steps=1000
  for lib in $LIBS; do
    base=${lib%%@*}; envs=""; [ "$base" != "$lib" ] && envs=$(echo "${lib#*@}" | tr ',' ' ')   # name@VAR=val,VAR2=val2
    so=$REPO/tools/ab/libkt_engine_$base.so
    [ "$base" == "tree" ] && so=$REPO/kube_throttler_amd/csrc/libkt_engine.so
    name=${TAG}_$(echo "$lib" | tr -c 'A-Za-z0-9\n' '_')_cfg${cfg}
    env $envs KT_ENGINE_LIB=$so timeout 400 python bench.py --config $cfg --steps $steps --warmup 10 --no-cpu-baseline --no-latency --no-extra $EXTRA > $OUT/$name.json 2> $OUT/$name.err
    rc=$?
    python - "$OUT/$name.json" "$lib" "$cfg" "$rc" <<'PY'
import json, sys
f, lib, cfg, rc = sys.argv[1:5]
try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    k = d["roofline"]["per_kernel_ms"]
    print("cfg%s %-10s %.4f ms/step | check %.4f agg %.4f reduce %.4f finalize %.4f prepare %.4f | sha1 %s | %s" % (
        cfg, lib, d["ms_per_step"], k["check"], k["aggregate"], k["reduce"], k["finalize"], k["prepare"],
        d["config"]["results_sha1"], d["config"]["engine_version"].split("src=")[-1]))
except Exception as ex:
    print("cfg%s %-10s exit %s: no bench line (%s)" % (cfg, lib, rc, ex))
PY
    [ $rc -ne 0 ] && tail -3 $OUT/$name.err
  done
done
