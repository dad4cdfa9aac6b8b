#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; mkdir -p gpurun_out
for v in "--dims 16" "--labels 16" "--dims 16 --labels 16"; do
  name=r05p_cfg2$(echo "$v" | tr -c 'A-Za-z0-9\n' '_')
  timeout 300 python bench.py --config 2 --steps 500 --warmup 10 --no-cpu-baseline --no-latency --no-extra $v > gpurun_out/$name.json 2> gpurun_out/$name.err
  python - gpurun_out/$name.json "$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k = d["roofline"]["per_kernel_ms"]; r = d["roofline"]
print("cfg2 %-22s %.4f ms/step | check %.4f (%s, frac %.3f) agg %.4f (%s, frac %.3f) fin %.4f | step frac %.3f" % (sys.argv[2], d["ms_per_step"], k["check"], r["check"]["kernel"], r["check"]["frac"], k["aggregate"], r["aggregate"]["kernel"], r["aggregate"]["frac"], k["finalize"] + k["reduce"], r["step"]["frac"]))
PY
done
bash tools/gpu_ab.sh r05p "tree" "2 4"
