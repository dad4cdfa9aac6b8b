#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
( time timeout 600 python -m pytest tests -m gpu -x -q -k "not config4" ) > $OUT/r03d_pytest_gpu.log 2>&1; echo "pytest -m gpu (without config4): exit $?"; tail -15 $OUT/r03d_pytest_gpu.log
show() {
python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%-34s %.4f ms/step | agg %s %.4f ms (frac %.3f) reduce %.4f finalize %.4f check %.4f" % (sys.argv[2], d["ms_per_step"], r["aggregate"]["kernel"], r["per_kernel_ms"]["aggregate"], r["aggregate"]["frac"], r["per_kernel_ms"]["reduce"], r["per_kernel_ms"]["finalize"], r["per_kernel_ms"]["check"]))
except Exception as ex:
    print(sys.argv[2], "no bench line:", ex); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
}
for cfg in 2 4; do
  for var in "" "KT_AGG_WGS_PER_CU=1" "KT_PK_NOPAD=1" "KT_NO_PACK=1"; do
    name=r03d_bench_cfg${cfg}_$(echo "$var" | tr -c 'A-Za-z0-9\n' '_')
    env $var KT_DEBUG_LDS=1 timeout 300 python bench.py --config $cfg --steps 200 --warmup 10 --no-cpu-baseline --no-latency --verify > $OUT/$name.json 2> $OUT/$name.err
    echo "bench cfg$cfg [$var]: exit $?"; grep -m1 "kt_aggregate_bitmap:" $OUT/$name.err; show $OUT/$name.json "cfg$cfg $var"
  done
done
