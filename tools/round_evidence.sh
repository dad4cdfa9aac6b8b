#!/bin/bash
# Evidence of a round in ONE GPU call: bench lines of configs 2 (driver arguments, CPU baseline + latency legs), 2 (long
# run), 3, 1, 4 and 2 with 4M pods; rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes of configs 2, 4, 2@4M, 1, 3;
# SQ counters of configs 2 and 4.
#   gpurun --timeout 1700 -- 'bash tools/round_evidence.sh r03'   -> gpurun_out/<tag>_*, gpurun_out/prof_<tag>_cfg*
set -u
TAG=${1:-rXX}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_cfg2_driver.json 2> $OUT/${TAG}_bench_cfg2_driver.err; echo "bench cfg2 (driver arguments): exit $?"
timeout 300 python bench.py --config 2 --steps 2000 --warmup 20 --no-cpu-baseline --no-latency --no-extra > $OUT/${TAG}_bench_cfg2.json 2> $OUT/${TAG}_bench_cfg2.err; echo "bench cfg2: exit $?"
for cfg in 3 1 4; do
  timeout 400 python bench.py --config $cfg --steps 200 --warmup 10 --no-cpu-baseline > $OUT/${TAG}_bench_cfg${cfg}.json 2> $OUT/${TAG}_bench_cfg${cfg}.err; echo "bench cfg$cfg: exit $?"
done
timeout 400 python bench.py --config 2 --pods-per-gpu 4000000 --steps 200 --warmup 10 --no-cpu-baseline > $OUT/${TAG}_bench_cfg2_4M.json 2> $OUT/${TAG}_bench_cfg2_4M.err; echo "bench cfg2 4M pods: exit $?"
# D = 16 resource names / L = 16 labels per pod (not BASELINE configurations: the product path beyond 8 of each)
timeout 300 python bench.py --config 2 --dims 16 --steps 300 --warmup 10 --no-cpu-baseline --no-latency > $OUT/${TAG}_bench_cfg2_D16.json 2> $OUT/${TAG}_bench_cfg2_D16.err; echo "bench cfg2 D=16: exit $?"
timeout 300 python bench.py --config 2 --labels 16 --steps 300 --warmup 10 --no-cpu-baseline --no-latency > $OUT/${TAG}_bench_cfg2_L16.json 2> $OUT/${TAG}_bench_cfg2_L16.err; echo "bench cfg2 L=16: exit $?"
timeout 300 python bench.py --config 4 --sweep --steps 200 --warmup 10 --no-cpu-baseline --no-latency > $OUT/${TAG}_bench_cfg4_sweep.json 2> $OUT/${TAG}_bench_cfg4_sweep.err; echo "bench cfg4 (kt_sweep_launch): exit $?"
for f in $OUT/${TAG}_bench_cfg*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%s: %.3e %s %.4f ms/step | check %.3f aggregate %.3f reconcile %.3f step %.3f after-event %s | %s" % (sys.argv[1].split("/")[-1], d["value"], d["unit"], d["ms_per_step"],
          r["check"]["frac"], r["aggregate"]["frac"], r["reconcile"]["frac"], r["step"]["frac"], (r.get("step_after_event") or {}).get("ratio"), r["per_kernel_ms"]))
    if d.get("latency"): print("   latency:", {k: d["latency"][k] for k in ("check1", "check1_busy", "sweep", "throttle_event", "recompile") if k in d["latency"]})
except Exception as ex:
    print(sys.argv[1], "no bench line:", ex)
PY
done
timeout 600 bash tools/profile.sh ${TAG}_cfg2 --config 2 > $OUT/${TAG}_profile_cfg2.log 2>&1; echo "profile cfg2: exit $?"
timeout 600 bash tools/profile.sh ${TAG}_cfg4 --config 4 > $OUT/${TAG}_profile_cfg4.log 2>&1; echo "profile cfg4: exit $?"
timeout 600 bash tools/profile.sh ${TAG}_cfg2_4M --config 2 --pods-per-gpu 4000000 > $OUT/${TAG}_profile_cfg2_4M.log 2>&1; echo "profile cfg2 4M: exit $?"
timeout 600 bash tools/profile.sh ${TAG}_cfg1 --config 1 > $OUT/${TAG}_profile_cfg1.log 2>&1; echo "profile cfg1: exit $?"
timeout 600 bash tools/profile.sh ${TAG}_cfg3 --config 3 > $OUT/${TAG}_profile_cfg3.log 2>&1; echo "profile cfg3: exit $?"
timeout 400 bash tools/pmc_sq.sh ${TAG}_cfg2 --config 2 > $OUT/${TAG}_sq_cfg2.log 2>&1; echo "sq cfg2: exit $?"
timeout 400 bash tools/pmc_sq.sh ${TAG}_cfg4 --config 4 > $OUT/${TAG}_sq_cfg4.log 2>&1; echo "sq cfg4: exit $?"
for c in cfg2 cfg4 cfg2_4M; do head -9 $(find $OUT/prof_${TAG}_$c/trace -name "*kernel_stats.csv" | head -1); done
