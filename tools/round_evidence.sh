#!/bin/bash
# Evidence of a round in ONE GPU call: bench lines of configs 2/3/1/4 (config 2 with the CPU baseline leg), the
# rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes of configs 2 and 4, SQ counters of config 2.
#   gpurun --timeout 1500 -- 'bash tools/round_evidence.sh r02'   -> gpurun_out/<tag>_*, gpurun_out/prof_<tag>_cfg*
set -u
TAG=${1:-rXX}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 600 python bench.py --config 2 --steps 200 --warmup 10 > $OUT/${TAG}_bench_cfg2.json 2> $OUT/${TAG}_bench_cfg2.err; echo "bench cfg2: exit $?"
for cfg in 3 1 4; do
  timeout 400 python bench.py --config $cfg --steps 50 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench_cfg${cfg}.json 2> $OUT/${TAG}_bench_cfg${cfg}.err; echo "bench cfg$cfg: exit $?"
done
timeout 400 python bench.py --config 2 --pods-per-gpu 4000000 --steps 30 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench_cfg2_4M.json 2> $OUT/${TAG}_bench_cfg2_4M.err; echo "bench cfg2 4M pods: exit $?"
for f in $OUT/${TAG}_bench_cfg*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%s: %.3e %s %.4f ms/step | check %.1f%% aggregate %.1f%% | %s" % (sys.argv[1].split("/")[-1], d["value"], d["unit"], d["ms_per_step"],
          100 * r["check"]["frac"], 100 * r["aggregate"]["frac"], r["per_kernel_ms"]))
except Exception as ex:
    print(sys.argv[1], "no bench line:", ex)
PY
done
timeout 900 bash tools/profile.sh ${TAG}_cfg2 --config 2 > $OUT/${TAG}_profile_cfg2.log 2>&1; echo "profile cfg2: exit $?"
timeout 900 bash tools/profile.sh ${TAG}_cfg4 --config 4 > $OUT/${TAG}_profile_cfg4.log 2>&1; echo "profile cfg4: exit $?"
timeout 600 bash tools/pmc_sq.sh ${TAG}_cfg2 --config 2 > $OUT/${TAG}_sq_cfg2.log 2>&1; echo "sq cfg2: exit $?"
head -12 $(find $OUT/prof_${TAG}_cfg2/trace -name "*kernel_stats.csv" | head -1)
