#!/bin/bash
# Round 4, last call: the -m gpu suite on the final sources (without the configs[4] shards: shard 0 ran on these very sources
# in tools/gpu_r04_r.sh, all three on the kernels of tools/gpu_r04_final.sh), the round's evidence with the quoted bench lines
# (tools/gpu_r04_evidence2.sh), and the bench lines of the two step forms that were measured and not made the default
# (--sweep: the fused kernel; --overlap: two streams).
set -u
TAG=${1:-r04z}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 1300 python -m pytest tests -m gpu -q --deselect "tests/test_engine_gpu.py::test_config4_one_shard[0]" --deselect "tests/test_engine_gpu.py::test_config4_one_shard[3]" --deselect "tests/test_engine_gpu.py::test_config4_one_shard[7]" > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest -m gpu: exit $?"; tail -5 $OUT/${TAG}_pytest_gpu.log
bash tools/gpu_r04_evidence2.sh $TAG
B="--steps 1000 --warmup 20 --no-cpu-baseline --no-latency"
for c in 1 2 3; do
  timeout 300 python bench.py --config $c $B --sweep > $OUT/${TAG}_sweep_bench_cfg$c.json 2> $OUT/${TAG}_sweep_bench_cfg$c.err; echo "sweep bench cfg$c: exit $?"
done
timeout 300 python bench.py --config 2 --pods-per-gpu 4000000 --steps 300 --warmup 10 --no-cpu-baseline --no-latency --sweep > $OUT/${TAG}_sweep_bench_cfg2_4M.json 2> $OUT/${TAG}_sweep_bench_cfg2_4M.err; echo "sweep bench cfg2 4M: exit $?"
timeout 300 python bench.py --config 2 $B --overlap > $OUT/${TAG}_overlap_bench_cfg2.json 2> $OUT/${TAG}_overlap_bench_cfg2.err; echo "overlap bench cfg2: exit $?"
for f in $OUT/${TAG}_sweep_bench_*.json $OUT/${TAG}_overlap_bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%s: %.4f ms/step %.3e decisions/s | %s fused=%s streams=%s | %s" % (sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], r["check"]["kernel"], r.get("fused_sweep"), d["config"].get("streams"), r["per_kernel_ms"]))
except Exception as ex:
    print(sys.argv[1], "no bench line:", ex)
PY
done
