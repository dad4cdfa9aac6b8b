cd $GRAFT_REPO_ROOT
for e in 0 1 2 3 4; do
  for w in 2 1; do
  KT_EXP=$e KT_CHECK_WGS_PER_CU=$w timeout 300 python bench.py --config 2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('exp=$e wgs=$w', d['roofline']['per_kernel_ms'])"
  done
done
