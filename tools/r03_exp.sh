#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
for cfg in 4; do
for e in 0 1 5 7 8; do
  KT_EXP=$e timeout 300 python bench.py --config $cfg --steps 50 --warmup 5 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg$cfg exp=$e', {k: round(v*1e3,1) for k,v in d['roofline']['per_kernel_ms'].items()}, 'step %.1f us' % (d['ms_per_step']*1e3))"
done
done
